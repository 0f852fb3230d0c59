"""Guard-band tests of every caller-owned output buffer of the C ABI (SURVEY.md section 5 hooks
"guard-banded device buffers").  The reference's ABI passes raw pointers of IMPLICIT length (`generate_spectrogram`
writes 240 000 doubles wherever `output` points and 200 + 200 reflect samples around the caller's audio, lib.rs:112-121),
and the model entries follow that convention: the callee computes the extent from B and the model dimensions.  Every
output here is allocated with 4 KiB canaries on both sides, at odd batch sizes (1, 7, 17, 57 -- partial batch blocks,
partial tiles), and the canaries must come back untouched; so must a weight-file parser fed truncated / lying files."""
import ctypes
import struct

import numpy as np
import pytest

from oracle import logmel_np as L
from oracle import whisper_ref as R

pytestmark = pytest.mark.gpu
GUARD = 4096


class Guarded:
    """A host array of `n` elements of `dtype` between two GUARD-byte canaries."""

    def __init__(self, n, dtype):
        self.dt = np.dtype(dtype)
        self.n = int(n)
        self.raw = np.full(2 * GUARD + self.n * self.dt.itemsize, 0xC3, dtype=np.uint8)
        self.view = self.raw[GUARD:GUARD + self.n * self.dt.itemsize].view(self.dt)

    @property
    def ptr(self):
        return ctypes.c_void_p(self.raw.ctypes.data + GUARD)

    def intact(self):
        return bool(np.all(self.raw[:GUARD] == 0xC3) and np.all(self.raw[-GUARD:] == 0xC3))


@pytest.fixture(scope="module")
def model(pkg):
    dims = dict(R.TINY_DIMS)
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(5, matrix_gain=4.0)
    ctx.finalize()
    yield dims, ctx
    ctx.close()


def _pcm(n):
    return np.stack([L.synth_chunk(400 + i).astype(np.float32) for i in range(min(n, 4))])[np.arange(n) % min(n, 4)]


@pytest.mark.parametrize("B", [1, 7, 17, 57])
def test_model_outputs_stay_inside_their_buffers(pkg, model, B):
    dims, ctx = model
    lib, Bn = ctx.lib, pkg.binding
    d, V, T = dims["n_audio_state"], dims["n_vocab"], 5
    pcm = np.ascontiguousarray(_pcm(B))
    # wm_logmel: f32 [B][80][3000] and the f64 flavour
    for dt, code in ((np.float32, Bn.WM_F32), (np.float64, Bn.WM_F64)):
        mel = Guarded(B * 80 * 3000, dt)
        assert lib.wm_logmel(ctx.handle, pcm.ctypes.data_as(ctypes.c_void_p), Bn.WM_F32, B, 80, mel.ptr, code, Bn.WM_MEM_HOST) == 0
        assert mel.intact() and np.isfinite(mel.view).all()
    mel32 = np.ascontiguousarray(mel.view.reshape(B, 80, 3000).astype(np.float32))
    # wm_encode: f32 [B][1500][d]
    xa = Guarded(B * 1500 * d, np.float32)
    assert lib.wm_encode(ctx.handle, mel32.ctypes.data_as(ctypes.c_void_p), B, xa.ptr, Bn.WM_MEM_HOST) == 0
    assert xa.intact() and np.isfinite(xa.view).all()
    xa_in = np.ascontiguousarray(xa.view.reshape(B, 1500, d))
    # wm_decode_logits: f32 [B][T][n_vocab]
    tok = np.ascontiguousarray(np.random.default_rng(B).integers(0, V, size=(B, T)).astype(np.int32))
    lg = Guarded(B * T * V, np.float32)
    assert lib.wm_decode_logits(ctx.handle, tok.ctypes.data_as(ctypes.c_void_p), B, T, xa_in.ctypes.data_as(ctypes.c_void_p), lg.ptr,
                                Bn.WM_MEM_HOST) == 0
    assert lg.intact() and np.isfinite(lg.view).all()
    # wm_detect_language: i32 [B]
    li = Guarded(B, np.int32)
    assert lib.wm_detect_language(ctx.handle, xa_in.ctypes.data_as(ctypes.c_void_p), B, 1, 2, min(V - 1, 40), li.ptr, Bn.WM_MEM_HOST) == 0
    assert li.intact() and np.all((li.view >= 0) & (li.view <= 38))
    # wm_transcribe_greedy: i32 [B][max_new] + i32 [B]; with early stop on (eot) so that the padding path writes too
    prompt = np.array([1, 2], dtype=np.int32)
    for max_new, eot in ((9, -1), (13, 3)):
        toks, lens = Guarded(B * max_new, np.int32), Guarded(B, np.int32)
        assert lib.wm_transcribe_greedy(ctx.handle, pcm.ctypes.data_as(ctypes.c_void_p), Bn.WM_F32, B, prompt.ctypes.data_as(ctypes.c_void_p),
                                        2, max_new, eot, toks.ptr, lens.ptr, Bn.WM_MEM_HOST) == 0, lib.wm_last_error()
        assert toks.intact() and lens.intact()
        assert np.all((lens.view >= 1) & (lens.view <= max_new)) and np.all((toks.view >= -1) & (toks.view < V))


def test_generate_spectrogram_writes_exactly_its_two_extents(pkg):
    """bridge.h:11 / lib.rs:110-122: `audio` is 480 400 doubles (the callee writes the 200 + 200 reflect samples into it),
    `output` 240 000 doubles -- and not one byte more on either side."""
    lib = pkg.binding.load_library()
    audio, out = Guarded(480400, np.float64), Guarded(240000, np.float64)
    audio.view[:] = 0.0
    audio.view[200:480200] = L.synth_chunk(9)
    lib.generate_spectrogram(audio.ptr, out.ptr)
    assert audio.intact() and out.intact() and np.isfinite(out.view).all()
    assert np.array_equal(audio.view[:200], audio.view[400:200:-1])          # the reflect pad was written (lib.rs:35-36)


def test_wav_chunker_and_front_end_outputs_are_guarded(pkg, tmp_path):
    lib, Bn = pkg.binding.load_library(), pkg.binding
    n = 480000 + 1234
    data = (np.arange(n) % 4001 - 2000).astype("<i2")
    p = tmp_path / "two_chunks.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", 36 + 2 * n) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data"
                  + struct.pack("<I", 2 * n) + data.tobytes())
    lib.wm_wav_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.wm_wav_read_chunks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.wm_wav_num_chunks.argtypes = [ctypes.c_void_p]
    lib.wm_wav_close.argtypes = [ctypes.c_void_p]
    h = ctypes.c_void_p()
    assert lib.wm_wav_open(str(p).encode(), ctypes.byref(h)) == 0 and lib.wm_wav_num_chunks(h) == 2
    pcm = Guarded(2 * 480000, np.int16)
    assert lib.wm_wav_read_chunks(h, 0, 2, pcm.ptr) == 0 and pcm.intact()
    assert np.array_equal(pcm.view[:n], data) and not pcm.view[n:].any()
    lib.wm_wav_close(h)
    fe = Bn.Context()
    mel = Guarded(2 * 128 * 3000, np.float32)
    assert fe.lib.wm_logmel(fe.handle, pcm.ptr, Bn.WM_I16, 2, 128, mel.ptr, Bn.WM_F32, Bn.WM_MEM_HOST) == 0 and mel.intact()
    fe.close()


def test_malformed_weight_files_are_refused_without_touching_memory(pkg, tmp_path):
    """wm_load_weights parses a caller-supplied file (name lengths, element counts come from the file).  Truncations at every
    structural boundary, a lying element count, a lying name length, an unknown tensor, wrong dimensions and a wrong magic:
    each must come back as WM_ERR_IO (or INVALID) with a message -- and the context must still work afterwards."""
    import importlib
    W = importlib.import_module("openai_whisper_coreml_amd.weights")
    dims = dict(R.TINY_DIMS)
    sd = W.synthetic_state_dict(dims, seed=3)
    good = tmp_path / "good.wm"
    W.save_flat(str(good), dims, sd)
    raw = good.read_bytes()
    ctx = pkg.binding.Context(dims)
    ctx.load_weights(str(good))
    first_name_len = struct.unpack_from("<i", raw, 52)[0]
    cases = {
        "empty": b"", "magic": b"XXXXXXXX" + raw[8:], "dims": raw[:8] + struct.pack("<i", 81) + raw[12:],
        "count_negative": raw[:48] + struct.pack("<i", -4) + raw[52:], "cut_in_header": raw[:30],
        "cut_in_name": raw[:52 + 4 + first_name_len // 2], "cut_in_count": raw[:52 + 4 + first_name_len + 3],
        "cut_in_data": raw[:52 + 4 + first_name_len + 8 + 10], "cut_last_byte": raw[:-1],
        "name_len_huge": raw[:52] + struct.pack("<i", 2 ** 30) + raw[56:], "name_len_negative": raw[:52] + struct.pack("<i", -1) + raw[56:],
        "elems_huge": raw[:56 + first_name_len] + struct.pack("<q", 2 ** 60) + raw[64 + first_name_len:],
        "elems_negative": raw[:56 + first_name_len] + struct.pack("<q", -8) + raw[64 + first_name_len:],
        "unknown_tensor": raw[:56] + b"Z" * first_name_len + raw[56 + first_name_len:],
    }
    for name, blob in cases.items():
        p = tmp_path / ("bad_%s.wm" % name)
        p.write_bytes(blob)
        rc = ctx.lib.wm_load_weights(ctx.handle, str(p).encode())
        assert rc != 0 and ctx.lib.wm_last_error(), name
    rc = ctx.lib.wm_load_weights(ctx.handle, str(tmp_path / "does_not_exist.wm").encode())
    assert rc != 0
    ctx.load_weights(str(good))                       # still usable
    ctx.finalize()
    toks, lens = ctx.transcribe_greedy(_pcm(1), [1, 2], 4)
    assert toks.shape == (1, 4)
    ctx.close()
