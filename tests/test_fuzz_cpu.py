"""Malformed-input tests of the host-only parsers behind the C ABI: the library reads untrusted bytes
in three places that need no GPU -- a WAV file (`wm_wav_*`, csrc/audio.cpp), a tokenizer vocabulary (`wm_vocab_load`,
csrc/detok.cpp) and the token payload of the all-gather (`wm_multi_unpack_tokens`, csrc/multi.cpp).  Whatever the bytes
say, a call must come back with a status (never crash, never read or write outside its buffers).  Under the normal build
a wild access shows up as a crash of the test process; the SANITIZER LEG (tools/run_sanitized.sh: the same tests against
libwhisper_mi355x_asan.so, host code under ASan + UBSan) turns every out-of-bounds byte and every signed overflow into a
failure.  (The weight-file parser needs a model context, i.e. a GPU: tests/test_model_gpu.py::test_malformed_weight_files.)

Reference: the reference reads query.wav through AVFoundation (AudioRecorder.swift:74-86) and has no parser of its own;
`generate_spectrogram` trusts its two pointers blindly (lib.rs:112,116)."""
import ctypes
import json
import os
import struct

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

FUZZ = settings(max_examples=int(os.environ.get("WM_FUZZ_EXAMPLES", "120")), deadline=None,
                suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module")
def lib(pkg):
    L = pkg.binding.load_library()
    L.wm_wav_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    L.wm_wav_close.argtypes = [ctypes.c_void_p]
    L.wm_wav_num_samples.argtypes = [ctypes.c_void_p]
    L.wm_wav_num_samples.restype = ctypes.c_long
    L.wm_wav_num_chunks.argtypes = [ctypes.c_void_p]
    L.wm_wav_read_chunks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.wm_vocab_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    L.wm_vocab_free.argtypes = [ctypes.c_void_p]
    L.wm_vocab_size.argtypes = [ctypes.c_void_p]
    L.wm_detokenize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                ctypes.POINTER(ctypes.c_size_t)]
    L.wm_multi_unpack_tokens.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p]
    L.wm_multi_pack_tokens.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return L


def _wav_bytes(n_samples, rate=16000, channels=1, bits=16, fmt=1, data_len=None, riff_len=None, extra_chunks=b""):
    data = (np.arange(n_samples, dtype=np.int64) % 2001 - 1000).astype("<i2").tobytes()
    dl = len(data) if data_len is None else data_len
    fmt_chunk = b"fmt " + struct.pack("<IHHIIHH", 16, fmt, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    body = b"WAVE" + fmt_chunk + extra_chunks + b"data" + struct.pack("<I", dl & 0xffffffff) + data
    rl = len(body) if riff_len is None else riff_len
    return b"RIFF" + struct.pack("<I", rl & 0xffffffff) + body


def _try_wav(lib, path):
    """Open + read through the C ABI the way a host would; returns the status of wm_wav_open."""
    h = ctypes.c_void_p()
    st_ = lib.wm_wav_open(path.encode(), ctypes.byref(h))
    if st_ != 0:
        assert not h.value, "a failed wm_wav_open must not hand out a handle"
        assert lib.wm_last_error()
        return st_
    n, nc = lib.wm_wav_num_samples(h), lib.wm_wav_num_chunks(h)
    assert n >= 0 and nc >= 1 and nc == max(1, -(-n // 480000)), (n, nc)
    guard = 4096
    out = np.full(min(nc, 2) * 480000 + 2 * guard, 0x5a5a, dtype=np.int16)
    rc = lib.wm_wav_read_chunks(h, 0, min(nc, 2), out[guard:].ctypes.data_as(ctypes.c_void_p))
    assert np.all(out[:guard] == 0x5a5a) and np.all(out[-guard:] == 0x5a5a), "wm_wav_read_chunks wrote outside its buffer"
    assert lib.wm_wav_read_chunks(h, nc, 1, out[guard:].ctypes.data_as(ctypes.c_void_p)) != 0     # past the end: refused
    assert lib.wm_wav_read_chunks(h, -1, 1, out[guard:].ctypes.data_as(ctypes.c_void_p)) != 0
    lib.wm_wav_close(h)
    return rc


def test_a_well_formed_wav_still_reads(lib, tmp_path):
    p = tmp_path / "ok.wav"
    p.write_bytes(_wav_bytes(1000, extra_chunks=b"LIST" + struct.pack("<I", 4) + b"abcd"))
    assert _try_wav(lib, str(p)) == 0


@FUZZ
@given(n=st.integers(0, 3000), rate=st.sampled_from([16000, 8000, 0, 44100]), ch=st.integers(0, 3), bits=st.sampled_from([16, 8, 24, 0]),
       fmt=st.sampled_from([1, 3, 0xfffe]), data_len=st.one_of(st.none(), st.integers(0, 2 ** 32 - 1)),
       riff_len=st.one_of(st.none(), st.integers(0, 2 ** 32 - 1)), cut=st.integers(0, 200), junk=st.binary(max_size=64))
def test_malformed_wav_headers_never_crash(lib, tmp_path, n, rate, ch, bits, fmt, data_len, riff_len, cut, junk):
    raw = _wav_bytes(n, rate, ch, bits, fmt, data_len, riff_len, extra_chunks=junk if len(junk) % 2 == 0 else b"")
    raw = raw[:max(0, len(raw) - cut)]
    p = tmp_path / "f.wav"
    p.write_bytes(raw)
    _try_wav(lib, str(p))      # any status; the assertions inside are about buffers and handles


@FUZZ
@given(blob=st.binary(max_size=400))
def test_random_bytes_as_wav_never_crash(lib, tmp_path, blob):
    p = tmp_path / "r.wav"
    p.write_bytes(blob)
    _try_wav(lib, str(p))
    p.write_bytes(b"RIFF" + blob)
    _try_wav(lib, str(p))


def _try_vocab(lib, path, ids):
    h = ctypes.c_void_p()
    st_ = lib.wm_vocab_load(path.encode(), ctypes.byref(h))
    if st_ != 0:
        assert not h.value
        return st_
    assert lib.wm_vocab_size(h) >= 0
    ids = np.asarray(ids, dtype=np.int32)
    need = ctypes.c_size_t(0)
    assert lib.wm_detokenize(h, ids.ctypes.data_as(ctypes.c_void_p), len(ids), 0, None, 0, ctypes.byref(need)) == 0
    guard = 64
    buf = np.full(need.value + 2 * guard, 0xa5, dtype=np.uint8)
    for cap in (need.value, max(0, need.value // 2), 1):
        buf[:] = 0xa5
        rc = lib.wm_detokenize(h, ids.ctypes.data_as(ctypes.c_void_p), len(ids), 1, buf[guard:].ctypes.data_as(ctypes.c_void_p), cap, None)
        assert rc == 0 or lib.wm_last_error()
        assert np.all(buf[:guard] == 0xa5) and np.all(buf[guard + cap:] == 0xa5), "wm_detokenize wrote outside [buf, buf + cap)"
    lib.wm_vocab_free(h)
    return 0


@FUZZ
@given(pieces=st.dictionaries(st.text(max_size=12), st.integers(-5, 70000), max_size=24), ids=st.lists(st.integers(-10, 70010), max_size=24),
       damage=st.integers(0, 3), cut=st.integers(0, 40))
def test_malformed_vocab_json_never_crashes(lib, tmp_path, pieces, ids, damage, cut):
    text = json.dumps(pieces, ensure_ascii=(damage == 1))
    if damage == 2:
        text = text.replace(":", "", 1).replace("\"", "", 1)
    if damage == 3:
        text = text[:max(0, len(text) - cut)]
    p = tmp_path / "vocab.json"
    p.write_bytes(text.encode("utf-8", "surrogatepass") if damage != 1 else text.encode())
    _try_vocab(lib, str(p), ids)


@FUZZ
@given(blob=st.binary(max_size=300), ids=st.lists(st.integers(-3, 300), max_size=8))
def test_random_bytes_as_vocab_never_crash(lib, tmp_path, blob, ids):
    p = tmp_path / "v.json"
    p.write_bytes(b"{" + blob)
    _try_vocab(lib, str(p), ids)


@FUZZ
@given(world=st.integers(1, 8), per=st.integers(1, 5), max_new=st.integers(1, 9), n_chunks=st.integers(0, 40),
       lens=st.lists(st.integers(-2 ** 31, 2 ** 31 - 1), min_size=40, max_size=40))
def test_token_payloads_with_hostile_lengths_never_overrun(lib, world, per, max_new, n_chunks, lens):
    """The payload of the token all-gather comes from other ranks: a length field outside [0, max_new] must not make the
    unpack read or write outside its arrays."""
    payload = np.zeros((world, per, 1 + max_new), dtype=np.int32)
    payload[..., 0] = np.resize(np.asarray(lens, dtype=np.int64), (world, per)).astype(np.int32)
    payload[..., 1:] = 7
    guard = 256
    n_out = max(n_chunks, 1)
    toks = np.full(n_out * max_new + 2 * guard, -77, dtype=np.int32)
    ln = np.full(n_out + 2 * guard, -77, dtype=np.int32)
    rc = lib.wm_multi_unpack_tokens(payload.ctypes.data_as(ctypes.c_void_p), world, per, max_new, n_chunks,
                                    toks[guard:].ctypes.data_as(ctypes.c_void_p), ln[guard:].ctypes.data_as(ctypes.c_void_p))
    assert np.all(toks[:guard] == -77) and np.all(toks[guard + n_chunks * max_new:] == -77)
    assert np.all(ln[:guard] == -77) and np.all(ln[guard + n_chunks:] == -77)
    if rc == 0:
        got = ln[guard:guard + n_chunks]
        assert np.all((got >= 0) & (got <= max_new)), got
