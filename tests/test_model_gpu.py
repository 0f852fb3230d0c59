"""GPU parity tests for boundary #2 (encoder / decoder / greedy decode) through the C ABI,
against the fp32 PyTorch oracle (oracle/whisper_ref.py) on identical synthetic weights.

Tolerances (stated here, as BASELINE.md asks): the HIP path computes GEMM inputs, K/V caches
and attention probabilities in bf16 with fp32 accumulation, fp32 LayerNorm / softmax /
residual stream.  Against the all-fp32 oracle that gives
    encoder output (post ln_post)  rel-L2 <= 5e-3   (measured 2.8e-4 at two layers .. 3.4e-3 at 32; per-site gates in TOL)
    teacher-forced logits          rel-L2 <= 1e-2   (measured 2.1e-3 .. 6.9e-3; per-site gates in TOL)
and the ALL-FP32 debug path (debug library, wmdbg_set_precision) <= 1e-4 as BASELINE.md requires (measured <= 2.2e-6)
and arg-max decisions are checked by margin: the token the GPU picks must be within
0.05 logit units of the oracle's maximum (exact equality whenever the oracle's top-2 gap is
larger than that)."""
import importlib
import os

import numpy as np
import pytest

from oracle import logmel_np as L
from oracle import whisper_ref as R

pytestmark = pytest.mark.gpu

W = importlib.import_module("openai_whisper_coreml_amd.weights")
ENC_TOL = 5e-3
LOGIT_TOL = 1e-2
MARGIN = 0.05
# Per-site gates (rel-L2 against the fp32 oracle).  Every gate is <= 3x what the site measured on an MI355X
# (profiles/r04_parity_margins_tests.txt, written by this module's own run: gate() records every measurement); ENC_TOL /
# LOGIT_TOL above are the envelope (the worst sites: 32-layer encoders 3.4e-3, lively full-depth logits 6.9e-3).
TOL = {   # name: gate                 measured on MI355X (round 4)   gate / measured
    "tiny.enc": 8e-4,               # 2.85e-04   2.8x
    "golden.enc_rows": 1e-3,        # 3.45e-04   2.9x
    "golden.logits_head": 1e-2,     # 3.34e-03   3.0x
    "tiny.logits": 1e-2,            # 4.38e-03   2.3x
    "tiny_en.enc": 3.3e-3,          # 1.14e-03   2.9x
    "tiny_en.logits": 1e-2,         # 3.88e-03   2.6x
    "large_v3_2layer.enc": 5e-3,    # 2.12e-03   2.4x
    "large_v3_2layer.logits": 1e-2,  # 3.61e-03   2.8x
    "large_v2_full.enc": 5e-3,      # 3.31e-03   1.5x
    "large_v2_full.logits": 1e-2,   # 4.01e-03   2.5x
    "tiny_lively.enc_silence": 5e-3,  # 2.81e-03   1.8x
    "small_full.enc": 5e-3,         # 2.66e-03   1.9x
    "small_full.logits_t1": 1e-2,   # 5.10e-03   2.0x
    "large_v2_full_b8.enc": 5e-3,   # 3.33e-03   1.5x
    "large_v3_full.enc": 5e-3,      # 3.36e-03   1.5x
    "large_v3_full.logits": 1e-2,   # 5.06e-03   2.0x
    "converted.enc": 8e-4,          # 2.82e-04   2.8x
    "converted.logits": 1e-2,       # 4.21e-03   2.4x
    "offset.logits": 1e-2,          # 4.35e-03 .. 4.68e-03 (offsets 0 / 3 / 10)   2.1x
    "offset_outlier.logits": 6e-3,  # 2.07e-03 (offset 3 + a massive-activation feature)   2.9x
    "policy.logits": 1e-2,          # 6.87e-03 / 6.49e-03 (lively, production vocabulary)   1.5x
    "workload.logits_all": 1e-2,    # 6.83e-03 (lively large-v2, full depth, 227 positions)   1.5x
    "workload.logits_tail": 1e-2,   # 6.79e-03   1.5x
    # round 6: configs[1] / configs[2] with the lively model of their own width, 224 tokens (gates = the envelope; the
    # measured values are appended to profiles/r06_parity_margins_tests.txt)
    "tiny_en_lively.enc": 5e-3, "tiny_en_lively.logits_all": 1e-2, "base_lively.logits_all": 1e-2, "base_lively.enc_rows": 5e-3,
    # the all-fp32 debug path: BASELINE.md's gate is 1e-4; measured 1.3e-07 .. 2.2e-06 (large-v2 full depth)
    "f32.enc": 7e-6, "f32.logits": 7e-6,
}
_MEASURED = []


def gate(name, value, tol=None):
    """assert-able: value <= tol (default TOL[name]); records the measurement (written to gpurun_out/ at module teardown)."""
    tol = TOL[name] if tol is None else tol
    _MEASURED.append((name, float(value), float(tol)))
    assert value <= tol, "%s: measured %.3e > gate %.3e" % (name, value, tol)
    return True


@pytest.fixture(scope="module", autouse=True)
def _write_measured_margins():
    yield
    from conftest import ROOT
    out = os.path.join(ROOT, "gpurun_out")
    if _MEASURED and os.path.isdir(out):
        with open(os.path.join(out, "parity_margins_tests.txt"), "w") as f:
            f.write("# tests/test_model_gpu.py: measured rel-L2 vs the fp32 oracle at every gated site (name, measured, gate, gate / measured)\n")
            for n, v, t in _MEASURED:
                f.write("%-36s %.3e  %.3e  %5.1fx\n" % (n, v, t, t / max(v, 1e-30)))


def nontrivial_ln(sd, seed=0):
    """Synthetic weights have LN gamma=1, beta=0; perturb so LN parameters are exercised."""
    rng = np.random.default_rng(seed)
    for k in sd:
        if "ln" in k and k.endswith("weight"):
            sd[k] = (1 + 0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
        if "ln" in k and k.endswith("bias"):
            sd[k] = (0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
    return sd


@pytest.fixture(scope="module")
def tiny(pkg):
    dims = dict(R.TINY_DIMS)
    sd_np = nontrivial_ln(W.synthetic_state_dict(dims, seed=11))
    ctx = pkg.binding.Context(dims)
    ctx.load_state_dict(sd_np)
    ctx.finalize()
    yield dims, sd_np, R.to_torch(sd_np), ctx
    ctx.close()


def mels(ctx, n, start=0):
    pcm = np.stack([L.synth_chunk(start + i) for i in range(n)])
    return pcm, ctx.logmel(pcm, out_dtype=np.float32)


LIVELY_GAIN = 4.0


@pytest.fixture(scope="module")
def lively(pkg):
    """The N(0, 0.02^2) synthetic weights make a nearly input-independent model (every chunk and position
    decodes to the same token), which is fine for value parity but blind for token-level tests.  Scaling every
    matrix by 4 (a power of two: still bf16-exact) and feeding amplitude-modulated tones gives token streams
    that depend on the audio AND on the decode history."""
    dims = dict(R.TINY_DIMS)
    sd_np = nontrivial_ln(W.synthetic_state_dict(dims, seed=11))
    for k in sd_np:
        if sd_np[k].ndim >= 2 and "positional" not in k:
            sd_np[k] = sd_np[k] * np.float32(LIVELY_GAIN)
    ctx = pkg.binding.Context(dims)
    ctx.load_state_dict(sd_np)
    ctx.finalize()
    yield dims, sd_np, R.to_torch(sd_np), ctx
    ctx.close()


def tone_chunk(i):
    n = np.arange(480000, dtype=np.float64)
    x = 0.3 * np.sin(2 * np.pi * (200 + 370 * i) * n / 16000) * (0.5 + 0.5 * np.sin(2 * np.pi * (0.3 + 0.1 * i) * n / 16000))
    return x.astype(np.float32)


def tones(n, start=0):
    return np.stack([tone_chunk(start + i) for i in range(n)])


@pytest.mark.parametrize("gain", [1.0, 4.0, 3.0])
def test_device_synthetic_generator_is_bit_identical(pkg, gain):
    """wm_init_synthetic / wm_init_synthetic_gain == weights.synthetic_state_dict on the host, bit for bit (gain 3 is not a
    power of two: the second rounding is part of the definition)."""
    dims = dict(R.TINY_DIMS)
    sd_np = W.synthetic_state_dict(dims, seed=5, matrix_gain=gain)
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(5, matrix_gain=gain)
    ctx.finalize()
    for name, shape, kind in W.tensor_specs(dims):
        got = ctx.get_tensor(name, shape)
        if kind == W.K_SINUSOID:
            assert np.abs(got - sd_np[name]).max() <= 1e-6, name
        else:
            assert np.array_equal(got, sd_np[name]), name
    if gain != 1.0:
        plain = W.synthetic_state_dict(dims, seed=5)
        assert np.array_equal(sd_np["decoder.positional_embedding"], plain["decoder.positional_embedding"])
        assert np.array_equal(sd_np["decoder.blocks.0.mlp.0.bias"], plain["decoder.blocks.0.mlp.0.bias"])
        assert np.abs(sd_np["decoder.blocks.0.mlp.0.weight"]).max() > 2 * np.abs(plain["decoder.blocks.0.mlp.0.weight"]).max()
    ctx.close()


def test_set_get_roundtrip_and_flat_file(pkg, tiny, tmp_path):
    dims, sd_np, _, ctx = tiny
    for name in ("encoder.conv1.weight", "encoder.conv2.weight", "decoder.blocks.1.cross_attn.key.weight",
                 "decoder.blocks.0.attn.value.bias", "decoder.token_embedding.weight"):
        got = ctx.get_tensor(name, sd_np[name].shape)
        want = W.bf16_round_f32(sd_np[name]) if name.endswith("weight") else sd_np[name]
        assert np.array_equal(got, want), name
    path = os.path.join(tmp_path, "tiny.wm")
    W.save_flat(path, dims, sd_np)
    d2, sd2 = W.load_flat(path)
    assert d2 == dims and all(np.array_equal(sd2[k], sd_np[k]) for k in sd_np)
    c2 = pkg.binding.Context(dims)
    c2.load_weights(path)
    c2.finalize()
    assert np.array_equal(c2.get_tensor("decoder.ln.weight", (128,)), sd_np["decoder.ln.weight"])
    c2.close()


def test_encoder_parity(tiny):
    dims, _, sd, ctx = tiny
    _, mel = mels(ctx, 2)
    got = ctx.encode_mel(mel)
    want = R.encode(sd, dims, mel).numpy()
    e = R.rel_l2(got, want)
    print("encoder rel-L2", e)
    assert got.shape == (2, 1500, 128) and gate("tiny.enc", e)


def test_committed_model_golden(pkg):
    """tests/golden/model_golden.npz (fp32 oracle outputs) vs the HIP path on device-generated weights."""
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "model_golden.npz"))
    dims = dict(R.TINY_DIMS)
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(int(g["seed"]))
    ctx.finalize()
    mel = g["mel"].astype(np.float32)[None]
    xa = ctx.encode_mel(mel)
    assert gate("golden.enc_rows", R.rel_l2(xa[0, g["rows"]], g["xa_rows"]))
    lg = ctx.decode_logits(g["tokens"][None], xa)
    assert gate("golden.logits_head", R.rel_l2(lg[0][:, :64], g["logits_head"]))
    ctx.close()


def test_whole_model_256_tile_is_bitwise_equal_to_the_128_tile(pkg):
    """Every epilogue of the 256 x 256 encoder GEMM (row-contiguous stores straight from C^T accumulators, permuted W
    staging, the V^T waves of the QKV projection that keep the other operand order, conv2 + positional embedding,
    the cross-K/V scatter) against the 128 x 128 kernel on a whole model: both accumulate every dot product in the
    same order, so encoder output, teacher-forced logits (through the cross-K/V cache) and greedy tokens must be
    identical bit for bit."""
    import ctypes
    dims = dict(R.TINY_DIMS)
    ctx = pkg.binding.Context(dims, debug=True)
    ctx.lib.wmdbg_set_gemm_tile.argtypes = [ctypes.c_int]
    try:
        ctx.init_synthetic(23)
        ctx.finalize()
        pcm = tones(3)
        mel = ctx.logmel(pcm, out_dtype=np.float32)
        toks = np.array([[1, 7, 300, 1023], [4, 4, 900, 17], [9, 2, 2, 511]], np.int32)
        outs = []
        ctx.lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        # 64: the single-chunk tile of round 6; (128, pipe 2): the 3-stage pipeline of the 128 tile; 0: the product's own choice
        for tile, pipe in ((128, 1), (256, 0), (64, 0), (128, 2), (0, 0)):
            assert ctx.lib.wmdbg_set_gemm_tile(tile) == 0
            assert ctx.lib.wmdbg_set_tuning(b"gemm128_pipe", pipe) == 0
            xa = ctx.encode_mel(mel)
            lg = ctx.decode_logits(toks, xa)
            gen, lens = ctx.transcribe_greedy(pcm, [1, 2], 6)
            outs.append((xa, lg, gen))
        assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][0]).max() > 0
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert np.array_equal(a, b)
    finally:
        ctx.lib.wmdbg_set_gemm_tile(0)
        ctx.lib.wmdbg_set_tuning(b"reset", 0)
        ctx.close()


def test_encoder_batch_independence(tiny):
    dims, _, _, ctx = tiny
    _, mel = mels(ctx, 3)
    all3 = ctx.encode_mel(mel)
    one = ctx.encode_mel(mel[1:2])
    assert np.array_equal(all3[1], one[0])   # chunks are independent units: bit-identical


def test_decode_logits_parity(tiny):
    dims, _, sd, ctx = tiny
    _, mel = mels(ctx, 2)
    xa = R.encode(sd, dims, mel).numpy()     # same fp32 xa for both sides
    toks = np.array([[1, 7, 300, 1023, 5, 9], [4, 4, 900, 17, 0, 511]], np.int32)
    got = ctx.decode_logits(toks, xa)
    want = R.decode_logits(sd, dims, toks, xa).numpy()
    e = R.rel_l2(got, want)
    print("logits rel-L2", e, "std", want.std())
    assert got.shape == (2, 6, 1024) and gate("tiny.logits", e)
    # T = 1 is the reference's exported decoder shape (whisper_to_cml.py:28)
    got1 = ctx.decode_logits(toks[:, :1], xa)
    assert np.array_equal(got1[:, 0], got[:, 0])


def _check_choice(ref_logits_row, gpu_choice):
    gap = float(ref_logits_row.max() - ref_logits_row[gpu_choice])
    assert gap <= MARGIN, "GPU picked %d, oracle gap %g" % (gpu_choice, gap)


def test_detect_language_mirrors_swift(tiny):
    """Whisper.swift:33-40 with the tiny vocabulary: SOT=10, languages 20..118 (99 ids)."""
    dims, _, sd, ctx = tiny
    _, mel = mels(ctx, 3)
    xa = R.encode(sd, dims, mel).numpy()
    got = ctx.detect_language(xa, sot=10, lang_first=20, lang_last=118)
    _, conf = R.detect_language(sd, dims, xa, sot=10, lang_first=20, lang_last=118)
    assert got.shape == (3,) and got.min() >= 0 and got.max() <= 98
    for b in range(3):
        _check_choice(conf[b], int(got[b]))
    # openai-whisper detect_language(): probabilities = softmax over the language-token logits only
    idx, probs = ctx.detect_language_probs(xa, sot=10, lang_first=20, lang_last=118)
    assert np.array_equal(idx, got) and probs.shape == (3, 99)
    assert np.allclose(probs.sum(axis=1), 1.0, atol=1e-5) and np.array_equal(probs.argmax(axis=1), idx)
    ex = np.exp(conf - conf.max(axis=1, keepdims=True))
    want = ex / ex.sum(axis=1, keepdims=True)
    assert np.abs(probs - want).max() <= 2e-3, np.abs(probs - want).max()


def test_greedy_transcribe_end_to_end(tiny):
    dims, _, sd, ctx = tiny
    pcm, mel = mels(ctx, 2, start=3)
    prompt = [10, 21, 3]
    toks, lens = ctx.transcribe_greedy(pcm, prompt, 12, eot=-1)
    assert toks.shape == (2, 12) and lens.tolist() == [12, 12]
    xa = ctx.encode_mel(mel)                          # the GPU's own encoder output
    for b in range(2):
        seq = np.concatenate([prompt, toks[b]])[None, :-1]
        ref = R.decode_logits(sd, dims, seq, xa[b:b + 1]).numpy()[0]
        for i in range(12):                           # every greedy choice, teacher-forced on the GPU's history
            _check_choice(ref[len(prompt) - 1 + i], int(toks[b, i]))
    # int16 PCM input gives the same tokens as the float path modulo quantisation of the audio
    s16 = np.round(pcm * 32767).astype(np.int16)
    toks16, _ = ctx.transcribe_greedy(s16, prompt, 4, eot=-1)
    assert toks16.shape == (2, 4)
    # eot handling: stop at the first generated token equal to eot, pad with eot
    eot = int(toks[0, 2])
    t2, l2 = ctx.transcribe_greedy(pcm, prompt, 12, eot=eot)
    first = int(np.argmax(toks[0] == eot))
    assert l2[0] == first + 1 and (t2[0, first:] == eot).all() and (t2[0, :first] == toks[0, :first]).all()


def test_greedy_is_deterministic_and_batch_invariant(tiny):
    dims, _, _, ctx = tiny
    pcm, _ = mels(ctx, 3, start=6)
    a, _ = ctx.transcribe_greedy(pcm, [10], 8)
    b, _ = ctx.transcribe_greedy(pcm, [10], 8)
    c, _ = ctx.transcribe_greedy(pcm[1:2], [10], 8)
    assert np.array_equal(a, b) and np.array_equal(a[1], c[0])


def test_swift_surface_mirror(pkg):
    """struct Whisper (Whisper.swift:11-41): init -> encode(audio) -> decode(features)."""
    dims = dict(R.TINY_DIMS, n_vocab=51865)
    sd_np = W.synthetic_state_dict(dims, seed=2)
    wh = pkg.Whisper(dims, state_dict=sd_np)
    audio = np.zeros(480000, np.float64)
    x = L.synth_chunk(1)
    audio[:160000] = x[:160000]                       # ContentView.swift:57-60: 10 s, zero-padded
    enc = wh.encode(audio)
    assert enc.shape == (1, 1500, 128)
    lang = wh.decode(enc)
    assert lang in pkg.Whisper.LANGUAGES and len(pkg.Whisper.LANGUAGES) == 99
    sd = R.to_torch(sd_np)
    idx, conf = R.detect_language(sd, dims, enc)
    _check_choice(conf[0], pkg.Whisper.LANGUAGES.index(lang))
    wh.ctx.close()


def test_tiny_en_dimensions(pkg):
    """BASELINE.json configs[1] geometry (tiny.en: d 384, 6 heads, 4+4 layers, vocab 51864)."""
    dims = pkg.binding.MODEL_DIMS["tiny.en"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(7)
    ctx.finalize()
    sd = R.to_torch({n: ctx.get_tensor(n, s) for n, s, _ in W.tensor_specs(dims)})
    pcm, mel = mels(ctx, 1)
    got = ctx.encode_mel(mel)
    want = R.encode(sd, dims, mel).numpy()
    e = R.rel_l2(got, want)
    print("tiny.en encoder rel-L2", e)
    assert gate("tiny_en.enc", e)
    toks = np.array([[50257, 50362, 100, 2000]], np.int32)
    lg = ctx.decode_logits(toks, want)
    ref = R.decode_logits(sd, dims, toks, want).numpy()
    e2 = R.rel_l2(lg, ref)
    print("tiny.en logits rel-L2", e2)
    assert gate("tiny_en.logits", e2)
    # BASELINE.json configs[1] as written: greedy decode of a single chunk (the flat cross-attention launch + combine,
    # 6 heads x 1 sequence), every choice against the oracle teacher-forced on the GPU's prefix
    prompt = [50257, 50362]
    gen, lens = ctx.transcribe_greedy(pcm, prompt, 8)
    assert gen.shape == (1, 8) and lens.tolist() == [8]
    _check_greedy_against_teacher_forced_oracle(sd, dims, want, prompt, gen)
    ctx.close()


def test_tiny_en_single_chunk_full_length_lively(pkg):
    """BASELINE.json configs[1] at the standard of the large-v2 / large-v3 tests: tiny.en, ONE 30 s chunk,
    the lively random-init model of THIS width (weights.lively_gain: 6 at d = 384) with perturbed LayerNorms, 224 new tokens
    (n_text_ctx // 2) through the product's own path (burst graphs, the flat cross-attention deal of 6 pairs) -- EVERY one of
    the 224 choices teacher-forced against the fp32 oracle; two different recordings give different, history-dependent rows."""
    dims = pkg.binding.MODEL_DIMS["tiny.en"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(20240928, matrix_gain=W.lively_gain(dims))
    _perturb_ln_on_device(ctx, dims, seed=21)
    ctx.finalize()
    sd = _oracle_weights(ctx, dims)
    prompt = [50257, 50362]
    NEW = dims["n_text_ctx"] // 2
    pcm = np.stack([L.synth_chunk(71), tone_chunk(3)])
    rows = []
    for i in range(2):                                   # one chunk per call: the configuration as written
        t, l = ctx.transcribe_greedy(pcm[i:i + 1], prompt, NEW, eot=-1)
        assert t.shape == (1, NEW) and l.tolist() == [NEW]
        rows.append(t[0])
    rows = np.stack(rows)
    assert not np.array_equal(rows[0], rows[1])
    changes = [int((r[1:] != r[:-1]).sum()) for r in rows]
    print("tiny.en lively: token changes per row", changes, "distinct tokens", [len(set(r.tolist())) for r in rows])
    assert max(changes) >= 16, changes
    both, _ = ctx.transcribe_greedy(pcm, prompt, NEW, eot=-1)     # ... and the two chunks as one group: same rows
    assert np.array_equal(both, rows)
    mel = ctx.logmel(pcm)
    xa = ctx.encode_mel(mel)
    want = R.encode(sd, dims, mel).numpy()
    assert gate("tiny_en_lively.enc", R.rel_l2(xa, want))
    worst = _check_greedy_against_teacher_forced_oracle(sd, dims, want, prompt, rows, scaled=True)
    seq = np.concatenate([np.tile(np.asarray(prompt, np.int32), (2, 1)), rows], axis=1)[:, :-1].astype(np.int32)
    ref = R.decode_logits(sd, dims, seq, want).numpy()
    got = ctx.decode_logits(seq, want)
    e = R.rel_l2(got, ref)
    print("tiny.en lively, 224 tokens: worst greedy gap %.3g logit (rms %.2f), teacher-forced logits rel-L2 %.3e"
          % (worst, float(np.sqrt((ref.astype(np.float64) ** 2).mean())), e))
    assert gate("tiny_en_lively.logits_all", e)
    ctx.close()


def test_error_paths(pkg):
    dims = dict(R.TINY_DIMS)
    ctx = pkg.binding.Context(dims)
    with pytest.raises(pkg.binding.WhisperError, match="never set"):
        ctx.finalize()
    with pytest.raises(pkg.binding.WhisperError, match="unknown tensor"):
        ctx.set_tensor("encoder.nope", np.zeros(4, np.float32))
    with pytest.raises(pkg.binding.WhisperError, match="expected"):
        ctx.set_tensor("encoder.conv1.bias", np.zeros(4, np.float32))
    with pytest.raises(pkg.binding.WhisperError, match="not finalised"):
        ctx.encode_mel(np.zeros((1, 80, 3000), np.float32))
    ctx.init_synthetic(1)
    ctx.finalize()
    xa = np.zeros((1, 1500, 128), np.float32)
    with pytest.raises(pkg.binding.WhisperError, match="token id"):
        ctx.decode_logits(np.array([[5000]], np.int32), xa)
    with pytest.raises(pkg.binding.WhisperError, match="T must be"):
        ctx.decode_logits(np.zeros((1, 449), np.int32), xa)
    with pytest.raises(pkg.binding.WhisperError, match="context"):
        ctx.transcribe_greedy(np.zeros((1, 480000), np.float32), [1, 2], 447)
    ctx.close()
    fe = pkg.binding.Context()
    with pytest.raises(pkg.binding.WhisperError, match="without a model"):
        fe.finalize()
    fe.close()
    with pytest.raises(pkg.binding.WhisperError):
        pkg.binding.Context(dict(dims, n_audio_state=100))


def test_cpp_host_harness_mirrors_the_swift_flow(pkg):
    """host/lid_main.cpp dlopens the .so and runs ContentView.swift:56-63 -> Whisper.swift:23-40."""
    import subprocess
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "openai-whisper-coreml_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_host()
    r = subprocess.run([exe, pkg.binding.LIB_PATH, "base", "synthetic:3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] in pkg.Whisper.LANGUAGES and float(lines[1]) > 0
    # the same flow from query.wav (AudioRecorder.swift:56-61's format) through the library's own WAV reader: identical
    # language to feeding the same samples through the Python mirror of the Swift surface
    import importlib
    import tempfile
    A = importlib.import_module("openai_whisper_coreml_amd.audio")
    x16 = np.round(tone_chunk(3)[:160000] * 32767).astype(np.int16)          # 10 s, as the app records
    with tempfile.TemporaryDirectory() as td:
        wav = os.path.join(td, "query.wav")
        A.write_wav_int16(wav, x16)
        r2 = subprocess.run([exe, pkg.binding.LIB_PATH, "base", "synthetic:3", wav], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr
    wh = pkg.Whisper("base", synthetic_seed=3)
    audio = np.zeros(480000, np.float64)
    audio[:160000] = (x16.astype(np.float32) / np.float32(32768.0)).astype(np.float64)
    assert r2.stdout.strip().splitlines()[0] == wh.decode(wh.encode(audio))
    wh.ctx.close()


def test_base_geometry_batch32(pkg):
    """BASELINE.json configs[2]: base multilingual, batch 32, at the standard of the large-v2 / large-v3 tests (VERDICT r5
    next #4): 32 DISTINCT recordings (tones + seeded noise), the lively random-init model of THIS width (weights.lively_gain:
    6 at d = 512 -- the gain-4 recipe gave 4 distinct rows of 32) with perturbed LayerNorms, 224 new tokens, the product's own
    group policy (round 6: two CU-masked half-chip groups of 16); at least 30 of the 32 rows are distinct; rows 0 / 15 / 16 / 31 -- the edges of
    both groups -- equal the same chunk decoded alone; the encoder rows and ALL 224 choices of three rows are checked
    against the fp32 oracle, teacher-forced on the GPU's own prefix."""
    import torch
    dims = pkg.binding.MODEL_DIMS["base"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(3, matrix_gain=W.lively_gain(dims))
    _perturb_ln_on_device(ctx, dims, seed=5)
    ctx.finalize()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    pcm = np.stack([tone_chunk(i) if i % 2 else L.synth_chunk(200 + i) for i in range(32)])
    mel = ctx.logmel(pcm)
    xa = ctx.encode_mel(mel)
    assert xa.shape == (32, 1500, 512)
    sd = _oracle_weights(ctx, dims)
    prompt = [50258, 50259, 50359, 50363]
    NEW = dims["n_text_ctx"] // 2
    toks, lens = ctx.transcribe_greedy(pcm, prompt, NEW, eot=-1)           # product policy: 2 x 16
    assert toks.shape == (32, NEW) and np.all(lens == NEW)
    # (a random-init model may drive two recordings into the same short token cycle: measured 31 - 32 distinct rows of 32 at
    # this gain, 3 - 4 of 32 at the d = 1280 gain of rounds 2-5 -- the point is that the cross-checks below are not blind)
    n_distinct = len({r.tobytes() for r in toks})
    assert n_distinct >= 30, "32 distinct recordings decode to only %d distinct rows" % n_distinct
    changes = [int((r[1:] != r[:-1]).sum()) for r in toks]
    print("base lively x 32: token changes per row min / median / max", min(changes), int(np.median(changes)), max(changes))
    assert sum(c >= 16 for c in changes) >= 16, changes
    for i in (0, 15, 16, 31):
        solo, _ = ctx.transcribe_greedy(pcm[i:i + 1], prompt, NEW, eot=-1)
        assert np.array_equal(solo[0], toks[i]), "row %d in the batch of 32 differs from the chunk decoded alone" % i
    ctx.set_lanes(1)                                                      # ... and from ONE group of 32 rows (two batch blocks)
    one, _ = ctx.transcribe_greedy(pcm, prompt, NEW, eot=-1)
    ctx.set_lanes(0)
    assert np.array_equal(one, toks)
    # the product policy at this width and size is two CU-MASKED half-chip groups (round 6, wm_lane_parts): early stop on
    # them == decode everything and truncate
    bud = [int(b) for b in np.random.default_rng(9).integers(20, 200, size=32)]
    t_es, l_es = ctx.transcribe_greedy(pcm, prompt, NEW, eot=-1, budgets=bud)
    assert list(l_es) == bud
    for i in range(32):
        assert np.array_equal(t_es[i, :bud[i]], toks[i, :bud[i]])
    pick = sorted({0, 17, int(np.argmax(changes))} | {31})[:3]
    want = R.encode(sd, dims, mel[pick]).numpy()
    assert gate("base_lively.enc_rows", R.rel_l2(xa[pick], want))
    worst = _check_greedy_against_teacher_forced_oracle(sd, dims, want, prompt, toks[pick], scaled=True)
    seq = np.concatenate([np.tile(np.asarray(prompt, np.int32), (1, 1)), toks[pick[:1]]], axis=1)[:, :-1].astype(np.int32)
    ref = R.decode_logits(sd, dims, seq, want[:1]).numpy()
    got = ctx.decode_logits(seq, want[:1])
    e = R.rel_l2(got, ref)
    print("base lively x 32, 224 tokens: worst greedy gap %.3g logit (rms %.2f), teacher-forced logits rel-L2 %.3e"
          % (worst, float(np.sqrt((ref.astype(np.float64) ** 2).mean())), e))
    assert gate("base_lively.logits_all", e)
    ctx.close()


def test_large_v3_front_end_and_vocabulary(pkg):
    """BASELINE.json configs[4] geometry, depth cut to 2+2 layers so the oracle stays fast: 128 mel
    bins (conv1 K = 384), vocabulary 51866, 100 language ids 50259...50358."""
    dims = dict(pkg.binding.MODEL_DIMS["large-v3"], n_audio_layer=2, n_text_layer=2)
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(5)
    ctx.finalize()
    pcm = np.stack([L.synth_chunk(11), L.synth_chunk(12)])
    mel = ctx.logmel(pcm, n_mels=128)
    assert mel.shape == (2, 128, 3000)
    sd = R.to_torch({n: ctx.get_tensor(n, s) for n, s, _ in W.tensor_specs(dims)})
    xa = ctx.encode_mel(mel)
    want = R.encode(sd, dims, mel).numpy()
    e = R.rel_l2(xa, want)
    print("large-v3 (2 layers) encoder rel-L2", e)
    assert gate("large_v3_2layer.enc", e)
    got = ctx.detect_language(want, sot=50258, lang_first=50259, lang_last=50358)
    _, conf = R.detect_language(sd, dims, want, sot=50258, lang_first=50259, lang_last=50358)
    assert conf.shape == (2, 100)
    for b in range(2):
        _check_choice(conf[b], int(got[b]))
    toks, _ = ctx.transcribe_greedy(pcm, [50258, 50259, 50360, 50364], 3)
    assert toks.shape == (2, 3) and toks.max() < 51866
    # decoder at the real width (d = 1280, 20 heads, K = 5120 MLP, 51 866-row tied embedding): teacher-forced logits
    tok = np.array([[50258, 50259, 50360], [50258, 50300, 50364]], dtype=np.int32)
    got_l = ctx.decode_logits(tok, want)
    ref_l = R.decode_logits(sd, dims, tok, want).numpy()
    e = R.rel_l2(got_l, ref_l)
    print("large-v3 (2 layers) logits rel-L2", e)
    assert gate("large_v3_2layer.logits", e)
    # ... and a decode group of 20 sequences (two batch blocks at this width) equals the same chunks alone
    idx = [0, 1] * 10
    many, _ = ctx.transcribe_greedy(pcm[idx], [50258, 50259, 50360, 50364], 3)
    assert np.array_equal(many, toks[idx])
    ctx.close()


def test_wav_to_tokens_pipeline(pkg, tiny, tmp_path):
    """8f rank 1: WAV (int16 RIFF) -> 30 s chunks -> greedy tokens, equal to feeding the chunks directly."""
    import importlib
    A = importlib.import_module("openai_whisper_coreml_amd.audio")
    dims, _, _, ctx = tiny
    x = np.concatenate([np.round(L.synth_chunk(1) * 32767), np.round(L.synth_chunk(2)[:100000] * 32767)]).astype(np.int16)
    p = os.path.join(tmp_path, "rec.wav")
    A.write_wav_int16(p, x)
    chunks = A.wav_to_chunks(p)
    assert chunks.shape == (2, 480000)
    toks, lens = A.transcribe_chunks(ctx, chunks, [10, 21], 5)
    direct, _ = ctx.transcribe_greedy(chunks, [10, 21], 5)
    assert np.array_equal(toks, direct) and lens.tolist() == [5, 5]


def test_cloned_contexts_overlap_and_agree(pkg, tiny):
    """wm_clone: weight-sharing contexts driven from concurrent host threads give the parent's exact tokens."""
    import threading
    dims, _, _, ctx = tiny
    pcm, _ = mels(ctx, 2, start=9)
    want, _ = ctx.transcribe_greedy(pcm, [10, 21], 10)
    clones = [ctx.clone() for _ in range(2)]
    got = [None] * 3

    def run(i, c):
        for _ in range(3):
            got[i], _ = c.transcribe_greedy(pcm, [10, 21], 10)

    th = [threading.Thread(target=run, args=(i, c)) for i, c in enumerate([ctx] + clones)]
    [t.start() for t in th]
    [t.join() for t in th]
    for g in got:
        assert np.array_equal(g, want)
    with pytest.raises(pkg.binding.WhisperError, match="shares"):
        clones[0].set_tensor("decoder.ln.bias", np.zeros(128, np.float32))
    assert np.array_equal(clones[1].get_tensor("decoder.ln.weight", (128,)), ctx.get_tensor("decoder.ln.weight", (128,)))
    for c in clones:
        c.close()


def test_lively_greedy_follows_the_oracle(lively):
    """Token-level parity on a model whose output depends on audio and history (see `lively`)."""
    dims, _, sd, ctx = lively
    pcm = tones(4)
    mel = ctx.logmel(pcm, out_dtype=np.float32)
    prompt = [10, 21, 5]
    toks, lens = ctx.transcribe_greedy(pcm, prompt, 16, eot=-1)
    assert len({tuple(r) for r in toks}) >= 3, toks             # audio-dependent
    assert all(len(set(r.tolist())) >= 2 for r in toks), toks   # history-dependent
    xa = ctx.encode_mel(mel)
    n_free = 0
    for b in range(4):
        seq = np.concatenate([prompt, toks[b]])[None, :-1]
        ref = R.decode_logits(sd, dims, seq, xa[b:b + 1]).numpy()[0]
        for i in range(16):
            row = ref[len(prompt) - 1 + i]
            _check_choice(row, int(toks[b, i]))
            n_free += int(np.argmax(row)) == int(toks[b, i])
    assert n_free >= 60, n_free                                  # essentially every choice is the oracle's arg-max
    # full free-running comparison against the oracle's own encoder + greedy loop, up to the first near-tie
    want, _, logits = R.greedy(sd, dims, R.encode(sd, dims, mel), prompt, 16)
    for b in range(4):
        for i in range(16):
            top2 = np.sort(logits[b, i])[-2:]
            if top2[1] - top2[0] < MARGIN:
                break
            assert toks[b, i] == want[b, i], (b, i)
    # batch invariance, bit-level: a chunk decodes to the same tokens alone, in a pair, or in the batch of 4
    for b in range(4):
        solo, _ = ctx.transcribe_greedy(pcm[b:b + 1], prompt, 16)
        assert np.array_equal(solo[0], toks[b]), b
    pair, _ = ctx.transcribe_greedy(pcm[2:4], prompt, 16)
    assert np.array_equal(pair, toks[2:4])


def test_large_call_is_spread_over_lanes(lively):
    """wm_transcribe_greedy with more chunks than one decode group runs balanced groups on concurrent
    weight-sharing lanes; chunks are independent, so every chunk must decode exactly as it does alone."""
    dims, _, _, ctx = lively
    base = tones(7)
    prompt = [10, 21, 5]
    want, _ = ctx.transcribe_greedy(base, prompt, 12)
    assert len({tuple(r) for r in want}) >= 5, want
    ctx.set_lanes(3)                             # an explicit lane count: groups of ~8 as soon as there are 8 chunks for each
    try:
        B = 19                                       # 3 lanes -> groups of 7, 6, 6
        idx = [(5 * i + 3) % 7 for i in range(B)]
        got, lens = ctx.transcribe_greedy(base[idx], prompt, 12)
        assert got.shape == (B, 12) and np.all(lens == 12)
        assert np.array_equal(got, want[idx])
        # more groups than lanes: 52 chunks -> 6 groups of 9/8 (two waves of 3 lanes)
        idx2 = [(3 * i + 1) % 7 for i in range(52)]
        got2, _ = ctx.transcribe_greedy(base[idx2], prompt, 12)
        assert np.array_equal(got2, want[idx2])
        # int16 PCM from host memory goes through the per-lane staging buffers
        s16 = np.round(base[idx] * 32767).astype(np.int16)
        got16, _ = ctx.transcribe_greedy(s16, prompt, 12)
        solo16, _ = ctx.transcribe_greedy(s16[:7], prompt, 12)
        assert np.array_equal(got16[:7], solo16) and np.array_equal(got16[7:13], ctx.transcribe_greedy(s16[7:13], prompt, 12)[0])
    finally:
        ctx.set_lanes(0)
    # the library's own policy (round 5: one group below 32 chunks, two up to 143, three from 144): same rows again
    for n in (19, 52, 150):
        idx3 = [(2 * i + n) % 7 for i in range(n)]
        got3, _ = ctx.transcribe_greedy(base[idx3], prompt, 12)
        assert np.array_equal(got3, want[idx3]), n


def test_suppress_filters_follow_the_oracle(lively, pkg):
    """wm_set_suppress == openai-whisper's SuppressTokens + SuppressBlank inside the fused logits / arg-max kernel."""
    dims, _, sd, ctx = lively
    pcm = tones(3)
    mel = ctx.logmel(pcm, out_dtype=np.float32)
    prompt = [10, 21, 5]
    free, _ = ctx.transcribe_greedy(pcm, prompt, 12)
    # forbid everything the unfiltered run produced, plus the runner-up of its first choice at the first position only
    banned = sorted({int(t) for t in free.ravel()})
    xa = ctx.encode_mel(mel)
    first_logits = R.decode_logits(sd, dims, np.tile(prompt, (3, 1)), xa).numpy()[:, -1]
    first_logits[:, banned] = -np.inf
    banned_first = sorted({int(np.argmax(r)) for r in first_logits})
    ctx.set_suppress(banned, banned_first)
    try:
        got, lens = ctx.transcribe_greedy(pcm, prompt, 12)
        assert not (set(got.ravel().tolist()) & set(banned))
        assert not (set(got[:, 0].tolist()) & set(banned_first))
        want, _, logits = R.greedy(sd, dims, R.encode(sd, dims, mel), prompt, 12, suppress=banned, suppress_first=banned_first)
        for b in range(3):
            seq = np.concatenate([prompt, got[b]])[None, :-1]
            ref = R.decode_logits(sd, dims, seq, xa[b:b + 1]).numpy()[0]
            for i in range(12):
                row = ref[len(prompt) - 1 + i].copy()
                row[banned] = -np.inf
                if i == 0:
                    row[banned_first] = -np.inf
                _check_choice(row, int(got[b, i]))
            for i in range(12):                       # free-running agreement up to the first near-tie
                top2 = np.sort(logits[b, i][np.isfinite(logits[b, i])])[-2:]
                if top2[1] - top2[0] < MARGIN:
                    break
                assert got[b, i] == want[b, i], (b, i)
        # a 19-chunk call runs on the lanes: they carry the same filter
        idx = [i % 3 for i in range(19)]
        many, _ = ctx.transcribe_greedy(pcm[idx], prompt, 12)
        assert np.array_equal(many, got[idx])
        with pytest.raises(pkg.binding.WhisperError, match="outside"):
            ctx.set_suppress([dims["n_vocab"]], [])
    finally:
        ctx.set_suppress([], [])
    again, _ = ctx.transcribe_greedy(pcm, prompt, 12)
    assert np.array_equal(again, free)


def test_decode_groups_above_sixteen_use_two_batch_blocks(lively):
    """A decode group of 17..64 chunks (up to 128) runs every skinny GEMM as blocks of 16 batch rows in one launch; every
    chunk must still decode exactly as it does alone."""
    dims, _, _, ctx = lively
    base = tones(7)
    prompt = [10, 21, 5]
    want, _ = ctx.transcribe_greedy(base, prompt, 12)
    for B in (17, 32):
        idx = [(3 * i + 2) % 7 for i in range(B)]
        got = ctx.detect_language(ctx.encode_mel(ctx.logmel(base[idx], out_dtype=np.float32)), sot=10, lang_first=20, lang_last=118)
        one = ctx.detect_language(ctx.encode_mel(ctx.logmel(base, out_dtype=np.float32)), sot=10, lang_first=20, lang_last=118)
        assert np.array_equal(got, one[idx]), B           # wm_detect_language at B > 16 (one decode step)
    for n in (90, 150, 330):                              # 3 lanes x 30 (two batch blocks), 3 x 50 (four), 3 x 110 (seven)
        idx = [(5 * i + 1) % 7 for i in range(n)]
        got, lens = ctx.transcribe_greedy(base[idx], prompt, 12)
        assert np.array_equal(got, want[idx]) and np.all(lens == 12), n


def test_timestamp_rules_follow_the_oracle(lively, pkg):
    """wm_set_timestamp_rules == openai-whisper's ApplyTimestampRules inside the fused logits / arg-max kernels:
    per-sequence admissible ranges (pairs, monotonic, initial timestamp) + the summed-probability rule."""
    import torch
    dims, _, sd, ctx = lively
    TS, EOT, MAXI, NEW = 900, 890, 20, 24            # tiny vocabulary (1024): timestamps 900..1023, specials 891..899
    pcm = tones(4)
    mel = ctx.logmel(pcm, out_dtype=np.float32)
    prompt = [10, 21, 5]
    specials = list(range(EOT + 1, TS))
    ctx.set_suppress(specials, [EOT])
    ctx.set_timestamp_rules(True, TS, EOT, MAXI)
    try:
        got, lens = ctx.transcribe_greedy(pcm, prompt, NEW)
        # structure: opens with a timestamp <= TS + MAXI; timestamps never decrease; never three timestamps in a row
        assert np.all(got[:, 0] >= TS) and np.all(got[:, 0] <= TS + MAXI)
        for b in range(4):
            ts = [t for t in got[b] if t >= TS]
            assert all(x <= y for x, y in zip(ts, ts[1:])), (b, ts)
            isT = [t >= TS for t in got[b]]
            assert not any(isT[i] and isT[i + 1] and isT[i + 2] for i in range(NEW - 2)), (b, got[b])
            assert not (set(got[b].tolist()) & set(specials))
        assert any((got[b] < TS).any() for b in range(4)) and any((got[b, 1:] >= TS).any() for b in range(4))
        # every choice against the oracle's filtered logits, teacher-forced on the GPU's own history
        xa = ctx.encode_mel(mel)
        n_forced = 0
        for b in range(4):
            seq = np.concatenate([prompt, got[b]])[None, :-1]
            ref = R.decode_logits(sd, dims, seq, xa[b:b + 1])[0]
            for i in range(NEW):
                row = ref[len(prompt) - 1 + i].clone()
                row[specials] = float("-inf")
                if i == 0:
                    row[EOT] = float("-inf")
                unforced = row.clone()
                forced, gap = R.timestamp_filter(row, [int(t) for t in got[b, :i]], TS, EOT, MAXI)
                n_forced += forced
                choice = int(got[b, i])
                if abs(gap) < MARGIN:            # the summed-probability rule is a near-tie: either branch is right
                    R.timestamp_filter(unforced, [int(t) for t in got[b, :i]], TS, EOT, MAXI, sum_rule=False)
                    only_ts = unforced.clone()
                    only_ts[:TS] = float("-inf")
                    ok = float(only_ts.max() - only_ts[choice]) <= MARGIN or float(unforced.max() - unforced[choice]) <= MARGIN
                    assert ok, (b, i, choice)
                else:
                    _check_choice(row.numpy(), choice)
        assert n_forced > 0
        # free-running oracle (its own encoder, its own history) up to the first near-tie
        want, _, logits = R.greedy(sd, dims, R.encode(sd, dims, mel), prompt, NEW, suppress=specials, suppress_first=[EOT],
                                   ts_rules=dict(ts_begin=TS, eot=EOT, max_initial=MAXI))
        agree = 0
        for b in range(4):
            for i in range(NEW):
                fin = logits[b, i][np.isfinite(logits[b, i])]
                top2 = np.sort(fin)[-2:] if fin.size >= 2 else np.array([0.0, 1.0])
                if top2[1] - top2[0] < MARGIN or got[b, i] != want[b, i]:
                    break
                agree += 1
        assert agree >= 8, agree
        # lanes carry the rules and their per-sequence state
        idx = [i % 4 for i in range(19)]
        many, _ = ctx.transcribe_greedy(pcm[idx], prompt, NEW)
        assert np.array_equal(many, got[idx])
        with pytest.raises(pkg.binding.WhisperError, match="timestamp"):
            ctx.set_timestamp_rules(True, 5, 7, -1)
    finally:
        ctx.set_timestamp_rules(False)
        ctx.set_suppress([], [])


def test_full_depth_large_v2_against_the_oracle(pkg):
    """BASELINE.json configs[3] at full depth (32 + 32 layers, d = 1280, 51 865-row tied embedding), one chunk:
    encoder output and teacher-forced decoder logits against the torch-fp32 oracle on the GPU's own (bf16-rounded)
    weights.  The error grows with depth (tiny 2-layer models: 3e-4 .. 1e-3) and stays inside the stated tolerances."""
    import torch
    dims = pkg.binding.MODEL_DIMS["large-v2"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(7)
    ctx.finalize()
    pcm = np.stack([L.synth_chunk(21)])
    mel = ctx.logmel(pcm)
    xa = ctx.encode_mel(mel)
    sd = R.to_torch({n: ctx.get_tensor(n, s) for n, s, _ in W.tensor_specs(dims)})
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    want = R.encode(sd, dims, mel).numpy()
    e_enc = R.rel_l2(xa, want)
    tok = np.array([[50258, 50259, 50359, 50363]], dtype=np.int32)
    got = ctx.decode_logits(tok, want)
    ref = R.decode_logits(sd, dims, tok, want).numpy()
    e_log = R.rel_l2(got, ref)
    print("large-v2 full depth: encoder rel-L2 %.3e, logits rel-L2 %.3e" % (e_enc, e_log))
    assert gate("large_v2_full.enc", e_enc) and gate("large_v2_full.logits", e_log)
    for t in range(4):
        _check_choice(ref[0, t], int(got[0, t].argmax()))
    ctx.close()


def test_full_text_context_and_degenerate_audio(lively):
    """Maximum sizes and degenerate inputs: a decode that fills all 448 text positions (self-attention over the whole
    cache, last K/V slot written) must equal the stateless full-prefix recompute of wm_decode_logits; all-zero and
    clipped full-scale audio (KAT-1: log-mel == -1.5 everywhere) must go through without NaNs."""
    dims, _, sd, ctx = lively
    T = dims["n_text_ctx"]
    pcm = tones(2)
    prompt = [10, 21, 5, 7]
    toks, lens = ctx.transcribe_greedy(pcm, prompt, T - len(prompt))
    assert toks.shape == (2, T - len(prompt)) and np.all(lens == T - len(prompt))
    xa = ctx.encode_mel(ctx.logmel(pcm, out_dtype=np.float32))
    seq = np.concatenate([np.tile(prompt, (2, 1)), toks], axis=1)[:, :T].astype(np.int32)
    full = ctx.decode_logits(seq[:, :T], xa)                     # [2][448][V], no cache: every prefix recomputed
    for b in range(2):
        for pos in (len(prompt) - 1, 100, 222, T - 3, T - 2):    # logits at pos choose the token at pos + 1
            row = full[b, pos]
            assert row.max() - row[seq[b, pos + 1]] <= MARGIN, (b, pos)
    # degenerate audio
    zero = np.zeros((1, 480000), np.float32)
    loud = np.ones((1, 480000), np.float32) * np.where(np.arange(480000) % 2 == 0, 1.0, -1.0).astype(np.float32)
    mel0 = ctx.logmel(zero, out_dtype=np.float32)
    assert np.abs(mel0 + 1.5).max() <= 1e-5                      # f32 fast path (the f64 ABI path gives -1.5 exactly, test_frontend_gpu)
    for x in (zero, loud, np.round(loud * 32767).astype(np.int16)):
        t, _ = ctx.transcribe_greedy(x, prompt, 6)
        assert t.shape == (1, 6) and t.min() >= 0 and t.max() < dims["n_vocab"]
    xa0 = ctx.encode_mel(mel0)
    assert np.isfinite(xa0).all()
    want0 = R.encode(sd, dims, mel0).numpy()
    assert gate("tiny_lively.enc_silence", R.rel_l2(xa0, want0))


# ---------------------------------------------------------------------------------------------------------------------
# Round 2: the geometries BASELINE.json / the reference name, at full size (VERDICT r1 "configs untested")
def _perturb_ln_on_device(ctx, dims, seed=3):
    """Synthetic weights have LN gamma = 1, beta = 0, which would leave the LayerNorm folding of the decode GEMVs
    (W' = W gamma, c2 = b + W beta) unexercised: overwrite every LayerNorm parameter with a perturbed one."""
    rng = np.random.default_rng(seed)
    for name, shape, kind in W.tensor_specs(dims):
        if kind == W.K_LN_W:
            ctx.set_tensor(name, (1 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif kind == W.K_LN_B:
            ctx.set_tensor(name, (0.1 * rng.standard_normal(shape)).astype(np.float32))


def _oracle_weights(ctx, dims):
    return R.to_torch({n: ctx.get_tensor(n, s) for n, s, _ in W.tensor_specs(dims)})


def _check_greedy_against_teacher_forced_oracle(sd, dims, xa_ref, prompt, toks, scaled=False):
    """Every token the GPU chose must be (within MARGIN; scaled=True: within _scaled_margin of the row) the oracle's
    arg-max given the same prefix."""
    Bn, n_new = toks.shape
    seq = np.concatenate([np.tile(np.asarray(prompt, np.int32), (Bn, 1)), toks], axis=1).astype(np.int32)
    ref = R.decode_logits(sd, dims, seq[:, :-1], xa_ref).numpy()          # logits at position p choose token p + 1
    worst = 0.0
    for b in range(Bn):
        for i in range(n_new):
            row = ref[b, len(prompt) - 1 + i]
            gap = float(row.max() - row[toks[b, i]])
            worst = max(worst, gap)
            mg = _scaled_margin(row) if scaled else MARGIN
            assert gap <= mg, "chunk %d, new token %d: GPU picked %d, oracle gap %g" % (b, i, toks[b, i], gap)
    return worst


def test_small_geometry_the_reference_model(pkg):
    """The reference's ONLY model: whisper_to_cml.py:7 loads "small" (d 768, 12 heads, 12 + 12 layers; audio features
    (1,1500,768) at :29).  Full depth: encoder, the T = 1 decoder step of Whisper.swift:33-36, the language arg-max of
    :37-38, and a short KV-cached greedy decode, all against the fp32 oracle on the GPU's own weights."""
    import torch
    dims = pkg.binding.MODEL_DIMS["small"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(19)
    _perturb_ln_on_device(ctx, dims)
    ctx.finalize()
    sd = _oracle_weights(ctx, dims)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    pcm = np.stack([L.synth_chunk(31), tone_chunk(2)])
    mel = ctx.logmel(pcm)
    xa = ctx.encode_mel(mel)
    want = R.encode(sd, dims, mel).numpy()
    e_enc = R.rel_l2(xa, want)
    assert xa.shape == (2, 1500, 768) and gate("small_full.enc", e_enc)
    sot = np.full((2, 1), 50258, np.int32)                                  # Whisper.swift:34-35
    got = ctx.decode_logits(sot, want)
    ref = R.decode_logits(sd, dims, sot, want).numpy()
    e_log = R.rel_l2(got, ref)
    assert got.shape == (2, 1, 51865) and gate("small_full.logits_t1", e_log)
    lang = ctx.detect_language(want)                                        # :37-38, ids 50259...50357
    _, conf = R.detect_language(sd, dims, want)
    for b in range(2):
        _check_choice(conf[b], int(lang[b]))
    prompt = [50258, 50259, 50359, 50363]
    toks, lens = ctx.transcribe_greedy(pcm, prompt, 8)
    worst = _check_greedy_against_teacher_forced_oracle(sd, dims, want, prompt, toks)
    print("small (reference model): encoder rel-L2 %.3e, T=1 logits rel-L2 %.3e, worst greedy gap %.3g" % (e_enc, e_log, worst))
    ctx.close()


def test_large_v2_batch8_greedy_choices_against_the_oracle(pkg):
    """BASELINE.json configs[3] as written: large-v2 at FULL depth, a batch of 8 chunks, KV-cached greedy decode of 16
    new tokens -- every one of the 8 x 16 choices checked against the fp32 oracle (teacher-forced on the GPU's prefix),
    plus the encoder output of all 8 chunks.  LayerNorm parameters perturbed (folded decode GEMVs at d = 1280)."""
    import torch
    dims = pkg.binding.MODEL_DIMS["large-v2"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(7)
    _perturb_ln_on_device(ctx, dims)
    ctx.finalize()
    sd = _oracle_weights(ctx, dims)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    pcm = np.stack([L.synth_chunk(40 + i) if i % 2 == 0 else tone_chunk(i) for i in range(8)])
    mel = ctx.logmel(pcm)
    prompt = [50258, 50259, 50359, 50363]
    toks, lens = ctx.transcribe_greedy(pcm, prompt, 16)
    assert toks.shape == (8, 16) and np.all(lens == 16)
    want = R.encode(sd, dims, mel).numpy()
    e_enc = R.rel_l2(ctx.encode_mel(mel), want)
    assert gate("large_v2_full_b8.enc", e_enc)
    worst = _check_greedy_against_teacher_forced_oracle(sd, dims, want, prompt, toks)
    print("large-v2 full depth, B = 8 x 16 tokens: encoder rel-L2 %.3e, worst greedy gap %.3g logit" % (e_enc, worst))
    ctx.close()


def test_large_v3_full_depth_one_chunk(pkg):
    """BASELINE.json configs[4] geometry at FULL depth (32 + 32 layers, 128 mel bins, 51 866 tokens), one chunk:
    encoder, teacher-forced logits, language id over the 100 ids 50259...50358, greedy choices."""
    import torch
    dims = pkg.binding.MODEL_DIMS["large-v3"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(23)
    _perturb_ln_on_device(ctx, dims, seed=4)
    ctx.finalize()
    sd = _oracle_weights(ctx, dims)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    pcm = np.stack([L.synth_chunk(77)])
    mel = ctx.logmel(pcm, n_mels=128)
    xa = ctx.encode_mel(mel)
    want = R.encode(sd, dims, mel).numpy()
    e_enc = R.rel_l2(xa, want)
    tok = np.array([[50258, 50259, 50360, 50364]], dtype=np.int32)
    got = ctx.decode_logits(tok, want)
    ref = R.decode_logits(sd, dims, tok, want).numpy()
    e_log = R.rel_l2(got, ref)
    print("large-v3 full depth: encoder rel-L2 %.3e, logits rel-L2 %.3e" % (e_enc, e_log))
    assert got.shape == (1, 4, 51866) and gate("large_v3_full.enc", e_enc) and gate("large_v3_full.logits", e_log)
    lang = ctx.detect_language(want, sot=50258, lang_first=50259, lang_last=50358)
    _, conf = R.detect_language(sd, dims, want, sot=50258, lang_first=50259, lang_last=50358)
    _check_choice(conf[0], int(lang[0]))
    toks, _ = ctx.transcribe_greedy(pcm, [50258, 50259, 50360, 50364], 6)
    _check_greedy_against_teacher_forced_oracle(sd, dims, want, [50258, 50259, 50360, 50364], toks)
    ctx.close()


def test_large_v3_fifteen_chunk_shard_on_the_products_own_lanes(pkg):
    """BASELINE.json configs[4] at its PER-GPU size: 1 h of audio = 120 chunks over 8 GPUs = 15 chunks per rank, large-v3 at
    FULL depth (128 mel bins, 51 866 tokens), through wm_transcribe_greedy with the product's DEFAULT group / lane policy
    (model_api.cpp, measured in round 5: below 32 chunks ONE decode group -- 8 + 7 on two lanes was 4 % slower) and with an
    explicit lane count (wm_set_lanes 3: groups of 8 + 7 on two weight-sharing lanes).  15 distinct recordings, `lively`
    weights with perturbed LayerNorms; every row must equal the row of the same chunk decoded ALONE (one chunk per call:
    a one-row group, other launch shapes, no lanes), the rows must be pairwise distinct, and three of them are
    teacher-forced against the fp32 oracle over all their tokens."""
    import torch
    dims = pkg.binding.MODEL_DIMS["large-v3"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(20240928, matrix_gain=LIVELY_GAIN)
    _perturb_ln_on_device(ctx, dims, seed=11)
    ctx.finalize()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    pcm = np.stack([tone_chunk(i) if i % 3 == 0 else L.synth_chunk(200 + i) for i in range(15)])
    prompt = [50258, 50259, 50360, 50364]
    NEW = 48
    ctx.set_lanes(0)                                             # the product's own policy
    toks, lens = ctx.transcribe_greedy(pcm, prompt, NEW, eot=-1)
    assert toks.shape == (15, NEW) and np.all(lens == NEW)
    assert len({r.tobytes() for r in toks}) == 15, "15 different recordings must give 15 different token rows"
    for i in (0, 7, 8, 14):                                      # both groups, first and last row of each
        alone, _ = ctx.transcribe_greedy(pcm[i:i + 1], prompt, NEW, eot=-1)
        assert np.array_equal(alone[0], toks[i]), "chunk %d: its row inside the 15-chunk call differs from the chunk alone" % i
    ctx.set_lanes(3)                                             # ... and the same call as 8 + 7 rows on two lanes
    two, _ = ctx.transcribe_greedy(pcm, prompt, NEW, eot=-1)
    ctx.set_lanes(0)
    assert np.array_equal(two, toks)
    sd = _oracle_weights(ctx, dims)
    pick = [1, 8, 14]
    mel = ctx.logmel(pcm[pick], n_mels=128)
    xa = ctx.encode_mel(mel)
    want = R.encode(sd, dims, mel).numpy()
    assert gate("large_v3_shard.enc", R.rel_l2(xa, want), 6e-3)
    worst = _check_greedy_against_teacher_forced_oracle(sd, dims, xa, prompt, toks[pick], scaled=True)
    print("large-v3 full depth, 15-chunk shard on the product's lanes: worst greedy gap %.3g logit" % worst)
    ctx.close()


def test_bit_level_batch_invariance_at_d1280(pkg):
    """at d >= 1024 the round-1 kernels changed their summation order with the batch (split-K wave
    count, flash-decoding splits), so the same chunk could decode differently in groups of 1-4, 5-8 and > 8.  Round 2:
    the K split depends on K only and the attention streams are canonical -- the logits of a chunk are BIT-identical
    whatever batch it is decoded in (1 .. 128 rows: one block, several blocks, several workgroups per tile, wide tile
    groups), and so are the tokens of wm_transcribe_greedy whatever the grouping."""
    dims = dict(pkg.binding.MODEL_DIMS["large-v2"], n_audio_layer=2, n_text_layer=2)
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(13)
    _perturb_ln_on_device(ctx, dims, seed=9)
    ctx.finalize()
    n = 72
    pcm = np.stack([tone_chunk(i % 9) if i % 3 else L.synth_chunk(i % 5) for i in range(n)])
    xa = ctx.encode_mel(ctx.logmel(pcm))
    rng = np.random.default_rng(0)
    tok = rng.integers(0, 51865, size=(n, 3)).astype(np.int32)
    ref1 = ctx.decode_logits(tok[:1], xa[:1])
    for Bn in (3, 4, 5, 8, 16, 17, 40, 56, 72):   # (56 / 72 rows: fc2 as two batch blocks per workgroup, 40: one)
        lg = ctx.decode_logits(tok[:Bn], xa[:Bn])
        assert np.array_equal(lg[0], ref1[0]), "row 0 changed with batch %d" % Bn
        if Bn > 16:
            assert np.array_equal(lg[16], ctx.decode_logits(tok[16:17], xa[16:17])[0]), "row 16 at batch %d" % Bn
        if Bn > 40:
            assert np.array_equal(lg[40], ctx.decode_logits(tok[40:41], xa[40:41])[0]), "row 40 at batch %d" % Bn
    prompt = [50258, 50259, 50359, 50363]
    all_t, _ = ctx.transcribe_greedy(pcm, prompt, 5)                 # one call: balanced groups on the lanes
    for i in (0, 7, 23, 71):
        one, _ = ctx.transcribe_greedy(pcm[i:i + 1], prompt, 5)      # the chunk alone
        assert np.array_equal(one[0], all_t[i]), i
    grp, _ = ctx.transcribe_greedy(pcm[16:40], prompt, 5)            # a different grouping of the same chunks
    assert np.array_equal(grp, all_t[16:40])
    ctx.close()


def test_language_argmax_tie_takes_the_first(pkg):
    """Whisper.swift:38: Swift's max(by:) keeps the FIRST maximal element.  Tied logits are produced exactly by giving
    language tokens identical embedding rows (the logits are x . E[n]: equal rows, equal bits)."""
    dims = dict(R.TINY_DIMS)
    sd_np = nontrivial_ln(W.synthetic_state_dict(dims, seed=11))
    lang_first, lang_last = 20, 118
    E = sd_np["decoder.token_embedding.weight"].copy()
    E[lang_first:lang_last + 1] = E[lang_first + 5]                  # all 99 language logits tie
    sd_np["decoder.token_embedding.weight"] = E
    ctx = pkg.binding.Context(dims)
    ctx.load_state_dict(sd_np)
    ctx.finalize()
    _, mel = mels(ctx, 3)
    xa = ctx.encode_mel(mel)
    got = ctx.detect_language(xa, sot=10, lang_first=lang_first, lang_last=lang_last)
    assert np.array_equal(got, np.zeros(3, np.int32))
    # a two-way tie at the top: copy the winning row to an earlier and to a later id
    sd2 = nontrivial_ln(W.synthetic_state_dict(dims, seed=11))
    c2 = pkg.binding.Context(dims)
    c2.load_state_dict(sd2)
    c2.finalize()
    win = c2.detect_language(xa, sot=10, lang_first=lang_first, lang_last=lang_last)
    for b in range(3):
        w = int(win[b])
        if w < 2 or w > 96:
            continue
        E2 = sd2["decoder.token_embedding.weight"].copy()
        E2[lang_first + w - 2] = E2[lang_first + w]
        E2[lang_first + w + 2] = E2[lang_first + w]
        c2.set_tensor("decoder.token_embedding.weight", E2)
        c2.finalize()
        again = c2.detect_language(xa[b:b + 1], sot=10, lang_first=lang_first, lang_last=lang_last)
        assert int(again[0]) == w - 2, (w, again)
    ctx.close()
    c2.close()


def test_timestamp_rule_change_invalidates_the_captured_graph(lively):
    """the decode hipGraph bakes timestamp_begin / eot into its kernel arguments; calling
    wm_set_timestamp_rules again with other ids must not replay the stale capture."""
    dims, _, sd, ctx = lively
    pcm = tones(2)
    prompt = [10, 21, 5]
    try:
        ctx.set_timestamp_rules(True, 400, 7, 20)
        a, _ = ctx.transcribe_greedy(pcm, prompt, 8)
        ctx.set_timestamp_rules(True, 300, 7, 20)
        b, _ = ctx.transcribe_greedy(pcm, prompt, 8)
        assert a[:, 0].min() >= 400 and a[:, 0].max() <= 400 + 20   # first run: opens inside ITS initial-timestamp window
        # the transcript opens with a timestamp of the NEW range: [300, 300 + max_initial] -- disjoint from the old
        # window [400, 420], so a replay of the stale capture cannot pass
        assert b[:, 0].min() >= 300 and b[:, 0].max() <= 300 + 20
        # ... and it is the oracle's choice under the new ids (margin rule), teacher-forced on the GPU's own encoder output
        xa = ctx.encode_mel(ctx.logmel(pcm, out_dtype=np.float32))
        for r in range(2):
            row = R.decode_logits(sd, dims, np.asarray(prompt)[None], xa[r:r + 1])[0, -1].clone()
            R.timestamp_filter(row, [], 300, 7, 20)
            _check_choice(row.numpy(), int(b[r, 0]))
    finally:
        ctx.set_timestamp_rules(False)


def test_wm_multi_single_process_all_gpus_path(pkg):
    """SURVEY 8b / 8e behind the C ABI: wm_multi_create(devices[], n) + wm_multi_transcribe_greedy -- block partition, one
    host thread per GPU, one ncclAllGather of the token streams (RCCL linked into the .so).  This box has one GPU, so
    n = 1: the whole code path (ncclCommInitAll, pack, all-gather, unpack) runs and must reproduce the plain
    single-context result bit for bit; the dlopen-only C++ host (host/multi_main.cpp) checks the same from outside."""
    import subprocess
    import importlib.util
    from conftest import ROOT
    dims = dict(R.TINY_DIMS)
    mc = pkg.binding.MultiContext(dims, devices=[0])
    assert mc.n == 1
    mc.init_synthetic(11)
    pcm = tones(5)
    prompt = [10, 21, 5, 7]
    toks, lens = mc.transcribe_greedy(pcm, prompt, 6)
    one = mc.device_ctx(0)
    want, wl = one.transcribe_greedy(pcm, prompt, 6)
    assert np.array_equal(toks, want) and np.array_equal(lens, wl)
    with pytest.raises(pkg.binding.WhisperError, match="listed twice"):
        pkg.binding.MultiContext(dims, devices=[0, 0])
    mc.close()
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "openai-whisper-coreml_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_host(name="multi_main")
    r = subprocess.run([exe, pkg.binding.LIB_PATH, "tiny.en", "1", "5", "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith(("chunk ", "identical_"))]   # RCCL prints a banner
    assert len(lines) == 6 and lines[5] == "identical_to_single_gpu 1", r.stdout
    # a recording of arbitrary length through the library's own WAV reader + chunker (wm_wav_*): 70 s -> 3 windows
    import tempfile
    A = importlib.import_module("openai_whisper_coreml_amd.audio")
    x16 = np.round(np.concatenate([tone_chunk(1), tone_chunk(4), tone_chunk(2)[:160000]]) * 32767).astype(np.int16)
    with tempfile.TemporaryDirectory() as td:
        wav = os.path.join(td, "long.wav")
        A.write_wav_int16(wav, x16)
        r = subprocess.run([exe, pkg.binding.LIB_PATH, "tiny.en", "1", wav, "6"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith(("chunk ", "identical_"))]
        assert len(lines) == 4 and lines[3] == "identical_to_single_gpu 1", r.stdout
        # ... equal to the Python path on the same file (device-generated synthetic weights, seed 11, in both)
        c = pkg.binding.Context(pkg.binding.MODEL_DIMS["tiny.en"])
        c.init_synthetic(11)
        c.finalize()
        want, _ = c.transcribe_greedy(A.wav_to_chunks(wav), [50257, 50362, 10, 11], 6)
        got = np.array([[int(t) for t in l.split()[3:]] for l in lines[:3]])
        assert np.array_equal(got, want)
        c.close()


def test_converted_checkpoints_load_and_match_the_oracle(pkg, tmp_path):
    """convert_openai_pt on a torch.save'd fp16 {"dims", "model_state_dict"} and convert_hf_safetensors on
    a safetensors file, both loaded on the GPU through wm_load_weights; wm_encode / wm_decode_logits vs the oracle."""
    import torch
    from safetensors.numpy import save_file
    dims = dict(R.TINY_DIMS)
    sd_np = nontrivial_ln(W.synthetic_state_dict(dims, seed=21))
    half = {k: torch.from_numpy(v).half() for k, v in sd_np.items()}
    pt = os.path.join(tmp_path, "m.pt")
    torch.save({"dims": dict(dims), "model_state_dict": half}, pt)
    W.convert_openai_pt(pt, os.path.join(tmp_path, "a.wm"))
    hf = {k: np.ascontiguousarray(v.astype(np.float16)) for k, v in W.openai_to_hf_state_dict(sd_np).items()}
    save_file(hf, os.path.join(tmp_path, "m.safetensors"))
    W.convert_hf_safetensors(os.path.join(tmp_path, "m.safetensors"), dims, os.path.join(tmp_path, "b.wm"))
    outs = []
    for name in ("a.wm", "b.wm"):
        ctx = pkg.binding.Context(dims)
        ctx.load_weights(os.path.join(tmp_path, name))
        ctx.finalize()
        sd = _oracle_weights(ctx, dims)
        _, mel = mels(ctx, 1)
        xa = ctx.encode_mel(mel)
        want = R.encode(sd, dims, mel).numpy()
        assert gate("converted.enc", R.rel_l2(xa, want))
        tok = np.array([[10, 21, 5, 7]], np.int32)
        lg = ctx.decode_logits(tok, want)
        assert gate("converted.logits", R.rel_l2(lg, R.decode_logits(sd, dims, tok, want).numpy()))
        outs.append((xa, lg))
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])   # same weights either way


def test_layernorm_fold_is_robust_to_a_common_mode_offset(pkg):
    """The decode GEMVs multiply the RAW bf16 residual (LayerNorm folded into the weights).  bf16 rounding of the raw row
    scales with |x|, so a common-mode offset of the stream -- which LayerNorm removes exactly -- would leak 2^-9 |offset|
    of noise per element (measured before the fix: offset 1.0 at std ~0.2 -> logits rel-L2 4.6e-2, offset 10 -> 0.43).
    The bf16 copy is therefore stored MEAN-CENTRED (offset = the row's previous LayerNorm mean, ping-pong buffers) and the
    fold uses (mean - offset): the error must not grow with the offset, nor with a massive-activation feature."""
    dims = dict(R.TINY_DIMS)
    errs = []
    for off, outl in ((0.0, 0.0), (3.0, 0.0), (10.0, 0.0), (3.0, 30.0)):
        sd_np = nontrivial_ln(W.synthetic_state_dict(dims, seed=11))
        pe = sd_np["decoder.positional_embedding"].copy()
        pe += off
        pe[:, 5] += outl
        sd_np["decoder.positional_embedding"] = pe.astype(np.float32)
        ctx = pkg.binding.Context(dims)
        ctx.load_state_dict(sd_np)
        ctx.finalize()
        sd = _oracle_weights(ctx, dims)
        _, mel = mels(ctx, 1, start=3)
        xa = R.encode(sd, dims, mel).numpy()
        tok = np.array([[10, 21, 5, 7, 100, 200]], np.int32)
        e = R.rel_l2(ctx.decode_logits(tok, xa), R.decode_logits(sd, dims, tok, xa).numpy())
        errs.append(e)
        assert gate("offset_%g_%g.logits" % (off, outl), e, TOL["offset_outlier.logits" if outl else "offset.logits"])
        toks, _ = ctx.transcribe_greedy(np.stack([L.synth_chunk(3)]), [10, 21, 5, 7], 6)     # the graph-replayed path too
        _check_greedy_against_teacher_forced_oracle(sd, dims, xa, [10, 21, 5, 7], toks)
        ctx.close()
    assert max(errs) <= 2.0 * errs[0] + 1e-3, errs


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: the decode policy at the PRODUCTION vocabulary, and the widths between base and large (VERDICT r2 / ADVICE r2)
def _lively_on_device(ctx, dims, gain=4.0):
    """The `lively` recipe for models too big to generate on the host: every matrix (not the positional tables) of the
    device-generated synthetic weights scaled by a power of two (still bf16-exact), so that the token streams depend on
    the audio and on the decode history."""
    for name, shape, kind in W.tensor_specs(dims):
        if kind == W.K_MATRIX and "positional" not in name:
            ctx.set_tensor(name, ctx.get_tensor(name, shape) * np.float32(gain))


def _openai_like_suppress_list(n_vocab, specials):
    """A suppress list shaped like openai-whisper's `non_speech_tokens` + specials (about 90 ids): scattered ordinary ids,
    the special tokens, and ids inside the LAST, partial 16-column tile of the vocabulary."""
    rng = np.random.default_rng(50257)
    scattered = sorted({int(t) for t in rng.integers(1, 50256, size=82)})
    last_tile = (n_vocab - 1) // 16 * 16
    return sorted(set(scattered) | set(specials) | {last_tile + 1, n_vocab - 1})


def _scaled_margin(row):
    """Arg-max margin in logit units for a model whose logits are not O(1): MARGIN (0.05) was set on models with logit
    rms <= 1; the `lively` production-shape model has rms ~2.9 (matrices x 4), and the stated value tolerance is
    RELATIVE (rel-L2 <= 1e-2), so the admissible gap scales with the row: 5 % of its rms, never below MARGIN."""
    fin = row[np.isfinite(row)]
    return max(MARGIN, 0.05 * float(np.sqrt(np.mean(fin.astype(np.float64) ** 2)))) if fin.size else MARGIN


def _check_policy_choices(sd, dims, xa_rows, prompt, got, suppress, suppress_first, TS, EOT, MAXI):
    """Every choice of `got` against the oracle's filtered logits, teacher-forced on the GPU's own history -- the
    production-shape twin of test_timestamp_rules_follow_the_oracle (margins scaled to the logit rms)."""
    n_forced = n_near = 0
    n_new = got.shape[1]
    for b in range(got.shape[0]):
        seq = np.concatenate([prompt, got[b]])[None, :-1]
        ref = R.decode_logits(sd, dims, seq, xa_rows[b:b + 1])[0]
        for i in range(n_new):
            row = ref[len(prompt) - 1 + i].clone()
            row[suppress] = float("-inf")
            if i == 0:
                row[suppress_first] = float("-inf")
            unforced = row.clone()
            mg = _scaled_margin(ref[len(prompt) - 1 + i].numpy())
            forced, gap = R.timestamp_filter(row, [int(t) for t in got[b, :i]], TS, EOT, MAXI)
            n_forced += forced
            choice = int(got[b, i])
            if abs(gap) < mg:                # the summed-probability rule is a near-tie: either branch is right
                n_near += 1
                R.timestamp_filter(unforced, [int(t) for t in got[b, :i]], TS, EOT, MAXI, sum_rule=False)
                only_ts = unforced.clone()
                only_ts[:TS] = float("-inf")
                ok = float(only_ts.max() - only_ts[choice]) <= mg or float(unforced.max() - unforced[choice]) <= mg
                assert ok, (b, i, choice)
            else:
                r = row.numpy()
                g = float(r.max() - r[choice])
                assert g <= mg, "chunk %d, new token %d: GPU picked %d, oracle gap %g (margin %g)" % (b, i, choice, g, mg)
    return n_forced, n_near


@pytest.mark.parametrize("model,TS", [("large-v2", 50364), ("large-v3", 50365)])
def test_decode_policy_at_the_production_vocabulary(pkg, model, TS):
    """VERDICT r2 "weak" #2 / next #1a: SuppressTokens + SuppressBlank + ApplyTimestampRules at the shapes they run at in
    production -- vocabulary 51 865 / 51 866 = 3 242 tiles of 16 columns, timestamp_begin 50 364 / 50 365 (t_first =
    ts_begin >> 4 with a partial first tile: 50364 % 16 = 12, 50365 % 16 = 13), 1 501 timestamp ids over 94 tiles, the
    wide (TN = 4, NBLK = 2) logits launch and more than 16 rows through the multi-workgroup arg-max arrival counter --
    for decode groups of 8, 17 and 40 rows.  d = 1280, 20 heads, two layers (the oracle stays fast); every choice of
    the 8 distinct chunks is checked with R.timestamp_filter, the larger groups must reproduce those rows bit for bit."""
    import torch
    dims = dict(pkg.binding.MODEL_DIMS[model], n_audio_layer=2, n_text_layer=2)
    V = dims["n_vocab"]
    EOT, SOT, MAXI, NEW = 50257, 50258, 50, 20
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(29)
    _perturb_ln_on_device(ctx, dims, seed=6)
    _lively_on_device(ctx, dims)
    ctx.finalize()
    sd = _oracle_weights(ctx, dims)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    lang_last = SOT + (100 if model == "large-v3" else 99)
    specials = [SOT, lang_last + 1, lang_last + 2, lang_last + 3, lang_last + 4, lang_last + 5, TS - 1]  # translate .. notimestamps
    assert TS - 1 == lang_last + 6 and TS + 1500 == V - 1
    suppress = _openai_like_suppress_list(V, specials)
    assert len(suppress) >= 88 and max(suppress) == V - 1 and (V - 1) // 16 == (V + 15) // 16 - 1
    prompt = [SOT, SOT + 1, lang_last + 2]            # sot, <|en|>, <|transcribe|>: decoding WITH timestamps
    pcm = tones(8)
    mel = ctx.logmel(pcm, n_mels=dims["n_mels"], out_dtype=np.float32)
    ctx.set_suppress(suppress, [220, EOT])            # SuppressBlank: " " and <|endoftext|>
    ctx.set_timestamp_rules(True, TS, EOT, MAXI)
    try:
        got, lens = ctx.transcribe_greedy(pcm, prompt, NEW)
        assert got.shape == (8, NEW) and np.all(lens == NEW)
        # structure (whisper/decoding.py ApplyTimestampRules)
        assert np.all(got[:, 0] >= TS) and np.all(got[:, 0] <= TS + MAXI), got[:, 0]
        assert got.max() < V and not (set(got.ravel().tolist()) & set(suppress))
        for b in range(8):
            ts = [t for t in got[b] if t >= TS]
            assert all(x <= y for x, y in zip(ts, ts[1:])), (b, ts)
            isT = [t >= TS for t in got[b]]
            assert not any(isT[i] and isT[i + 1] and isT[i + 2] for i in range(NEW - 2)), (b, got[b])
        xa = ctx.encode_mel(mel)
        # the values first: teacher-forced logits of the GPU vs the oracle on this model (the stated relative tolerance)
        seq0 = np.concatenate([prompt, got[0]])[None, :8].astype(np.int32)
        e_log = R.rel_l2(ctx.decode_logits(seq0, xa[:1]), R.decode_logits(sd, dims, seq0, xa[:1]).numpy())
        assert gate("%s_policy.logits" % model, e_log, TOL["policy.logits"])
        n_forced, n_near = _check_policy_choices(sd, dims, xa, prompt, got, suppress, [220, EOT], TS, EOT, MAXI)
        n_text = int((got < TS).sum())
        print("%s production vocabulary: %d text / %d timestamp tokens, %d forced by the sum rule, %d near-ties, "
              "teacher-forced logits rel-L2 %.2e" % (model, n_text, got.size - n_text, n_forced, n_near, e_log))
        assert n_text > 0 and (got[:, 1:] >= TS).any()          # both branches of the rules are exercised
        # decode groups of 17 rows (two batch blocks, one arg-max workgroup per 16 rows) and of 40 (three blocks, the wide
        # logits launch): a call of 51 / 120 chunks runs as three balanced groups on the lanes
        for n in ((51, 120) if model == "large-v2" else (51,)):
            idx = [(3 * i + 1) % 8 for i in range(n)]
            many, ml = ctx.transcribe_greedy(pcm[idx], prompt, NEW)
            assert np.array_equal(many, got[idx]) and np.all(ml == NEW), n
    finally:
        ctx.set_timestamp_rules(False)
        ctx.set_suppress([], [])
    free, _ = ctx.transcribe_greedy(pcm[:2], prompt, 6)          # filters off again: text may open the transcript
    _check_greedy_against_teacher_forced_oracle(sd, dims, xa[:2], prompt, free, scaled=True)
    ctx.close()


@pytest.mark.parametrize("model,d,heads", [("small", 768, 12), ("medium", 1024, 16), ("d640", 640, 10), ("d576", 576, 9)])
def test_decode_groups_above_sixteen_at_d768_and_d1024(pkg, model, d, heads):
    """at d = 768 (whisper-small, the reference's model) and d = 1024 the fc2 product splits K over
    16 waves; with more than 16 rows its launch shape was rejected (WM_ERR_INVALID on every decode step).  ADVICE r3: the
    widths whose K = 4d splits over <= 8 waves with MORE than 8 k-steps each (d = 640: 8 x 10, d = 576: 6 x 12) hit the same
    rejection through the two-block shape; wm_create accepts them, so they must decode.  One decode step
    at 17 / 32 rows (wm_detect_language) and KV-cached greedy in groups of 17 / 18 (a 52-chunk call over three lanes)
    against the same chunks decoded in a group of 7, plus the oracle on the small group."""
    import torch
    dims = dict(pkg.binding.MODEL_DIMS["small"], n_audio_state=d, n_audio_head=heads, n_text_state=d, n_text_head=heads,
                n_audio_layer=2, n_text_layer=2)
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(31)
    _perturb_ln_on_device(ctx, dims, seed=2)
    _lively_on_device(ctx, dims)
    ctx.finalize()
    sd = _oracle_weights(ctx, dims)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    base = tones(7)
    mel = ctx.logmel(base, out_dtype=np.float32)
    xa = ctx.encode_mel(mel)
    one = ctx.detect_language(xa)
    _, conf = R.detect_language(sd, dims, xa)
    for b in range(7):
        _check_choice(conf[b], int(one[b]))
    for Bn in (17, 32, 96, 128):   # (96 / 128 rows: the fc2 product runs two batch blocks per workgroup at these widths)
        idx = [(3 * i + 2) % 7 for i in range(Bn)]
        assert np.array_equal(ctx.detect_language(xa[idx]), one[idx]), Bn
    prompt = [50258, 50259, 50359, 50363]
    want, _ = ctx.transcribe_greedy(base, prompt, 8)
    _check_greedy_against_teacher_forced_oracle(sd, dims, xa, prompt, want, scaled=True)
    idx = [(5 * i + 1) % 7 for i in range(52)]                   # 3 lanes -> groups of 18 / 17 / 17
    got, lens = ctx.transcribe_greedy(base[idx], prompt, 8)
    assert np.array_equal(got, want[idx]) and np.all(lens == 8)
    ctx.close()


def _truncate_like_the_host(full, eot, budgets=None):
    """What decoding every position and truncating on the host gives (the round-2 behaviour, and the definition of the
    result): length = first eot + 1, capped by the chunk's budget; padding = eot."""
    Bn, n = full.shape
    toks = np.full((Bn, n), eot, np.int32)
    lens = np.zeros(Bn, np.int32)
    for b in range(Bn):
        ln = n if budgets is None else min(n, int(budgets[b]))
        hit = np.nonzero(full[b, :ln] == eot)[0] if eot >= 0 else np.zeros(0, int)
        if hit.size:
            ln = int(hit[0]) + 1
        toks[b, :ln] = full[b, :ln]
        lens[b] = ln
    return toks, lens


def test_early_stop_equals_truncation_and_cuts_the_work(lively, pkg):
    """a sequence that has emitted <|endoftext|> (or used up its token budget) leaves the decode -- its
    rows drop out of the attention pair walk, and a group whose rows have all stopped is not decoded any further (the host
    polls a device-side live count between bursts of positions).  Tokens and lengths must be exactly what decoding every
    position and truncating gives; groups of 1 / 7 / 19 (three lanes) / 120 (three groups of 40: three arg-max workgroups,
    cross-workgroup live-list rebuild) rows."""
    dims, _, _, ctx = lively
    base = tones(7)
    prompt = [10, 21, 5]
    NEW = 40
    full7, _ = ctx.transcribe_greedy(base, prompt, NEW, eot=-1)
    vals, counts = np.unique(full7[:, 2:], return_counts=True)
    rng = np.random.default_rng(4)
    for n in (1, 7, 19, 120):
        idx = [(5 * i + 3) % 7 for i in range(n)]
        full = full7[idx]
        # stop tokens that occur at different positions in different rows (and one that never occurs)
        cands = [int(v) for v in vals[np.argsort(-counts)][:3]] + [int(full7[0, 9]), 1023]
        for eot in cands:
            want_t, want_l = _truncate_like_the_host(full, eot)
            got_t, got_l = ctx.transcribe_greedy(base[idx], prompt, NEW, eot=eot)
            assert np.array_equal(got_l, want_l), (n, eot, got_l, want_l)
            assert np.array_equal(got_t, want_t), (n, eot)
        budgets = rng.integers(1, NEW + 8, size=n)          # some above max_new: clamped
        want_t, want_l = _truncate_like_the_host(full, -1, np.minimum(budgets, NEW))
        got_t, got_l = ctx.transcribe_greedy(base[idx], prompt, NEW, eot=-1, budgets=budgets)
        assert np.array_equal(got_l, want_l) and np.array_equal(got_t, want_t), n
        eot = cands[0]                                       # both at once
        want_t, want_l = _truncate_like_the_host(full, eot, np.minimum(budgets, NEW))
        got_t, got_l = ctx.transcribe_greedy(base[idx], prompt, NEW, eot=eot, budgets=budgets)
        assert np.array_equal(got_l, want_l) and np.array_equal(got_t, want_t), n
    # degenerate budgets / stops: every row done after its FIRST generated token (the group is over after one burst)
    one_t, one_l = ctx.transcribe_greedy(base, prompt, NEW, eot=-1, budgets=[1] * 7)
    assert np.all(one_l == 1) and np.array_equal(one_t[:, 0], full7[:, 0]) and np.all(one_t[:, 1:] == -1)
    for r in range(7):                                       # eot == a row's very first token
        e = int(full7[r, 0])
        want_t, want_l = _truncate_like_the_host(full7, e)
        got_t, got_l = ctx.transcribe_greedy(base, prompt, NEW, eot=e)
        assert np.array_equal(got_l, want_l) and np.array_equal(got_t, want_t) and got_l[r] == 1, r
    # budgets are consumed by one call and must match its B
    again, _ = ctx.transcribe_greedy(base, prompt, NEW, eot=-1)
    assert np.array_equal(again, full7)
    ctx.set_token_budgets([3, 3])
    with pytest.raises(pkg.binding.WhisperError, match="budgets"):
        ctx.transcribe_greedy(base, prompt, NEW)
    with pytest.raises(pkg.binding.WhisperError, match="< 1"):
        ctx.set_token_budgets([0])
    # budgets are consumed by the NEXT call even when that call fails validation -- they must not stay armed and
    # silently truncate a later call that happens to have the same B
    ctx.set_token_budgets([2] * 7)
    with pytest.raises(pkg.binding.WhisperError, match="context"):
        ctx.transcribe_greedy(base, prompt, 446)             # prompt + new tokens do not fit: fails before decoding
    again, lens_again = ctx.transcribe_greedy(base, prompt, NEW, eot=-1)
    assert np.all(lens_again == NEW) and np.array_equal(again, full7)
    # timestamp rules + suppress lists + early stop together (the rule state of a finished row is simply not used)
    TS, EOT = 900, 890
    ctx.set_suppress(list(range(EOT + 1, TS)), [EOT])
    ctx.set_timestamp_rules(True, TS, EOT, 20)
    try:
        f2, _ = ctx.transcribe_greedy(base, prompt, NEW, eot=-1)
        for eot in (EOT, int(f2[1, 5])):
            want_t, want_l = _truncate_like_the_host(f2, eot)
            got_t, got_l = ctx.transcribe_greedy(base, prompt, NEW, eot=eot)
            assert np.array_equal(got_l, want_l) and np.array_equal(got_t, want_t), eot
    finally:
        ctx.set_timestamp_rules(False)
        ctx.set_suppress([], [])


def test_early_stop_decode_time_follows_the_longest_live_sequence(pkg):
    """Decode time must scale with the positions actually needed: at large-v2 width (4 layers), 24 chunks, 200 new tokens,
    (a) every chunk stopping after 20 tokens costs a fraction of the full decode, (b) ONE straggler keeps the group alive
    to the end but the 23 finished rows no longer stream their cross-attention caches."""
    dims = dict(pkg.binding.MODEL_DIMS["large-v2"], n_audio_layer=2, n_text_layer=4)
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(5)
    ctx.finalize()
    n, NEW = 24, 200
    pcm = np.stack([L.synth_chunk(i % 3) for i in range(n)]).astype(np.float32)
    prompt = [50258, 50259, 50359, 50363]
    dp = ctx.to_device(np.round(pcm * 32767).astype(np.int16))

    def run(budgets):
        best = None
        for _ in range(2):
            toks, lens = ctx.transcribe_greedy(dp, prompt, NEW, eot=-1, mem=pkg.binding.WM_MEM_DEVICE,
                                               pcm_dtype=pkg.binding.WM_I16, B=n, budgets=budgets)
            ms = float(ctx.last_stage_ms()[2])
            best = ms if best is None else min(best, ms)
        return best, toks, lens

    t_full, full, _ = run(None)
    t_short, short, l_short = run([20] * n)
    t_strag, strag, l_strag = run([20] * (n - 1) + [NEW])
    print("decode ms: full %.1f, all stop at 20: %.1f, one straggler: %.1f" % (t_full, t_short, t_strag))
    assert np.all(l_short == 20) and np.array_equal(short[:, :20], full[:, :20])
    assert l_strag.tolist() == [20] * (n - 1) + [NEW] and np.array_equal(strag[-1], full[-1])
    assert t_short <= 0.25 * t_full, (t_short, t_full)          # 23 positions (+ <= 2 bursts of slack) instead of 203
    # weights still stream, 23 of 24 cache streams do not.  (Round 5: the 24 chunks are ONE decode group now -- measured
    # 0.81 of the full decode; rounds 3-4 cut them into three groups of 8, two of which ended after 20 tokens: 0.6 - 0.7)
    assert t_strag <= 0.88 * t_full, (t_strag, t_full)
    ctx.set_lanes(3)                                            # the rounds-3-4 split: finished GROUPS are not decoded further
    t_strag3, strag3, _ = run([20] * (n - 1) + [NEW])
    ctx.set_lanes(0)
    print("one straggler, three groups of 8: %.1f ms" % t_strag3)
    assert np.array_equal(strag3, strag)
    ctx.dev_free(dp)
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the TIMED workload itself against the oracle, at its own size
def test_timed_workload_against_the_oracle_at_its_own_size(pkg):
    """bench.py's timed region decodes large-v2 at FULL depth in decode groups of 56 rows (7 batches of 8), 224 new tokens
    after a 4-token prompt (228 positions), with 8-position burst graphs, three groups in flight.  Until round 3 no decode
    position beyond ~20 had been compared with the oracle at production width, and never in a multi-block group.  Here:
    the `lively` random-init model (matrix gain 4: tokens depend on the audio and on the history), one decode group of 56
    rows (8 distinct chunks tiled 7 times) on ONE lane and the same chunks inside a 168-row call on THREE lanes; the rows
    must be pairwise distinct per chunk, identical across the copies of a chunk and across the two calls, and ALL 224
    choices of three distinct rows must be (within the scaled margin) the fp32 oracle's arg-max, teacher-forced on the GPU's
    own prefix.  Measured margins are printed (profiles/r04_parity_margins.txt holds a run's figures)."""
    import torch
    dims = pkg.binding.MODEL_DIMS["large-v2"]
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(20240928, matrix_gain=LIVELY_GAIN)       # bench.py's weights
    _perturb_ln_on_device(ctx, dims, seed=8)
    ctx.finalize()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    base = np.stack([tone_chunk(i) if i % 2 else L.synth_chunk(60 + i) for i in range(8)])
    prompt = [50258, 50259, 50359, 50363]
    NEW = dims["n_text_ctx"] // 2                                # 224, openai-whisper's sample_len
    idx56 = [i % 8 for i in range(56)]
    ctx.set_lanes(1)
    t56, l56 = ctx.transcribe_greedy(base[idx56], prompt, NEW, eot=-1)   # ONE decode group of 56 rows (4 batch blocks)
    ctx.set_lanes(0)
    assert t56.shape == (56, NEW) and np.all(l56 == NEW)
    rows = t56[:8]
    assert len({r.tobytes() for r in rows}) == 8, "the 8 chunks must decode to pairwise distinct token rows"
    n_distinct = [len(set(r.tolist())) for r in rows]
    changes = [int((r[1:] != r[:-1]).sum()) for r in rows]
    print("distinct tokens per row", n_distinct, "token changes per row", changes)
    # history-dependent: most rows keep changing over the 224 positions (a random-init model may drive a chunk into a short
    # cycle -- measured: two of the four noise chunks do, after ~3 tokens); the rows teacher-forced below are the liveliest
    assert sum(c >= 16 for c in changes) >= 5 and sum(n >= 8 for n in n_distinct) >= 5, (n_distinct, changes)
    assert np.array_equal(t56, rows[idx56])                      # every copy of a chunk, in every batch block
    idx168 = [(5 * i + 3) % 8 for i in range(168)]               # three lanes x 56 rows, other row positions
    t168, l168 = ctx.transcribe_greedy(base[idx168], prompt, NEW, eot=-1)
    assert np.array_equal(t168, rows[idx168]) and np.all(l168 == NEW)
    # teacher-forced oracle on three distinct rows (a tone chunk, two noise chunks) x ALL 224 positions
    sd = _oracle_weights(ctx, dims)
    pick = sorted(int(i) for i in np.argsort(changes)[-3:])
    mel = ctx.logmel(base[pick])
    xa = ctx.encode_mel(mel)
    worst = _check_greedy_against_teacher_forced_oracle(sd, dims, xa, prompt, rows[pick], scaled=True)
    # the values too: GPU teacher-forced logits (stateless full-prefix path) vs the oracle over the whole 228-token prefix
    seq = np.concatenate([np.tile(np.asarray(prompt, np.int32), (1, 1)), rows[pick[:1]]], axis=1)[:, :-1].astype(np.int32)
    ref = R.decode_logits(sd, dims, seq, xa[:1]).numpy()
    got = ctx.decode_logits(seq, xa[:1])
    e_all = R.rel_l2(got, ref)
    e_tail = R.rel_l2(got[:, -32:], ref[:, -32:])
    agree = float((got[0].argmax(axis=1) == ref[0].argmax(axis=1)).mean())
    print("large-v2 full depth, lively, 56-row group x 224 tokens: worst greedy gap %.3g logit (rms %.2f); teacher-forced "
          "logits rel-L2 %.3e over 227 positions, %.3e over the last 32; arg-max agreement %.3f"
          % (worst, float(np.sqrt((ref.astype(np.float64) ** 2).mean())), e_all, e_tail, agree))
    assert gate("workload.logits_all", e_all) and gate("workload.logits_tail", e_tail)
    ctx.close()


@pytest.mark.parametrize("case", ["tiny", "tiny_lively", "tiny.en", "small_2layer", "large-v2_full"])
def test_fp32_debug_path_matches_the_oracle(pkg, case):
    """BASELINE.md's parity gate: "fp32 debug path must match to <= 1e-4 rel-L2" (VERDICT r3 next #2).  The debug library's
    wmdbg_set_precision(ctx, WM_F32) runs wm_encode / wm_decode_logits with f32 activations, f32 K/V and f32 accumulation on
    the product's own weight buffers (bf16 values, product layouts: csrc/f32_path.hip).  Encoder output AND teacher-forced
    logits within 1e-4 of oracle/whisper_ref.py (measured 1e-7 .. 3e-6, profiles/r04_parity_margins.txt) at tiny dims, tiny.en
    in full, small (2 + 2 layers) and large-v2 at FULL depth -- so whatever the bf16 product path differs by at the same
    geometry (3e-4 .. 5e-3 encoder, 4e-3 .. 6e-3 logits) is rounding, not semantics, layout or indexing.  The same context
    switched back to WM_BF16 must reproduce the product library's own output bit for bit."""
    import torch
    MD = pkg.binding.MODEL_DIMS
    dims, seed, gain, toks = {
        "tiny": (dict(R.TINY_DIMS), 11, 1.0, [1, 7, 300, 1023, 5, 9]),
        "tiny_lively": (dict(R.TINY_DIMS), 11, LIVELY_GAIN, [1, 7, 300, 1023, 5, 9]),
        "tiny.en": (MD["tiny.en"], 7, 1.0, [50257, 50362, 100, 2000]),
        "small_2layer": (dict(MD["small"], n_audio_layer=2, n_text_layer=2), 19, 1.0, [50258, 50259, 50359, 50363, 1000]),
        "large-v2_full": (MD["large-v2"], 7, 1.0, [50258, 50259, 50359, 50363, 1000]),
    }[case]
    ctx = pkg.binding.Context(dims, debug=True)
    ctx.init_synthetic(seed, matrix_gain=gain)
    _perturb_ln_on_device(ctx, dims)
    ctx.finalize()
    sd = _oracle_weights(ctx, dims)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    n = 1 if case == "large-v2_full" else 2
    pcm = np.stack([tone_chunk(1), L.synth_chunk(5)][:n])
    mel = ctx.logmel(pcm, out_dtype=np.float32)
    want = R.encode(sd, dims, mel).numpy()
    tok = np.tile(np.asarray(toks, np.int32), (n, 1))
    ref = R.decode_logits(sd, dims, tok, want).numpy()
    bf_xa, bf_lg = ctx.encode_mel(mel), ctx.decode_logits(tok, want)
    ctx.set_precision(True)
    xa, lg = ctx.encode_mel(mel), ctx.decode_logits(tok, want)
    e_enc, e_log = R.rel_l2(xa, want), R.rel_l2(lg, ref)
    print("%s: fp32 debug path encoder rel-L2 %.2e, logits %.2e (bf16 product path: %.2e, %.2e)"
          % (case, e_enc, e_log, R.rel_l2(bf_xa, want), R.rel_l2(bf_lg, ref)))
    assert gate("f32.enc", e_enc) and gate("f32.logits", e_log)
    assert np.array_equal(lg.argmax(-1), ref.argmax(-1))
    ctx.set_precision(False)
    assert np.array_equal(ctx.encode_mel(mel), bf_xa) and np.array_equal(ctx.decode_logits(tok, want), bf_lg)
    ctx.close()


def test_flat_cross_attention_launch_shapes_are_bitwise_equal(pkg):
    """Below 96 (sequence, head) pairs the cross-attention runs as a flat deal of (pair, stream) units (+ a combine launch),
    since round 4 with every block of a stream requested up front (the latency shape); the arithmetic of a stream is the
    same in every shape.  Same bits at tiny.en: a row decoded alone (6 pairs: flat, deep) vs inside a batch of 3 (18 pairs)
    vs inside a batch of 16 (96 pairs: one workgroup per pair, block by block, merged in the kernel) -- logits and greedy
    tokens; and the flat kernel block by block (debug tuning xattn_no_deep, a second context so that no captured graph is
    reused) gives the very same outputs."""
    import ctypes
    dims = pkg.binding.MODEL_DIMS["tiny.en"]
    pcm = tones(3)
    prompt = [50257, 50362]
    rng = np.random.default_rng(3)
    tok = rng.integers(0, 51864, size=(16, 5)).astype(np.int32)
    outs = []
    lib = pkg.binding.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    try:
        for no_deep in (0, 1):
            assert lib.wmdbg_set_tuning(b"xattn_no_deep", no_deep) == 0
            ctx = pkg.binding.Context(dims, debug=True)
            ctx.init_synthetic(17, matrix_gain=LIVELY_GAIN)
            _perturb_ln_on_device(ctx, dims, seed=5)
            ctx.finalize()
            xa = ctx.encode_mel(ctx.logmel(pcm, out_dtype=np.float32))
            xa16 = xa[[i % 3 for i in range(16)]]
            lg1 = ctx.decode_logits(tok[:1], xa16[:1])            # 6 pairs: flat cross-attention
            lg3 = ctx.decode_logits(tok[:3], xa16[:3])            # 18 pairs
            lg16 = ctx.decode_logits(tok, xa16)                   # 96 pairs: one workgroup per pair, merged in the kernel
            gen, _ = ctx.transcribe_greedy(pcm, prompt, 40)
            solo, _ = ctx.transcribe_greedy(pcm[1:2], prompt, 40)
            outs.append((lg1, lg3, lg16, gen, solo))
            ctx.close()
    finally:
        lib.wmdbg_set_tuning(b"reset", 0)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    lg1, lg3, lg16, gen, solo = outs[0]
    assert np.array_equal(lg1[0], lg16[0]) and np.array_equal(lg3, lg16[:3])
    assert np.array_equal(solo[0], gen[1]) and len({r.tobytes() for r in gen}) == 3


def test_fused_query_cross_attention_is_bitwise_equal_to_the_two_launches(pkg):
    """Round 5: with 96 .. 256 (sequence, head) pairs and nothing else decoding on the device, cross_attn_ln + query projection
    run INSIDE the cross-attention launch (dec_xattn_fq_kernel: every pair's workgroup forms its own 64 query values with the
    GEMV's own K split, summation order and LayerNorm fold).  Same bits by construction -- checked here: large-v2 width (20
    heads), 3 decoder layers, groups of 5 / 8 / 12 rows (100 / 160 / 240 pairs) with the fusion on (product) and off (debug
    knob xattn_fuse_q = 0, a fresh context so that no captured graph is reused): teacher-forced logits and 40 greedy tokens
    identical; with a stop token and per-row budgets (the fused kernel walks the live list) identical too; and a row decoded
    in a group of 8 (fused) equals the same row alone (1 row: flat deal, separate GEMV)."""
    import ctypes
    dims = dict(pkg.binding.MODEL_DIMS["large-v2"], n_audio_layer=2, n_text_layer=3)
    lib = pkg.binding.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    pcm = np.stack([tone_chunk(i) if i % 2 else L.synth_chunk(500 + i) for i in range(12)])
    prompt = [50258, 50259, 50359, 50363]
    rng = np.random.default_rng(12)
    tok = rng.integers(0, 51865, size=(12, 6)).astype(np.int32)
    outs = []
    try:
        for fuse in (1, 0):
            lib.wmdbg_set_tuning(b"reset", 0)
            assert lib.wmdbg_set_tuning(b"xattn_fuse_q", fuse) == 0
            ctx = pkg.binding.Context(dims, debug=True)
            ctx.init_synthetic(41, matrix_gain=LIVELY_GAIN)
            _perturb_ln_on_device(ctx, dims, seed=9)
            ctx.finalize()
            ctx.set_lanes(1)
            xa = ctx.encode_mel(ctx.logmel(pcm, out_dtype=np.float32))
            res = []
            for n in (5, 8, 12):
                res.append(ctx.decode_logits(tok[:n], xa[:n]))
                res.append(ctx.transcribe_greedy(pcm[:n], prompt, 40)[0])
            free = res[3]                                              # the 8-row greedy run
            stop_tok = int(np.bincount(free[:, 5:].ravel()).argmax())      # a token that does occur: rows stop at different times
            t_es, l_es = ctx.transcribe_greedy(pcm[:8], prompt, 40, eot=stop_tok, budgets=[40, 7, 40, 19, 3, 40, 40, 11])
            res += [t_es, l_es, ctx.transcribe_greedy(pcm[2:3], prompt, 40)[0]]
            outs.append(res)
            ctx.close()
    finally:
        lib.wmdbg_set_tuning(b"reset", 0)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    assert len({r.tobytes() for r in outs[0][3]}) >= 6
    assert np.array_equal(outs[0][-1][0], outs[0][3][2])              # row 2 alone == row 2 in the fused group of 8
    assert outs[0][-2].min() < 40                                     # the stop token / budgets did end rows early
    # more than ONE 16-row block in the fused shape (blk > 0: the means of a block are written by ITS head-0
    # workgroups, blocks whose rows are all finished are skipped) -- base width, 8 heads x 24 rows = 192 pairs, early stop on
    # with budgets that finish the whole SECOND block long before the first
    dims_b = dict(pkg.binding.MODEL_DIMS["base"], n_audio_layer=2, n_text_layer=3)
    pcm_b = np.stack([tone_chunk(i) if i % 2 else L.synth_chunk(700 + i) for i in range(24)])
    bud = [40 if i < 16 else 3 + (i % 5) for i in range(24)]
    bud[5], bud[9] = 6, 13
    outs_b = []
    try:
        for fuse in (1, 0):
            lib.wmdbg_set_tuning(b"reset", 0)
            assert lib.wmdbg_set_tuning(b"xattn_fuse_q", fuse) == 0
            ctx = pkg.binding.Context(dims_b, debug=True)
            ctx.init_synthetic(43, matrix_gain=LIVELY_GAIN)
            _perturb_ln_on_device(ctx, dims_b, seed=10)
            ctx.finalize()
            ctx.set_lanes(1)
            free = ctx.transcribe_greedy(pcm_b, prompt, 40)[0]
            t_es, l_es = ctx.transcribe_greedy(pcm_b, prompt, 40, eot=-1, budgets=bud)
            outs_b.append((free, t_es, l_es, ctx.transcribe_greedy(pcm_b[17:18], prompt, 40)[0]))
            ctx.close()
    finally:
        lib.wmdbg_set_tuning(b"reset", 0)
    for a, b in zip(outs_b[0], outs_b[1]):
        assert np.array_equal(a, b)
    free, t_es, l_es, solo = outs_b[0]
    assert list(l_es) == bud and np.array_equal(solo[0], free[17])
    for i in range(24):                                               # early stop == decode everything and truncate
        assert np.array_equal(t_es[i, :bud[i]], free[i, :bud[i]])


def test_cross_attention_persistent_and_short_lived_shapes_are_bitwise_equal_under_load(pkg):
    """which cross-attention launch shape a burst of positions gets (<= 256 persistent workgroups walking the
    pairs, or one short-lived workgroup per pair) is decided at run time from what else decodes on the device, so bit-level
    batch invariance needs the two shapes to give the same bits -- not just statistically.  tiny.en, a 48-row group
    (288 pairs: persistent workgroups walk two pairs each), decoded (a) alone = persistent, (b) while a second context's
    thread keeps decoding on the same device = short-lived (the process-wide counter), (c) under the same load with the
    debug knob xattn_never_short = persistent under load.  A fresh context per case (a captured graph keeps its shape).
    Greedy tokens of 64 positions: identical in all three."""
    import ctypes
    import threading
    import time
    dims = pkg.binding.MODEL_DIMS["tiny.en"]
    lib = pkg.binding.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    pcm = np.stack([tone_chunk(i) if i % 2 else L.synth_chunk(300 + i) for i in range(48)])
    prompt = [50257, 50362]

    def run(never_short, load):
        lib.wmdbg_set_tuning(b"reset", 0)
        assert lib.wmdbg_set_tuning(b"xattn_never_short", never_short) == 0
        ctx = pkg.binding.Context(dims, debug=True)
        ctx.init_synthetic(29, matrix_gain=LIVELY_GAIN)
        _perturb_ln_on_device(ctx, dims, seed=6)
        ctx.finalize()
        ctx.set_lanes(1)
        other = ctx.clone() if load else None
        stop = threading.Event()

        def churn():
            other.set_lanes(1)
            while not stop.is_set():
                other.transcribe_greedy(pcm[:24], prompt, 64)
        th = threading.Thread(target=churn) if load else None
        try:
            if th:
                th.start()
                time.sleep(0.2)                  # the other context's decode is in flight
            return [ctx.transcribe_greedy(pcm, prompt, 64)[0] for _ in range(3)]
        finally:
            stop.set()
            if th:
                th.join()
                other.close()
            ctx.close()
    try:
        alone = run(0, False)
        assert len({r.tobytes() for r in alone[0]}) >= 24
        for t in alone[1:] + run(0, True) + run(1, True):
            assert np.array_equal(t, alone[0])
    finally:
        lib.wmdbg_set_tuning(b"reset", 0)


def _hip_free_bytes():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_context_lifecycle_returns_device_memory(pkg):
    """wm_create ... wm_destroy in a long-lived host process (the reference keeps ONE `Whisper` for the app's lifetime,
    ContentView.swift:11,15 -- a server creates and drops many): every path that allocates lazily is driven -- encoder
    work space, cross-K/V, lanes with their captured graphs (both cross-attention variants), early stop, budgets, a clone,
    the stateless logits call -- and after wm_destroy the device's free memory is back where it was.  The first cycle is
    the warm-up (HIP's own module / graph pools fill once); cycles 2..4 must not lose anything."""
    dims = dict(R.TINY_DIMS)
    pcm = tones(5)

    def cycle():
        ctx = pkg.binding.Context(dims)
        ctx.init_synthetic(3, matrix_gain=4.0)
        ctx.finalize()
        ctx.set_lanes(3)
        t0, l0 = ctx.transcribe_greedy(pcm, [10, 21], 12)
        ctx.set_token_budgets([3, 12, 7, 12, 5])
        t1, l1 = ctx.transcribe_greedy(pcm, [10, 21], 12, eot=int(t0[0, 4]))
        c = ctx.clone()
        t2, _ = c.transcribe_greedy(pcm[:2], [10, 21], 6)
        xa = ctx.encode_mel(ctx.logmel(pcm[:1]))
        ctx.decode_logits(np.array([[10, 21, 5]], np.int32), xa)
        c.close()
        ctx.close()
        return t0, t2

    first = cycle()
    base = _hip_free_bytes()
    for i in range(3):
        again = cycle()
        assert np.array_equal(again[0], first[0]) and np.array_equal(again[1], first[1])
        lost = base - _hip_free_bytes()
        assert lost <= (8 << 20), "cycle %d: %.1f MiB of device memory not returned" % (i + 2, lost / 2**20)


def test_sub_chip_lanes_carry_the_decode_policy_state(pkg):
    """Round 6: the product policy runs a 24 .. 128-chunk call of a NARROW model (decoder width <= 512) as two decode groups on
    CU-masked half-chip lanes (wm_lane_parts) -- weight-sharing clones created on first use.  Whatever state a call depends on
    must reach them: suppress lists and timestamp rules set BEFORE the lanes exist (copied at clone time) and changed AFTER
    (propagated by wm_set_suppress / wm_set_timestamp_rules), a stop token, per-chunk budgets.  Base width, 2 + 2 layers, 40
    chunks: every variant's tokens and lengths on the masked lanes (default policy) == ONE group on the context's own stream
    (wm_set_lanes(1)), bit for bit; the debug policy hook confirms the call is a two-part one."""
    dims = dict(pkg.binding.MODEL_DIMS["base"], n_audio_layer=2, n_text_layer=2)
    lib = pkg.binding.load_debug_library()
    assert lib.wmdbg_lane_parts(40, 3, 0, dims["n_text_state"]) == 2
    ctx = pkg.binding.Context(dims)
    ctx.init_synthetic(77, matrix_gain=W.lively_gain(dims))
    _perturb_ln_on_device(ctx, dims, seed=4)
    ctx.finalize()
    pcm = np.stack([tone_chunk(i) if i % 3 else L.synth_chunk(900 + i) for i in range(40)])
    prompt = [50258, 50259, 50359]
    TS, EOT, NEW = 50364, 50257, 40
    specials = list(range(50258, 50364))

    def both(**kw):
        ctx.set_lanes(0)
        a = ctx.transcribe_greedy(pcm, prompt, NEW, **kw)
        ctx.set_lanes(1)
        b = ctx.transcribe_greedy(pcm, prompt, NEW, **kw)
        ctx.set_lanes(0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        return a
    # 1. rules set before the masked lanes exist
    ctx.set_suppress(specials, [EOT])
    ctx.set_timestamp_rules(True, TS, EOT, 50)
    t1, _ = both()
    assert np.all(t1[:, 0] >= TS) and not (set(t1.ravel().tolist()) & set(specials))
    # 2. rules CHANGED after they exist: another suppress list, timestamps off; a stop token that does occur; budgets
    ctx.set_timestamp_rules(False, TS, EOT, 50)
    ctx.set_suppress(specials + [int(t1[0, 3])], [])
    t2, _ = both()
    assert not np.array_equal(t1, t2) and int(t1[0, 3]) not in set(t2.ravel().tolist())
    stop_tok = int(np.bincount(t2[:, 4:].ravel()).argmax())
    bud = [int(b) for b in np.random.default_rng(3).integers(5, NEW + 1, size=40)]
    t3, l3 = both(eot=stop_tok, budgets=bud)
    assert l3.min() < NEW and len({r.tobytes() for r in t2}) >= 30
    for i in range(40):
        assert np.array_equal(t3[i, :l3[i]], t2[i, :l3[i]])          # early stop == decode everything and truncate
    ctx.close()


def test_masked_lane_soak_alternating_shapes(pkg):
    """tools/gpu_masked_lane_soak.py as a test: base at full depth, 40 calls of 8 .. 72 chunks in random order through the
    product policy (two CU-masked half-chip groups from 24 chunks, one group below), early stop on / off at random -- every
    result equals the single-group reference decoded once per (size, stop mode).  What a one-shot test cannot see: graph
    re-use across alternating shapes on the masked lanes, state left over from a call of another size."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_masked_lane_soak.py"), "40", "base"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
