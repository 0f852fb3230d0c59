"""GPU: measured parity margins of the product (bf16) path and of the all-fp32 debug path against the fp32 oracle, at every
geometry the parity tests run -- the figures the tolerances in tests/test_model_gpu.py are set from (<= 3x the worst
measured value).  Writes a table to stdout (committed as profiles/r04_parity_margins.txt).

    python tools/gpu_parity_margins.py [--quick]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import openai_whisper_coreml_amd as pkg  # noqa: E402
from oracle import logmel_np as L  # noqa: E402
from oracle import whisper_ref as R  # noqa: E402

W = importlib.import_module("openai_whisper_coreml_amd.weights")
B = pkg.binding


def tone_chunk(i):
    n = np.arange(480000, dtype=np.float64)
    x = 0.3 * np.sin(2 * np.pi * (200 + 370 * i) * n / 16000) * (0.5 + 0.5 * np.sin(2 * np.pi * (0.3 + 0.1 * i) * n / 16000))
    return x.astype(np.float32)


def perturb_ln(ctx, dims, seed=3):
    rng = np.random.default_rng(seed)
    for name, shape, kind in W.tensor_specs(dims):
        if kind == W.K_LN_W:
            ctx.set_tensor(name, (1 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif kind == W.K_LN_B:
            ctx.set_tensor(name, (0.1 * rng.standard_normal(shape)).astype(np.float32))


def measure(tag, dims, seed, gain, n_chunks, tokens):
    ctx = B.Context(dims, debug=True)
    ctx.init_synthetic(seed, matrix_gain=gain)
    perturb_ln(ctx, dims)
    ctx.finalize()
    sd = R.to_torch({n: ctx.get_tensor(n, s) for n, s, _ in W.tensor_specs(dims)})
    pcm = np.stack([tone_chunk(i) if i % 2 else L.synth_chunk(20 + i) for i in range(n_chunks)])
    mel = ctx.logmel(pcm, n_mels=dims["n_mels"], out_dtype=np.float32)
    t0 = time.time()
    want = R.encode(sd, dims, mel).numpy()
    tok = np.tile(np.asarray(tokens, np.int32), (n_chunks, 1))
    ref = R.decode_logits(sd, dims, tok, want).numpy()
    t_cpu = time.time() - t0
    out = {}
    for prec in ("bf16", "f32"):
        ctx.set_precision(prec == "f32")
        xa = ctx.encode_mel(mel)
        lg = ctx.decode_logits(tok, want)
        out[prec] = (R.rel_l2(xa, want), R.rel_l2(lg, ref), float(np.abs(lg - ref).max()),
                     float((lg.argmax(-1) == ref.argmax(-1)).mean()))
    ctx.close()
    print("%-34s enc L %2d dec L %2d d %4d gain %g | bf16: enc %.2e logits %.2e (max abs %.3g, argmax agree %.3f) | "
          "f32: enc %.2e logits %.2e (max abs %.3g, argmax agree %.3f) | logit rms %.2f | oracle %.1f s"
          % (tag, dims["n_audio_layer"], dims["n_text_layer"], dims["n_text_state"], gain, *out["bf16"], *out["f32"],
             float(np.sqrt((ref.astype(np.float64) ** 2).mean())), t_cpu), flush=True)
    return out


def main():
    quick = "--quick" in sys.argv
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    MD = B.MODEL_DIMS
    big_tok = [50258, 50259, 50359, 50363, 1000, 2000]
    cases = [("tiny dims (tests' TINY_DIMS)", dict(R.TINY_DIMS), 11, 1.0, 2, [1, 7, 300, 1023, 5, 9]),
             ("tiny dims, lively", dict(R.TINY_DIMS), 11, 4.0, 2, [1, 7, 300, 1023, 5, 9]),
             ("tiny.en full", MD["tiny.en"], 7, 1.0, 1, [50257, 50362, 100, 2000]),
             ("base full", MD["base"], 3, 1.0, 1, big_tok),
             ("small 2+2 layers", dict(MD["small"], n_audio_layer=2, n_text_layer=2), 19, 1.0, 1, big_tok),
             ("large-v2 2+2 layers, lively", dict(MD["large-v2"], n_audio_layer=2, n_text_layer=2), 29, 4.0, 1, big_tok)]
    if not quick:
        cases += [("small full (the reference's model)", MD["small"], 19, 1.0, 1, big_tok),
                  ("large-v2 full", MD["large-v2"], 7, 1.0, 1, big_tok),
                  ("large-v2 full, lively", MD["large-v2"], 20240928, 4.0, 1, big_tok),
                  ("large-v3 full", MD["large-v3"], 23, 1.0, 1, [50258, 50259, 50360, 50364, 1000, 2000])]
    print("# rel-L2 against oracle/whisper_ref.py (fp32, CPU) on the GPU's own weights; logits teacher-forced on the ORACLE's "
          "encoder output; LayerNorm parameters perturbed")
    for c in cases:
        measure(*c)


if __name__ == "__main__":
    main()
