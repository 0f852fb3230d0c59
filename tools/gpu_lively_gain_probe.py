"""GPU probe: which matrix gain makes the random-init model of each geometry "lively" -- N distinct recordings decode to N
distinct token rows whose tokens depend on the decode history?  (The gain-4 recipe calibrated at d = 1280 is NOT lively at
d = 512: bench.py's base batch-32 entry reported 4 distinct rows of 32 distinct recordings, VERDICT r5 weak #8.)

    python tools/gpu_lively_gain_probe.py [model,model,...] [gain,gain,...] [chunks] [tokens]
Prints per (model, gain): distinct rows / chunks, mean distinct tokens per row, and whether a row decoded alone equals itself in the batch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def synth_pcm16(i):
    rng = np.random.default_rng(1234 + i)
    return np.round(np.clip(0.1 * rng.standard_normal(480000), -1.0, 1.0).astype(np.float32) * 32767).astype(np.int16)


def tone_pcm16(i):
    n = np.arange(480000, dtype=np.float64)
    x = 0.3 * np.sin(2 * np.pi * (200 + 370 * i) * n / 16000) * (0.5 + 0.5 * np.sin(2 * np.pi * (0.3 + 0.1 * i) * n / 16000))
    return np.round(x * 32767).astype(np.int16)


def main():
    models = (sys.argv[1] if len(sys.argv) > 1 else "tiny.en,base,small").split(",")
    gains = [float(g) for g in (sys.argv[2] if len(sys.argv) > 2 else "4,6,8,12,16").split(",")]
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    n_new = int(sys.argv[4]) if len(sys.argv) > 4 else 224
    noise = np.stack([synth_pcm16(i) for i in range(nb)])
    mixed = np.stack([tone_pcm16(i) if i % 2 else synth_pcm16(i) for i in range(nb)])
    for model in models:
        dims = B.MODEL_DIMS[model]
        prompt = [50258, 50259, 50359, 50363] if dims["n_vocab"] >= 51865 else [50257, 50362]
        for g in gains:
            ctx = B.Context(dims)
            ctx.init_synthetic(20240928, matrix_gain=g)
            ctx.finalize()
            ctx.set_lanes(1)
            out = []
            for name, pcm in (("noise", noise), ("tones+noise", mixed)):
                toks, _ = ctx.transcribe_greedy(pcm, prompt, n_new)
                solo, _ = ctx.transcribe_greedy(pcm[5:6], prompt, n_new)
                rows = len({r.tobytes() for r in toks})
                per = np.mean([len(set(r.tolist())) for r in toks])
                out.append("%s: %2d/%d rows distinct, %.1f distinct tokens/row, alone==batch %s" % (
                    name, rows, nb, per, bool(np.array_equal(solo[0], toks[5]))))
            print("%-9s gain %-5g %s" % (model, g, " | ".join(out)), flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
