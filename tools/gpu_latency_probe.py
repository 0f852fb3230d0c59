"""GPU probe: decode latency (ms per position) of small decode groups on ONE lane, by model and batch, under the debug
library's launch-shape knobs (wmdbg_set_tuning): the latency regime of BASELINE.json configs[1] / [3].

    python tools/gpu_latency_probe.py [key=value,...;key=value,...]   (each ';'-separated set is one column)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def run(model, nb, tuning, new=224):
    lib = B.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.wmdbg_set_tuning(b"reset", 0)
    for k, v in tuning.items():
        assert lib.wmdbg_set_tuning(k.encode(), int(v)) == 0, k
    dims = B.MODEL_DIMS[model]
    ctx = B.Context(dims, debug=True)
    ctx.init_synthetic(20240928, matrix_gain=4.0)
    ctx.finalize()
    ctx.set_lanes(1)
    rng = np.random.default_rng(1)
    pcm = np.round(np.clip(0.1 * rng.standard_normal((nb, 480000)), -1, 1) * 32767).astype(np.int16)
    dp = ctx.to_device(pcm)
    prompt = [50258, 50259, 50359, 50363] if dims["n_vocab"] >= 51865 else [50257, 50362]
    best = None
    for i in range(4):
        toks, _ = ctx.transcribe_greedy(dp, prompt, new, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=nb)
        ms = float(ctx.last_stage_ms()[2]) / (len(prompt) + new - 1)
        if i and (best is None or ms < best):
            best = ms
    ctx.dev_free(dp)
    ctx.close()
    lib.wmdbg_set_tuning(b"reset", 0)
    return best, toks


def main():
    sets = [{}]
    if len(sys.argv) > 1:
        sets = [dict(kv.split("=") for kv in s.split(",") if kv) for s in sys.argv[1].split(";")]
    print("decode ms per position; columns: " + " | ".join(str(s) for s in sets))
    cfgs = (("tiny.en", 1), ("tiny.en", 8), ("base", 1), ("base", 8), ("small", 1), ("large-v2", 1), ("large-v2", 2),
            ("large-v2", 4), ("large-v2", 8), ("large-v2", 12), ("large-v3", 15), ("large-v2", 24))
    if len(sys.argv) > 2:   # model:batch,model:batch,...
        cfgs = tuple((m, int(b)) for m, b in (c.split(":") for c in sys.argv[2].split(",")))
    for model, nb in cfgs:
        row, ref = [], None
        for s in sets:
            ms, toks = run(model, nb, s)
            same = "" if ref is None else (" same" if np.array_equal(ref, toks) else " DIFFERENT TOKENS")
            ref = toks if ref is None else ref
            row.append("%.4f%s" % (ms, same))
        print("%-10s B=%-3d %s" % (model, nb, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
