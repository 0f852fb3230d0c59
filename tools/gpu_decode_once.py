"""GPU helper for profilers: ONE wm_transcribe_greedy call (after one warm-up call) of a model on B synthetic chunks with
the debug library's knobs applied -- rocprofv3 wraps this (WM_NO_GRAPH=1: eager launches are traceable).

    python tools/gpu_decode_once.py model B new_tokens [knob=value ...]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def main():
    model, nb, new = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lib = B.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.wmdbg_set_tuning(b"reset", 0)
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        assert lib.wmdbg_set_tuning(k.encode(), int(v)) == 0, kv
    dims = B.MODEL_DIMS[model]
    ctx = B.Context(dims, debug=True)
    ctx.init_synthetic(20240928, matrix_gain=4.0)
    ctx.finalize()
    rng = np.random.default_rng(1)
    pcm = np.round(np.clip(0.1 * rng.standard_normal((nb, 480000)), -1, 1) * 32767).astype(np.int16)
    dp = ctx.to_device(pcm)
    prompt = [50258, 50259, 50359, 50363] if dims["n_vocab"] >= 51865 else [50257, 50362]
    for i in range(2):
        t0 = time.perf_counter()
        ctx.transcribe_greedy(dp, prompt, new, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=nb)
        dt = time.perf_counter() - t0
    print("%s x %d, %d tokens, %s: %.2f ms, decode %.4f ms/position" % (
        model, nb, new, " ".join(sys.argv[4:]) or "product", dt * 1e3, float(ctx.last_stage_ms()[2]) / (len(prompt) + new - 1)))
    ctx.dev_free(dp)
    ctx.close()


if __name__ == "__main__":
    main()
