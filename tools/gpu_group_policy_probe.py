"""GPU probe: how should ONE wm_transcribe_greedy call of B chunks be cut into decode groups over the lanes?  Sweeps the
preferred group size (debug knob group_chunks; the product's rule is in model_api.cpp) for call sizes between one
and three groups' worth: audio-s/s of the whole call (front end + encoder + 224-token decode), min of 2 after a warm-up.

    python tools/gpu_group_policy_probe.py [model] [B,B,...] [gc,gc,...]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "large-v2"
    sizes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "9,12,15,16,20,24,32,48").split(",")]
    gcs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8,12,16,24,128").split(",")]
    lib = B.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.wmdbg_set_tuning(b"reset", 0)
    dims = B.MODEL_DIMS[model]
    ctx = B.Context(dims, debug=True)
    ctx.init_synthetic(20240928, matrix_gain=4.0)
    ctx.finalize()
    rng = np.random.default_rng(1)
    nmax = max(sizes)
    pcm = np.round(np.clip(0.1 * rng.standard_normal((nmax, 480000)), -1, 1) * 32767).astype(np.int16)
    dp = ctx.to_device(pcm)
    prompt = [50258, 50259, 50359, 50363]
    print("%s: audio-s/s of one call of B chunks by preferred group size (3 lanes); groups shown as n x size" % model)
    print("B    " + "".join("gc=%-12d" % g for g in gcs))
    for nb in sizes:
        row, ref = "%-4d " % nb, None
        for gc in gcs:
            assert lib.wmdbg_set_tuning(b"group_chunks", gc) == 0
            best = None
            for i in range(3):
                t0 = time.perf_counter()
                toks, _ = ctx.transcribe_greedy(dp, prompt, 224, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=nb)
                dt = time.perf_counter() - t0
                if i and (best is None or dt < best):
                    best = dt
            ok = "" if ref is None or np.array_equal(ref, toks) else "!"
            ref = toks if ref is None else ref
            g = -(-nb // gc) if nb <= gc * 3 else max(3, -(-nb // 128))
            row += "%7.0f%s (%dx%d) " % (30.0 * nb / best, ok, g, -(-nb // g))
        print(row, flush=True)
    lib.wmdbg_set_tuning(b"reset", 0)
    ctx.dev_free(dp)
    ctx.close()


if __name__ == "__main__":
    main()
