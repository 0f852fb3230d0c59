"""GPU probe: how should ONE wm_transcribe_greedy call of B chunks be cut into decode groups over the lanes?  Audio-s/s of the
whole call (front end + encoder + 224-token decode), min of 2 after a warm-up, for a list of policies (columns):

    gc=N        unmasked lanes, preferred group size N (debug knob group_chunks; N = 128: one group up to 128 chunks)
    parts=P     P CU-masked sub-chip lanes (round 6: hipExtStreamCreateWithCUMask), a slice of the CUs of every XCD each
    split=P     P groups of ceil(B / P) chunks on P UNMASKED lanes (= gc=ceil(B / P))
    product     the product's own rule (no knob)

    python tools/gpu_group_policy_probe.py [model] [B,B,...] [col,col,...] [earlystop]
earlystop: every chunk gets a token budget of uniform(40..200) (seeded; wm_set_token_budgets) instead of the fixed 224 tokens --
a group runs until its LAST row is finished, so smaller groups stop earlier (the default policy was measured at fixed length).
Every column's tokens are compared with the first column's ('!' = differ: a launch-shape choice must never change a token)."""
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
    cols = (sys.argv[3] if len(sys.argv) > 3 else "gc=8,gc=128,parts=2,parts=3,product").split(",")
    early = len(sys.argv) > 4 and sys.argv[4] == "earlystop"
    lib = B.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.wmdbg_set_tuning(b"reset", 0)
    dims = B.MODEL_DIMS[model]
    rng = np.random.default_rng(1)
    nmax = max(sizes)
    pcm = np.round(np.clip(0.1 * rng.standard_normal((nmax, 480000)), -1, 1) * 32767).astype(np.int16)
    prompt = [50258, 50259, 50359, 50363] if dims["n_vocab"] >= 51865 else [50257, 50362]
    ctx = B.Context(dims, debug=True)
    ctx.init_synthetic(20240928, matrix_gain=4.0)
    ctx.finalize()
    dp = ctx.to_device(pcm)

    print("%s: audio-s/s of one call of B chunks by group policy (3 lanes available)%s" % (
        model, "; EARLY STOP: per-chunk token budgets uniform(40..200)" if early else ""))
    print("B    " + "".join("%-13s" % c for c in cols))
    for nb in sizes:
        row, ref = "%-4d " % nb, None
        for col in cols:
            lib.wmdbg_set_tuning(b"group_chunks", 0)
            lib.wmdbg_set_tuning(b"lane_parts", 0)
            if col.startswith("gc="):
                lib.wmdbg_set_tuning(b"group_chunks", int(col[3:]))
                lib.wmdbg_set_tuning(b"lane_parts", 1)
            elif col.startswith("split="):
                lib.wmdbg_set_tuning(b"group_chunks", -(-nb // int(col[6:])))
                lib.wmdbg_set_tuning(b"lane_parts", 1)
            elif col.startswith("parts="):
                lib.wmdbg_set_tuning(b"lane_parts", int(col[6:]))
            else:
                assert col == "product", col
            best = None
            for i in range(3):
                t0 = time.perf_counter()
                bud = np.random.default_rng(4000 + nb).integers(40, 201, size=nb) if early else None
                toks, _ = ctx.transcribe_greedy(dp, prompt, 224, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=nb, budgets=bud)
                dt = time.perf_counter() - t0
                if i and (best is None or dt < best):
                    best = dt
            ok = " " if ref is None or np.array_equal(ref, toks) else "!"
            ref = toks if ref is None else ref
            row += "%8.0f%s    " % (30.0 * nb / best, ok)
        print(row, flush=True)
    lib.wmdbg_set_tuning(b"reset", 0)
    ctx.dev_free(dp)
    ctx.close()


if __name__ == "__main__":
    main()
