"""GPU soak of the sub-chip-lane path (round 6): a NARROW model (base, full depth) decodes calls of changing size through the
product's own policy -- 24 .. 128 chunks: two CU-masked half-chip groups; below: one group on the context's stream -- with and
without early stop, in a random order, many times; every result must equal the reference decoded ONCE per (size, stop mode) as a
single group (wm_set_lanes(1)).  Catches what a one-shot test cannot: graph re-use across alternating shapes on the masked
lanes, lane state left over from a call of another size, budgets consumed by the wrong call.

    python tools/gpu_masked_lane_soak.py [rounds=60] [model=base]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    model = sys.argv[2] if len(sys.argv) > 2 else "base"
    import importlib
    W = importlib.import_module("openai_whisper_coreml_amd.weights")
    dims = B.MODEL_DIMS[model]
    ctx = B.Context(dims)
    ctx.init_synthetic(20240928, matrix_gain=W.lively_gain(dims))
    ctx.finalize()
    rng = np.random.default_rng(11)
    nmax = 72
    pcm = np.round(np.clip(0.1 * rng.standard_normal((nmax, 480000)), -1, 1) * 32767).astype(np.int16)
    dp = ctx.to_device(pcm)
    prompt = [50258, 50259, 50359, 50363] if dims["n_vocab"] >= 51865 else [50257, 50362]
    NEW = 64
    sizes = [8, 16, 24, 31, 32, 40, 64, 72]
    ref = {}

    def run(nb, stop, lanes):
        ctx.set_lanes(lanes)
        bud = np.random.default_rng(100 + nb).integers(5, NEW + 1, size=nb) if stop else None
        t, l = ctx.transcribe_greedy(dp, prompt, NEW, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=nb, budgets=bud)
        ctx.set_lanes(0)
        return t, l
    for nb in sizes:
        for stop in (False, True):
            ref[(nb, stop)] = run(nb, stop, 1)
    t0 = time.perf_counter()
    bad = 0
    for i in range(rounds):
        nb = int(rng.choice(sizes))
        stop = bool(rng.integers(0, 2))
        t, l = run(nb, stop, 0)
        rt, rl = ref[(nb, stop)]
        ok = np.array_equal(l, rl) and all(np.array_equal(t[r, :l[r]], rt[r, :rl[r]]) for r in range(nb))
        bad += not ok
        if not ok:
            print("MISMATCH round %d: %d chunks, early stop %s" % (i, nb, stop), flush=True)
    print("%s: %d calls of %s chunks through the product policy (masked lanes from 24), early stop on / off at random: %d mismatches, %.1f s"
          % (model, rounds, sizes, bad, time.perf_counter() - t0))
    ctx.dev_free(dp)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
