"""GPU check at production scale: early stop == decode everything and truncate.  large-v2 at FULL depth, one decode group
of 56 chunks and a 168-chunk call (three lanes), per-chunk budgets uniform(8..60): the same process decodes with the stop
machinery ON; a child process on the DEBUG library with wmdbg_set_tuning("no_early_stop", 1) decodes every position and
truncates on the host (the round-2 behaviour); tokens and lengths must be identical.  Prints the decode times of both."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, '.')


def run(tag):
    import openai_whisper_coreml_amd as pkg
    B = pkg.binding
    dims = B.MODEL_DIMS["large-v2"]
    ctx = B.Context(dims, debug=(tag == "truncate"))
    if tag == "truncate":
        import ctypes
        ctx.lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        assert ctx.lib.wmdbg_set_tuning(b"no_early_stop", 1) == 0
    ctx.init_synthetic(20240928)
    ctx.finalize()
    prompt = [50258, 50259, 50359, 50363]
    out = {}
    for n in (56, 168):
        rng = np.random.default_rng(n)
        pcm = np.round(np.clip(0.1 * rng.standard_normal((8, 480000)), -1, 1) * 32767).astype(np.int16)[np.arange(n) % 8]
        dp = ctx.to_device(pcm)
        bud = np.random.default_rng(7 + n).integers(8, 61, size=n)
        toks, lens = ctx.transcribe_greedy(dp, prompt, 64, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=n, budgets=bud)
        out["t%d" % n], out["l%d" % n] = toks, lens
        print("%s: %3d chunks, budgets 8..60: decode stage %.1f ms, lens %d..%d" % (tag, n, ctx.last_stage_ms()[2], lens.min(), lens.max()), flush=True)
        ctx.dev_free(dp)
    np.savez("/tmp/es_%s.npz" % tag, **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        run("early_stop")
        subprocess.run([sys.executable, __file__, "truncate"], check=True)
        a, b = np.load("/tmp/es_early_stop.npz"), np.load("/tmp/es_truncate.npz")
        for k in a.files:
            assert np.array_equal(a[k], b[k]), k
        print("identical tokens and lengths: early stop == decode-all-and-truncate at full depth", flush=True)
