"""GPU helper for profilers: run ONLY the encoder (wm_encode, device memory) of a model on B synthetic chunks, n times.
rocprofv3 wraps this to get kernel traces / PMC passes of the encoder kernels at a chosen geometry without a decode.

    python tools/gpu_encode_only.py [model=large-v2] [B=56] [reps=2] [knob=value ...]   (knobs: wmdbg_set_tuning, debug library)"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def main():
    knobs = [a for a in sys.argv[1:] if "=" in a]
    pos = [a for a in sys.argv[1:] if "=" not in a]
    model = pos[0] if pos else "large-v2"
    nb = int(pos[1]) if len(pos) > 1 else 56
    reps = int(pos[2]) if len(pos) > 2 else 2
    dims = B.MODEL_DIMS[model]
    ctx = B.Context(dims, debug=bool(knobs))
    if knobs:
        ctx.lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        for kv in knobs:
            k, v = kv.split("=")
            assert ctx.lib.wmdbg_set_tuning(k.encode(), int(v)) == 0, kv
    ctx.init_synthetic(1)
    ctx.finalize()
    mel = (np.random.default_rng(0).standard_normal((nb, dims["n_mels"], 3000)) * 0.3).astype(np.float32)
    d_mel = ctx.to_device(mel)
    d_xa = ctx.dev_malloc(nb * 1500 * dims["n_audio_state"] * 4)
    ctx.lib.wm_encode(ctx.handle, d_mel, nb, d_xa, 1)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        assert ctx.lib.wm_encode(ctx.handle, d_mel, nb, d_xa, 1) == 0
    ctx.sync()
    print("%s x %d: encoder %.3f ms per call" % (model, nb, (time.perf_counter() - t0) / reps * 1e3))
    ctx.dev_free(d_mel)
    ctx.dev_free(d_xa)
    ctx.close()


if __name__ == "__main__":
    main()
