"""GPU probe: what handing over HOST buffers costs (the reference ABI does; bench.py keeps PCM resident in HBM and `value` excludes
PCIe by contract).  One decode group of 56 chunks, int16 PCM: device-resident vs pageable host memory (H2D inside the call),
and the reference-ABI shape (f64, 480 000 doubles per chunk) through wm_logmel alone."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg

B = pkg.binding
dims = B.MODEL_DIMS["large-v2"]
ctx = B.Context(dims)
ctx.init_synthetic(20240928)
ctx.finalize()
n, new = 56, 224
rng = np.random.default_rng(0)
pcm = np.round(np.clip(0.1 * rng.standard_normal((n, 480000)), -1, 1) * 32767).astype(np.int16)
prompt = [50258, 50259, 50359, 50363]
dp = ctx.to_device(pcm)
for name, arg, kw in (("device-resident int16", dp, dict(mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=n)), ("host int16 (pageable)", pcm, {})):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.transcribe_greedy(arg, prompt, new, eot=-1, **kw)
        best = min(best, time.perf_counter() - t0)
    print("%-26s %8.1f ms per group of %d chunks = %.0f audio-s/s (one lane)" % (name, best * 1e3, n, 30.0 * n / best), flush=True)
x64 = (pcm[:8].astype(np.float64) / 32768.0)
for dt_name, arr in (("f64 host -> f64 host (reference ABI shape)", x64), ("int16 host -> f32 host", pcm[:8])):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.logmel(arr, out_dtype=np.float64 if arr.dtype == np.float64 else np.float32)
        best = min(best, time.perf_counter() - t0)
    print("wm_logmel 8 chunks, %-44s %7.2f ms (%.2f MB over PCIe each way in total)" % (dt_name, best * 1e3, (arr.nbytes + 8 * 80 * 3000 * (8 if arr.dtype == np.float64 else 4)) / 1e6), flush=True)
