"""GPU probe: encoder-side kernel families at large-v2 geometry, B=8 (event-timed)."""
import sys, json
sys.path.insert(0, '.')
import numpy as np
import openai_whisper_coreml_amd as pkg
B = pkg.binding
dims = B.MODEL_DIMS["large-v2"]
ctx = B.Context(dims); ctx.init_synthetic(1); ctx.finalize()
mel = np.random.default_rng(0).standard_normal((8, 80, 3000)).astype(np.float32) * 0.3
d_mel = ctx.to_device(mel); d_xa = ctx.dev_malloc(8 * 1500 * 1280 * 4)
for it in range(2): ctx.lib.wm_encode(ctx.handle, d_mel, 8, d_xa, 1)
ctx.sync(); ctx.profile_reset(); ctx.profile_enable(True)
for it in range(3): ctx.lib.wm_encode(ctx.handle, d_mel, 8, d_xa, 1)
p = ctx.profile(); ctx.profile_enable(False)
tot = 0
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-18s %8.3f ms/pass  n=%3d  avg %8.2f us" % (k, v["ms"] / 3, v["n"] / 3, v["ms"] / v["n"] * 1e3)); tot += v["ms"] / 3
print("encoder total (event sum) %.2f ms" % tot)
