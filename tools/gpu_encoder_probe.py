"""GPU probe: encoder-side kernel families at large-v2 geometry, B=8 (event-timed), for each GEMM tile choice."""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
import openai_whisper_coreml_amd as pkg
B = pkg.binding
dims = B.MODEL_DIMS["large-v2"]
ctx = B.Context(dims, debug=True); ctx.init_synthetic(1); ctx.finalize()
ctx.lib.wmdbg_set_gemm_tile.argtypes = [ctypes.c_int]
ctx.lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
for kv in [a for a in sys.argv[1:] if "=" in a]:
    k, v = kv.split("=")
    assert ctx.lib.wmdbg_set_tuning(k.encode(), int(v)) == 0
sys.argv = [a for a in sys.argv if "=" not in a]
mel = np.random.default_rng(0).standard_normal((8, 80, 3000)).astype(np.float32) * 0.3
d_mel = ctx.to_device(mel); d_xa = ctx.dev_malloc(8 * 1500 * 1280 * 4)
out = {}
for tile in [int(a) for a in sys.argv[1:]] or [128, 0]:
    ctx.lib.wmdbg_set_gemm_tile(tile)
    for it in range(2): ctx.lib.wm_encode(ctx.handle, d_mel, 8, d_xa, 1)
    ctx.sync()
    t0 = time.perf_counter()
    for it in range(5): ctx.lib.wm_encode(ctx.handle, d_mel, 8, d_xa, 1)
    ctx.sync()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    out[tile] = ctx.download(d_xa, (8, 1500, 1280), np.float32)
    ctx.profile_reset(); ctx.profile_enable(True)
    for it in range(3): ctx.lib.wm_encode(ctx.handle, d_mel, 8, d_xa, 1)
    p = ctx.profile(); ctx.profile_enable(False)
    print("=== gemm tile %s: encoder wall %.2f ms (2272.7 GF/chunk x 8 -> %.0f TF/s)" % (tile or "auto", wall, 8 * 2272.7 / wall))
    tot = 0
    for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"]):
        print("%-18s %8.3f ms/pass  n=%3d  avg %8.2f us" % (k, v["ms"] / 3, v["n"] / 3, v["ms"] / v["n"] * 1e3)); tot += v["ms"] / 3
    print("encoder total (event sum) %.2f ms" % tot)
ks = list(out)
if len(ks) > 1:
    a, b = out[ks[0]], out[ks[1]]
    print("tile %s vs %s: max abs diff %.3g, rel-L2 %.3g" % (ks[0], ks[1], np.abs(a - b).max(), np.linalg.norm(a - b) / np.linalg.norm(a)))
