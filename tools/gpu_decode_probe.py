"""GPU probe: per-launch time of the decode kernels at large-v2 geometry (B = argv[1], default 8)."""
import ctypes, sys
BQ = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
c = pkg.binding.Context(debug=True)
lib = c.lib
us = ctypes.c_float()
e_us, g_us = ctypes.c_float(), ctypes.c_float()
for grid in (1, 256, 1024):
    lib.wmdbg_bench_launch_floor(c.handle, 500, grid, ctypes.byref(e_us), ctypes.byref(g_us))
    print('launch floor grid=%d: eager %.2f us/kernel, graph %.2f us/kernel' % (grid, e_us.value, g_us.value))
def gemv(B, N, K, ln, resid, nw=0, mats=32, iters=320):
    st = lib.wmdbg_bench_dec_gemv(c.handle, B, N, K, ln, resid, mats, iters, nw, ctypes.byref(us))
    gb = N * K * 2 / 1e9
    return "%6.2f us  %6.0f GB/s" % (us.value, gb / (us.value * 1e-6)) if st == 0 else "ERR " + lib.wm_last_error().decode()
for (name, N, K, ln, resid) in [("ln_qkv", 3840, 1280, 1, 0), ("attn_out", 1280, 1280, 0, 1), ("ln_q", 1280, 1280, 1, 0),
                                ("ln_fc1", 5120, 1280, 1, 0), ("fc2", 1280, 5120, 0, 1), ("logits", 51865, 1280, 1, 0)]:
    line = "%-9s N=%5d K=%4d:" % (name, N, K)
    for mats in (1, 2, 4, 64):
        if N > 10000 and mats > 4: continue
        line += "  mats=%d %s |" % (mats, gemv(BQ, N, K, ln, resid, 0, mats=mats, iters=(40 if N > 10000 else 320)))
    print(line)
def attn(B, H, T, nk, ns, slices=8, iters=160):
    st = lib.wmdbg_bench_dec_attention(c.handle, B, H, T, nk, ns, slices, iters, ctypes.byref(us))
    gb = B * H * nk * 64 * 2 * 2 / 1e9
    return "%6.2f us  %6.0f GB/s" % (us.value, gb / (us.value * 1e-6)) if st == 0 else "ERR"
for ns in (1, 2, 4):
    print("cross attn B=%d H=20 keys=1500 nsplit=%d: %s" % (BQ, ns, attn(BQ, 20, 1500, 1500, ns)))
for nk in (1, 64, 224, 448):
    print("self  attn B=%d H=20 keys=%4d nsplit=1: %s" % (BQ, nk, attn(BQ, 20, 448, nk, 1)))
