"""GPU probe: the K = 4d residual product (fc2) at more than one batch block -- one block per workgroup (product) vs two
(knob gemv_ppw2_nblk=2), large-v2 / medium / small shapes."""
import ctypes, sys
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
c = pkg.binding.Context(debug=True)
lib = c.lib
us = ctypes.c_float()
def gemv(B, N, K, mats=32, iters=640):
    st = lib.wmdbg_bench_dec_gemv(c.handle, B, N, K, 0, 1, mats, iters, 0, ctypes.byref(us))
    return us.value if st == 0 else float('nan')
for (d, name) in [(1280, "large"), (1024, "medium"), (768, "small")]:
    for B in (24, 32, 48, 53, 56, 64, 80, 96, 128):
        r = []
        for knob in (0, 2, 0, 2):
            assert lib.wmdbg_set_tuning(b"gemv_ppw2_nblk", knob) == 0
            r.append(gemv(B, d, 4 * d))
        print("%-6s fc2 rows=%3d  one block %6.2f %6.2f us   two blocks %6.2f %6.2f us" % (name, B, r[0], r[2], r[1], r[3]), flush=True)
assert lib.wmdbg_set_tuning(b"gemv_ppw2_nblk", 0) == 0
def gemv_ln(B, N, K, mats=32, iters=640):
    st = lib.wmdbg_bench_dec_gemv(c.handle, B, N, K, 1, 0, mats, iters, 0, ctypes.byref(us))
    return us.value if st == 0 else float('nan')
for (name, N, K) in [("fc1 large", 5120, 1280), ("qkv large", 3840, 1280), ("fc1 small", 3072, 768), ("qkv small", 2304, 768)]:
    for B in (40, 53, 64, 72, 96, 128):
        r = []
        for knob in (1, 0, 1, 0):
            assert lib.wmdbg_set_tuning(b"gemv_no_tn3", knob) == 0
            r.append(gemv_ln(B, N, K))
        print("%-10s rows=%3d  groups of 1/2/4 %6.2f %6.2f us   with 3 %6.2f %6.2f us" % (name, B, r[0], r[2], r[1], r[3]), flush=True)
