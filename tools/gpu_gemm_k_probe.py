import sys, ctypes
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
ctx = pkg.binding.Context(debug=True); lib = ctx.lib
lib.wmdbg_set_gemm_tile.argtypes = [ctypes.c_int]
lib.wmdbg_bench_gemm.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
lib.wmdbg_set_gemm_tile(256)
for (M, N) in [(12000, 5120), (12000, 1280), (96000, 5120)]:
    for epi, name in [(0, "bf16"), (1, "gelu"), (6, "f32"), (2, "resid")]:
        row = "M=%6d N=%5d %-5s" % (M, N, name)
        for K in (1280, 2560, 5120):
            if M * K > 96000 * 2560: continue
            us = ctypes.c_float()
            assert lib.wmdbg_bench_gemm(ctx.handle, M, N, K, epi, 16, 8, ctypes.byref(us)) == 0
            row += "  K=%4d %8.1f us %5.0f TF" % (K, us.value, 2.0 * M * N * K / us.value / 1e6)
        print(row)
