"""GPU probe: the encoder's six GEMM shapes at the driver run's size (M = 84 000 = 56 chunks) through both tile kernels."""
import sys, ctypes
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
B = pkg.binding
ctx = B.Context(debug=True)
lib = ctx.lib
lib.wmdbg_set_gemm_tile.argtypes = [ctypes.c_int]
lib.wmdbg_bench_gemm.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
EPI = {"f32": 6, "bf16": 0, "gelu": 1, "resid": 2}
M = int(sys.argv[1]) if len(sys.argv) > 1 else 84000
for N, K, e, what in ((1280, 1280, "resid", "out-proj"), (1280, 5120, "resid", "fc2"), (5120, 1280, "gelu", "fc1"), (3840, 1280, "bf16", "qkv-like"), (2560, 1280, "bf16", "xkv-like")):
    row = "%-9s %6d x %5d x %5d %-6s" % (what, M, N, K, e)
    for tile in (128, 256):
        lib.wmdbg_set_gemm_tile(tile)
        us = ctypes.c_float()
        assert lib.wmdbg_bench_gemm(ctx.handle, M, N, K, EPI[e], 16, 8, ctypes.byref(us)) == 0, lib.wm_last_error()
        row += "   tile %3d: %8.1f us %6.0f TF/s" % (tile, us.value, 2.0 * M * N * K / us.value / 1e6)
    print(row, flush=True)
