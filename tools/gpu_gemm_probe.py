"""GPU probe: encoder GEMM shapes through the 128x128 and 256x256 kernels (random operands, back-to-back launches)."""
import sys, ctypes
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
B = pkg.binding
ctx = B.Context(debug=True)
lib = ctx.lib
lib.wmdbg_set_gemm_tile.argtypes = [ctypes.c_int]
lib.wmdbg_bench_gemm.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
EPI = {"f32": 6, "bf16": 0, "gelu": 1, "resid": 2}
shapes = [(4096, 4096, 4096, "f32"), (8192, 8192, 8192, "bf16"), (12000, 5120, 1280, "gelu"), (12000, 5120, 1280, "bf16"),
          (12000, 1280, 5120, "resid"), (12000, 3840, 1280, "bf16"), (12000, 1280, 1280, "resid"), (12000, 2560, 1280, "bf16")]
tiles = [int(a) for a in sys.argv[1:]] or [128, 256]
for M, N, K, e in shapes:
  for n_w in (1, 32):
    if n_w > 1 and M != 12000:
        continue
    row = "%6d x %5d x %5d %-6s W x%-2d" % (M, N, K, e, n_w)
    for tile in tiles:
        lib.wmdbg_set_gemm_tile(tile)
        us = ctypes.c_float()
        st = lib.wmdbg_bench_gemm(ctx.handle, M, N, K, EPI[e], 32, n_w, ctypes.byref(us))
        assert st == 0, lib.wm_last_error()
        row += "   tile %3d: %8.1f us %7.0f TF/s" % (tile, us.value, 2.0 * M * N * K / us.value / 1e6)
    print(row)
