"""Race screen for the 256 x 256 staggered-phase GEMM (counted vmcnt, asm ds_reads): the 128 x 128 kernel accumulates in the
same order, so on the same operands the two must agree BIT FOR BIT.  Fresh random operands every iteration, several shapes
(K-tile counts 2 .. 80, M / N tails), both epilogue families; run it next to another GPU job to perturb the timing."""
import ctypes, sys
sys.path.insert(0, '.')
import numpy as np
import openai_whisper_coreml_amd as pkg
ctx = pkg.binding.Context(debug=True); lib = ctx.lib
vp, ip = ctypes.c_void_p, ctypes.c_int
lib.wmdbg_gemm.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip]
lib.wmdbg_set_gemm_tile.argtypes = [ip]
P = lambda a: a.ctypes.data_as(vp)
def bf(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).view(np.float32)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(2048, 2048, 1280), (1536, 5120, 1280), (3000, 1280, 5120), (1031, 768, 448), (777, 1280, 128), (2600, 2560, 3840)]
bad = 0
for it in range(iters):
    for (M, N, K) in shapes:
        rng = np.random.default_rng(1000 * it + M + N + K)
        A = bf(rng.standard_normal((M, K))); W = bf(rng.standard_normal((N, K)) * 0.05)
        bias = rng.standard_normal(N).astype(np.float32)
        for epi in (6, 0, 2):
            outs = []
            for tile in (128, 256):
                lib.wmdbg_set_gemm_tile(tile)
                C = np.full((M, N), 0.25, np.float32)
                assert lib.wmdbg_gemm(ctx.handle, P(A), P(W), P(bias), P(C), M, N, K, epi) == 0
                outs.append(C)
            if not np.array_equal(outs[0], outs[1]):
                bad += 1
                d = np.argwhere(outs[0] != outs[1])
                print("MISMATCH it=%d shape=%s epi=%d: %d elements, first at %s" % (it, (M, N, K), epi, len(d), d[0]))
    if it % 5 == 4: print("iteration", it + 1, "mismatches so far", bad, flush=True)
lib.wmdbg_set_gemm_tile(0)
print("race screen done: %d iterations x %d shapes x 3 epilogues, mismatches %d" % (iters, len(shapes), bad))
sys.exit(1 if bad else 0)
