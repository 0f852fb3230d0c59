"""GPU unit tests: each hand-written HIP kernel of boundary #2 against a plain fp32/fp64
numpy / torch restatement of the same op, through the wmdbg_* hooks of the C ABI
(include/whisper_mi355x_debug.h).  Inputs are rounded to bf16 the same way HBM holds them,
so the tolerances below measure the kernel (fp32 accumulate, bf16 outputs), not the input
quantisation.  Asymmetric operands everywhere: a transposed fragment layout must fail."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W = importlib.import_module("openai_whisper_coreml_amd.weights")


def bf(x):
    return W.bf16_round_f32(np.ascontiguousarray(x, dtype=np.float32))


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.binding.Context(debug=True)   # libwhisper_mi355x_dbg.so: the product objects + the wmdbg_* hooks
    lib = c.lib
    vp, ip = ctypes.c_void_p, ctypes.c_int
    lib.wmdbg_gemm.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip]
    lib.wmdbg_layernorm.argtypes = [vp, vp, vp, vp, ip, ip, vp, vp]
    lib.wmdbg_enc_attention.argtypes = [vp, vp, vp, vp, ip, ip, ip, vp]
    lib.wmdbg_dec_gemv.argtypes = [vp, vp, vp, vp, vp, vp, vp, ip, ip, ip]
    lib.wmdbg_dec_attention.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    lib.wmdbg_dec_gemv_resid.argtypes = [vp, vp, vp, vp, vp, vp, vp, ip, ip, ip]
    yield c
    c.close()


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))


@pytest.fixture(params=[64, 128, 256])
def gemm_tile(ctx, request):
    """Run a GEMM test through the three encoder GEMM kernels (64 x 64 for single-chunk products, 128 x 128 and the
    256 x 256 staggered-phase one)."""
    ctx.lib.wmdbg_set_gemm_tile.argtypes = [ctypes.c_int]
    assert ctx.lib.wmdbg_set_gemm_tile(request.param) == 0
    yield request.param
    ctx.lib.wmdbg_set_gemm_tile(0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 256, 192), (1500, 128, 1280),
                                   (77, 512, 64), (1031, 768, 448), (512, 1280, 5120)])
def test_gemm_f32_out(ctx, gemm_tile, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = bf(rng.standard_normal((M, K)))
    Wt = bf(rng.standard_normal((N, K)) * 0.1 + np.linspace(-0.05, 0.05, N)[:, None])  # asymmetric
    bias = rng.standard_normal(N).astype(np.float32)
    C = np.zeros((M, N), np.float32)
    st = ctx.lib.wmdbg_gemm(ctx.handle, P(A), P(Wt), P(bias), P(C), M, N, K, 6)
    assert st == 0, ctx.lib.wm_last_error()
    ref = A.astype(np.float64) @ Wt.astype(np.float64).T + bias
    assert np.abs(C - ref).max() <= 2e-4 * np.abs(ref).max(), (np.abs(C - ref).max(), np.abs(ref).max())


def test_gemm_identity_asymmetric(ctx, gemm_tile):
    """A = I picks rows of W^T: any row<->col swap in the C write shows up exactly."""
    M = N = K = 128 if gemm_tile <= 128 else 512
    A = np.eye(M, K, dtype=np.float32)
    Wt = bf(np.arange(N * K, dtype=np.float32).reshape(N, K) % 251 - 100.0)
    C = np.zeros((M, N), np.float32)
    assert ctx.lib.wmdbg_gemm(ctx.handle, P(A), P(Wt), None, P(C), M, N, K, 6) == 0
    assert np.array_equal(C, Wt.T)


@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_epilogues(ctx, gemm_tile, epi):
    rng = np.random.default_rng(epi)
    M, N, K = 200, 256, 128
    A = bf(rng.standard_normal((M, K)))
    Wt = bf(rng.standard_normal((N, K)) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    C = C0.copy()
    assert ctx.lib.wmdbg_gemm(ctx.handle, P(A), P(Wt), P(bias), P(C), M, N, K, epi) == 0
    lin = A.astype(np.float64) @ Wt.astype(np.float64).T + bias
    if epi == 0:
        ref, tol = lin, 8e-3          # bf16 output: 2^-8 relative
    elif epi == 1:
        ref, tol = torch.nn.functional.gelu(torch.from_numpy(lin)).numpy(), 8e-3
    else:
        ref, tol = C0 + lin, 1e-5
    assert np.abs(C - ref).max() <= tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("rows,d", [(5, 128), (1500, 384), (33, 768), (64, 1280)])
def test_layernorm(ctx, rows, d):
    rng = np.random.default_rng(d)
    x = (rng.standard_normal((rows, d)) * 3 + 1.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    o32 = np.zeros_like(x)
    o16 = np.zeros_like(x)
    assert ctx.lib.wmdbg_layernorm(ctx.handle, P(x), P(g), P(b), rows, d, P(o32), P(o16)) == 0
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x).double(), (d,), torch.from_numpy(g).double(),
                                         torch.from_numpy(b).double(), 1e-5).numpy()
    assert np.abs(o32 - ref).max() <= 2e-5
    assert np.abs(o16 - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("B,H,S", [(1, 2, 1500), (2, 1, 200), (1, 3, 64), (1, 1, 129)])
def test_encoder_attention(ctx, B, H, S):
    rng = np.random.default_rng(S)
    d = H * 64
    q = bf(rng.standard_normal((B, S, d)))
    k = bf(rng.standard_normal((B, S, d)))
    v = bf(rng.standard_normal((B, S, d)) + np.linspace(-1, 1, d)[None, None, :])
    # force an online-softmax rescale late in the sequence: one key that dominates one query
    k[0, S - 3, :64] = q[0, 5, :64] * 3.0
    out = np.zeros((B, S, d), np.float32)
    assert ctx.lib.wmdbg_enc_attention(ctx.handle, P(q), P(k), P(v), B, H, S, P(out)) == 0
    tq, tk, tv = (torch.from_numpy(a).double().view(B, S, H, 64).permute(0, 2, 1, 3) for a in (q, k, v))
    w = torch.softmax(tq @ tk.transpose(-1, -2) / 8.0, dim=-1)
    ref = (w @ tv).permute(0, 2, 1, 3).reshape(B, S, d).numpy()
    assert np.abs(out - ref).max() <= 2e-2 * np.abs(ref).max(), np.abs(out - ref).max()
    assert rel(out, ref) <= 6e-3


@pytest.mark.parametrize("B,N,K,ln", [(8, 1280, 1280, True), (8, 1280, 5120, False), (1, 384, 384, True),
                                      (16, 100, 512, True), (3, 1030, 64, False), (8, 3840, 1280, True),
                                      (2, 128, 1536, False)])
def test_decode_gemv(ctx, B, N, K, ln):
    rng = np.random.default_rng(N + K)
    x = (rng.standard_normal((B, K)) * 2 + 0.3).astype(np.float32)
    Wt = bf(rng.standard_normal((N, K)) * 0.05 + np.linspace(-0.02, 0.02, N)[:, None])
    bias = rng.standard_normal(N).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    b = (0.1 * rng.standard_normal(K)).astype(np.float32)
    out = np.zeros((B, N), np.float32)
    st = ctx.lib.wmdbg_dec_gemv(ctx.handle, P(x), P(g) if ln else None, P(b) if ln else None, P(Wt), P(bias),
                                P(out), B, N, K)
    assert st == 0, ctx.lib.wm_last_error()
    if ln:
        a = torch.nn.functional.layer_norm(torch.from_numpy(x).double(), (K,), torch.from_numpy(g).double(),
                                           torch.from_numpy(b).double(), 1e-5).numpy()
        a = bf(a).astype(np.float64)   # the kernel feeds the matrix pipe bf16 activations
    else:
        a = bf(x).astype(np.float64)
    ref = a @ Wt.astype(np.float64).T + bias
    # LN path: our bf16 rounding of LN(x) can differ from the kernel's by one ulp on a few elements
    tol = (4e-3 if ln else 2e-4) * np.abs(ref).max()
    assert np.abs(out - ref).max() <= tol, (np.abs(out - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("B,N,K", [(8, 768, 3072), (17, 768, 3072), (32, 768, 3072), (17, 1024, 4096), (32, 1024, 4096),
                                   (56, 1280, 5120), (17, 768, 768), (40, 1024, 1024), (33, 512, 2048), (128, 384, 1536),
                                   # two batch blocks per workgroup in the two-parts-per-wave kernel (and the sizes either side)
                                   (72, 1280, 5120), (90, 1280, 5120), (128, 1280, 5120), (96, 768, 3072), (128, 1024, 4096)])
def test_decode_gemv_residual_at_every_width_and_group_size(ctx, B, N, K):
    """the fc2 product (K = 4d) splits K over 16 waves at d = 768 / 1024 / 1280; above one batch block
    the two-parts-per-wave kernel serves it.  At d = 768 / 1024 the launch-shape choice used to contradict it
    (WM_ERR_INVALID on every decode step of whisper-small / -medium with more than 16 rows).  Residual update, its bf16
    copy and the LayerNorm partials, at every model width, one block and several."""
    rng = np.random.default_rng(B + N + K)
    x = bf(rng.standard_normal((B, K)) * 0.5 + np.linspace(-0.2, 0.2, K)[None, :])
    Wt = bf(rng.standard_normal((N, K)) * 0.05 + np.linspace(-0.02, 0.02, N)[:, None])
    bias = rng.standard_normal(N).astype(np.float32)
    r0 = rng.standard_normal((B, N)).astype(np.float32)
    res = r0.copy()
    cp = np.zeros((B, N), np.float32)
    stt = np.zeros((B, 2), np.float32)
    st = ctx.lib.wmdbg_dec_gemv_resid(ctx.handle, P(x), P(Wt), P(bias), P(res), P(cp), P(stt), B, N, K)
    assert st == 0, ctx.lib.wm_last_error()
    ref = r0.astype(np.float64) + x.astype(np.float64) @ Wt.astype(np.float64).T + bias
    scale = np.abs(ref).max()
    assert np.abs(res - ref).max() <= 2e-4 * scale, (np.abs(res - ref).max(), scale)
    assert np.abs(cp - res).max() <= 2 ** -8 * scale                        # the bf16 copy of the same rows
    assert np.allclose(stt[:, 0], res.sum(axis=1), rtol=1e-4, atol=1e-3 * scale)
    assert np.allclose(stt[:, 1], (res.astype(np.float64) ** 2).sum(axis=1), rtol=1e-4)
    # bit-level batch invariance of the residual product: row 0 alone == row 0 inside the group
    one = r0[:1].copy()
    assert ctx.lib.wmdbg_dec_gemv_resid(ctx.handle, P(x[:1].copy()), P(Wt), P(bias), P(one), P(cp[:1].copy()),
                                        P(stt[:1].copy()), 1, N, K) == 0
    assert np.array_equal(one[0], res[0])
    last = r0[B - 1:B].copy()   # ... and the last row (the second block of a two-block workgroup, a partly filled block)
    assert ctx.lib.wmdbg_dec_gemv_resid(ctx.handle, P(x[B - 1:B].copy()), P(Wt), P(bias), P(last), P(cp[:1].copy()),
                                        P(stt[:1].copy()), 1, N, K) == 0
    assert np.array_equal(last[0], res[B - 1])


@pytest.mark.parametrize("B,H,T,n_keys,nsplit", [(2, 2, 448, 1, 1), (2, 2, 448, 37, 1), (1, 3, 448, 448, 1),
                                                 (2, 2, 1500, 1500, 4), (1, 1, 1500, 1500, 2), (8, 20, 1500, 1500, 1),
                                                 (3, 2, 448, 130, 8),
                                                 # nsplit 0 = the decoder's self-attention kernel (4-wave workgroup per pair)
                                                 (2, 2, 448, 1, 0), (2, 3, 448, 37, 0), (1, 3, 448, 128, 0),
                                                 (3, 2, 448, 129, 0), (2, 2, 448, 448, 0), (32, 20, 448, 227, 0),
                                                 # nsplit -1 = the cross-attention path (8-wave block-streaming kernel, <= 256 workgroups)
                                                 (8, 20, 1500, 1500, -1), (3, 2, 1500, 1500, -1), (40, 20, 1500, 1500, -1),
                                                 (2, 2, 1500, 300, -1)])
def test_decode_attention(ctx, B, H, T, n_keys, nsplit):
    rng = np.random.default_rng(T + n_keys)
    q = rng.standard_normal((B, H * 64)).astype(np.float32)
    k = bf(rng.standard_normal((B, H, T, 64)))
    v = bf(rng.standard_normal((B, H, T, 64)) + np.linspace(-1, 1, 64))
    k[:, :, n_keys:] = 1e3    # poison positions the kernel must not read into the softmax
    out = np.zeros((B, H * 64), np.float32)
    st = ctx.lib.wmdbg_dec_attention(ctx.handle, P(q), P(k), P(v), B, H, T, n_keys, nsplit, P(out))
    assert st == 0, ctx.lib.wm_last_error()
    tq = torch.from_numpy(q).double().view(B, H, 1, 64)
    tk = torch.from_numpy(k[:, :, :n_keys]).double()
    tv = torch.from_numpy(v[:, :, :n_keys]).double()
    w = torch.softmax(tq @ tk.transpose(-1, -2) / 8.0, dim=-1)
    ref = (w @ tv).reshape(B, H * 64).numpy()
    # head outputs are written as bf16 (they are the out-projection's MFMA operand)
    assert np.abs(out - ref).max() <= 2 ** -8 * max(1.0, np.abs(ref).max())


def test_decode_attention_is_bitwise_independent_of_the_split(ctx):
    """The 8 streams of a (sequence, head) pair may be dealt to 1, 2, 4 or 8 workgroups (few pairs: fill the chip): the
    head outputs must not change by a single bit, or the tokens would depend on the size of the decode group."""
    rng = np.random.default_rng(5)
    B, H, T = 3, 2, 1500
    q = rng.standard_normal((B, H * 64)).astype(np.float32)
    k = bf(rng.standard_normal((B, H, T, 64)))
    v = bf(rng.standard_normal((B, H, T, 64)))
    outs = []
    for ns in (1, 2, 4, 8, -1):
        out = np.zeros((B, H * 64), np.float32)
        assert ctx.lib.wmdbg_dec_attention(ctx.handle, P(q), P(k), P(v), B, H, T, 1500, ns, P(out)) == 0
        outs.append(out)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_decode_attention_rejects_bad_split(ctx):
    q = np.zeros((1, 64), np.float32)
    k = np.zeros((1, 1, 64, 64), np.float32)
    out = np.zeros((1, 64), np.float32)
    assert ctx.lib.wmdbg_dec_attention(ctx.handle, P(q), P(k), P(k), 1, 1, 64, 64, 9, P(out)) == 1
    assert b"nsplit" in ctx.lib.wm_last_error()


def test_gemm_256_tile_is_bitwise_equal_to_the_128_tile(ctx):
    """Race screen for the staggered-phase 256 x 256 kernel (counted vmcnt, asm ds_reads): it accumulates in the same
    order as the 128 x 128 kernel (and the 64 x 64 one of round 6), so any difference on the same operands is a
    synchronisation bug, not rounding.
    (tools/gpu_gemm_race_screen.py is the long version: 1080 comparisons next to a running bench, 0 mismatches.)"""
    ctx.lib.wmdbg_set_gemm_tile.argtypes = [ctypes.c_int]
    ctx.lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    try:
        for it in range(3):
            for (M, N, K) in [(2048, 2048, 1280), (1031, 768, 448), (1536, 1280, 5120)]:
                rng = np.random.default_rng(17 * it + M + K)
                A = bf(rng.standard_normal((M, K)))
                Wt = bf(rng.standard_normal((N, K)) * 0.05)
                bias = rng.standard_normal(N).astype(np.float32)
                for epi in (6, 1, 2):
                    outs = []
                    # 128: the double-buffer kernel (pipe 1) and the 3-stage pipeline of round 6 (pipe 2); 256; 64
                    for tile, pipe in ((128, 1), (256, 0), (64, 0), (128, 2)):
                        assert ctx.lib.wmdbg_set_gemm_tile(tile) == 0
                        assert ctx.lib.wmdbg_set_tuning(b"gemm128_pipe", pipe) == 0
                        C = np.full((M, N), 0.25, np.float32)
                        assert ctx.lib.wmdbg_gemm(ctx.handle, P(A), P(Wt), P(bias), P(C), M, N, K, epi) == 0
                        outs.append(C)
                    for o in outs[1:]:
                        assert np.array_equal(outs[0], o), (it, M, N, K, epi)
    finally:
        ctx.lib.wmdbg_set_gemm_tile(0)
