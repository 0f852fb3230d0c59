// gemm.hip -- bf16 MFMA GEMM for gfx950:  C[M][N] (epilogue) = A[M][K] . W[N][K]^T
//
// Every encoder-side matrix product goes through this kernel (conv1/conv2 as implicit
// GEMMs via row-address mapping, Q/K/V, attention out-proj, MLP fc1/fc2, cross-attention
// K/V projection): SURVEY.md section 2 rows K4, K5, K7, K9.
//
// Structure (CDNA4-first, not a warp-tiling port):
//   * 128 x 128 x 64 block tile, 256 threads = 4 wave64 as 2(M) x 2(N); each wave owns a
//     64 x 64 output sub-tile = 4 x 4 fragments of v_mfma_f32_16x16x32_bf16 (f32 accumulate).
//   * both operands are K-contiguous in HBM, so a tile row is one 128-byte line; tiles are
//     DMA'd HBM -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR round trip),
//     double-buffered (2 x 32 KiB LDS -> 2 workgroups per CU).
//   * LDS image is lane-linear (a requirement of global_load_lds); the bank-conflict fix
//     is an XOR swizzle of the 16-byte chunk index with (row & 7), applied to the per-lane
//     SOURCE address and again on the ds_read_b128 side (same involution both sides).
//   * workgroup id -> tile mapping is XCD-aware: the 8 XCDs (private L2s) each get a
//     contiguous range of tiles, so neighbouring tiles that share an A panel hit one L2.
//   * A and C rows are addressed as (m / rpb) * bstride + (m % rpb) * rstride, which turns
//     Whisper's two Conv1d layers into GEMMs without materialising im2col: with the
//     activations stored time-major and one zero row of padding, the 3-tap receptive field
//     of an output frame is a CONTIGUOUS run of 3*C elements.
#include <stdlib.h>

#include "model.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (inputs are finite)
    return (bf16_t)(u >> 16);
}

struct GemmDev {
    const bf16_t *A;
    long a_rpb, a_bstride, a_rstride;
    const bf16_t *W;
    const float *bias;
    void *C;
    long c_rpb, c_bstride, c_rstride;
    int M, N, K;
    const float *pos;
    bf16_t *vt;
    int d_model, n_head, seq, seq_pad, batch;
    int tiles_m, tiles_n;
    int group_m;
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmDev p) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][TILE_BYTES];  // [buf][A|B]

    // ---- XCD-aware, bijective workgroup -> tile map -------------------------------------
    const int nwg = p.tiles_m * p.tiles_n;
    int wg = blockIdx.x;
    {
        const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    // grouped tile order inside each XCD's contiguous range: the ~64 tiles an XCD runs concurrently
    // form a GM x 8 block (GM A panels + 8 W panels live in its 4 MiB L2) instead of 1 x 64
    // (1 A panel + 64 W panels).  Worth 3-7 % on the encoder GEMMs (measured; not the main limiter).
    const int GM = p.group_m;
    const int per_group = GM * p.tiles_n;
    const int grp = wg / per_group, in_grp = wg % per_group;
    const int first_m = grp * GM;
    const int gsz = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
    const int tm = first_m + in_grp % gsz, tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- staging addresses: 4 passes x (32 rows x 8 chunks of 16 B) per operand ---------
    // thread -> (row = pass*32 + tid/8, physical chunk = tid%8); it fetches logical chunk
    // (pchunk ^ (row & 7)) so that the lane-linear LDS image is the swizzled one.
    const int srow = tid >> 3, pch = tid & 7;
    const bf16_t *a_src[4];
    const bf16_t *w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + srow;
        const int lch = pch ^ (row & 7);
        long m = m0 + row;
        if (m > p.M - 1) m = p.M - 1;  // clamp: tail rows are masked in the epilogue
        a_src[i] = p.A + (m / p.a_rpb) * p.a_bstride + (m % p.a_rpb) * p.a_rstride + lch * 8;
        long n = n0 + row;
        if (n > p.N - 1) n = p.N - 1;
        w_src[i] = p.W + n * (long)p.K + lch * 8;
    }
    auto stage = [&](int buf, int kt) {
        const long ko = (long)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // LDS destination: wave-uniform base; the hardware adds lane * 16
            char *da = &lds[buf][0][(i * 32 + wave * 8) * 128];
            char *db = &lds[buf][1][(i * 32 + wave * 8) * 128];
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(a_src[i] + ko),
                (__attribute__((address_space(3))) void *)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(w_src[i] + ko),
                (__attribute__((address_space(3))) void *)db, 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier

    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const char *sa = lds[buf][0];
        const char *sb = lds[buf][1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
            const int lch = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wr * 64 + i * 16 + frow;
                af[i] = *(const bf16x8 *)(sa + ra * 128 + ((lch ^ (ra & 7)) << 4));
                const int rb = wc * 64 + i * 16 + frow;
                bfr[i] = *(const bf16x8 *)(sb + rb * 128 + ((lch ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: D fragment (i,j): col n = lane & 15, rows m = (lane >> 4) * 4 + r ------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mb = m0 + wr * 64 + i * 16 + fq * 4;  // first of 4 consecutive rows
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + frow;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
            if (EPI == EPI_QKV_ENC && n >= 2 * p.d_model) {
                // value projection -> V^T [b][h][e][seq_pad]: 4 consecutive frames, one 8-B store
                if (mb < p.M) {
                    const int b = mb / p.seq, s = mb % p.seq;
                    const int hn = n - 2 * p.d_model, h = hn >> 6, e = hn & 63;
                    unsigned lo = (unsigned)f2bf(acc[i][j][0] + bv) | ((unsigned)f2bf(acc[i][j][1] + bv) << 16);
                    unsigned hi = (unsigned)f2bf(acc[i][j][2] + bv) | ((unsigned)f2bf(acc[i][j][3] + bv) << 16);
                    bf16_t *dst = p.vt + (((long)(b * p.n_head + h) * 64 + e) * p.seq_pad + s);
                    *(uint2 *)dst = make_uint2(lo, hi);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = mb + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == EPI_XKV) {
                    // n -> (kv, h, e); out[kv][b][h][s][e]
                    const int kv = n / p.d_model, hn = n % p.d_model, h = hn >> 6, e = hn & 63;
                    const int b = (int)(m / p.seq), s = (int)(m % p.seq);
                    bf16_t *dst = (bf16_t *)p.C +
                                  ((((long)kv * p.batch + b) * p.n_head + h) * p.seq + s) * 64 + e;
                    *dst = f2bf(v);
                    continue;
                }
                const long co = (m / p.c_rpb) * p.c_bstride + (m % p.c_rpb) * p.c_rstride + n;
                if (EPI == EPI_BIAS_BF16 || EPI == EPI_QKV_ENC) {
                    ((bf16_t *)p.C)[co] = f2bf(v);
                } else if (EPI == EPI_GELU_BF16) {
                    ((bf16_t *)p.C)[co] = f2bf(gelu_erf(v));
                } else if (EPI == EPI_RESID_F32) {
                    ((float *)p.C)[co] += v;
                } else if (EPI == EPI_CONV2_F32) {
                    ((float *)p.C)[co] = gelu_erf(v) + p.pos[(m % p.c_rpb) * (long)p.N + n];
                } else {  // EPI_F32
                    ((float *)p.C)[co] = v;
                }
            }
        }
    }
}

}  // namespace

int wm_gemm(wm_ctx *ctx, const GemmArgs &g) {
    WM_REQUIRE(g.K % BK == 0 && g.K >= BK, WM_ERR_INVALID, "gemm: K=%d must be a multiple of %d", g.K, BK);
    WM_REQUIRE(g.M > 0 && g.N > 0, WM_ERR_INVALID, "gemm: empty problem");
    GemmDev p;
    p.A = g.A; p.a_rpb = g.a_rpb; p.a_bstride = g.a_bstride; p.a_rstride = g.a_rstride;
    p.W = g.W; p.bias = g.bias; p.C = g.C;
    p.c_rpb = g.c_rpb; p.c_bstride = g.c_bstride; p.c_rstride = g.c_rstride;
    p.M = g.M; p.N = g.N; p.K = g.K; p.pos = g.pos; p.vt = g.vt;
    p.d_model = g.d_model; p.n_head = g.n_head; p.seq = g.seq; p.seq_pad = g.seq_pad; p.batch = g.batch;
    p.tiles_m = (g.M + BM - 1) / BM;
    p.tiles_n = (g.N + BN - 1) / BN;
    const int grid = p.tiles_m * p.tiles_n;
    static const int env_gm = getenv("WM_GEMM_GM") ? atoi(getenv("WM_GEMM_GM")) : 0;
    p.group_m = env_gm > 0 ? env_gm : 4;  // measured at large-v2, B = 8: GM 1 / 4 / 8 / 16 -> fc1 341 / 328 / 335 / 335 us
    static const char *names[] = {"gemm_bias_bf16", "gemm_gelu_bf16", "gemm_resid_f32", "gemm_conv2_f32",
                                  "gemm_qkv_enc", "gemm_xkv", "gemm_f32"};
    WmProfScope ps(&ctx->prof, names[g.epi], ctx->stream);
    switch (g.epi) {
        case EPI_BIAS_BF16: gemm_bf16_kernel<EPI_BIAS_BF16><<<grid, 256, 0, ctx->stream>>>(p); break;
        case EPI_GELU_BF16: gemm_bf16_kernel<EPI_GELU_BF16><<<grid, 256, 0, ctx->stream>>>(p); break;
        case EPI_RESID_F32: gemm_bf16_kernel<EPI_RESID_F32><<<grid, 256, 0, ctx->stream>>>(p); break;
        case EPI_CONV2_F32: gemm_bf16_kernel<EPI_CONV2_F32><<<grid, 256, 0, ctx->stream>>>(p); break;
        case EPI_QKV_ENC: gemm_bf16_kernel<EPI_QKV_ENC><<<grid, 256, 0, ctx->stream>>>(p); break;
        case EPI_XKV: gemm_bf16_kernel<EPI_XKV><<<grid, 256, 0, ctx->stream>>>(p); break;
        case EPI_F32: gemm_bf16_kernel<EPI_F32><<<grid, 256, 0, ctx->stream>>>(p); break;
        default: wm_set_error("gemm: bad epilogue %d", g.epi); return WM_ERR_INVALID;
    }
    WM_HIP(hipGetLastError());
    return WM_OK;
}
