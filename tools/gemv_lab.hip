// gemv_lab.hip -- standalone gfx950 micro-benchmarks behind the decode-kernel redesign of round 2
// (not part of the product library; build: hipcc --offload-arch=gfx950 -O3 tools/gemv_lab.hip -o gpurun_out/gemv_lab).
//
//  1. pull:  how fast can n_wg workgroups pull disjoint contiguous slices (HBM-cold vs cache-warm, nt vs plain)?
//            -> is there a per-CU cap, i.e. must every decode GEMV be spread over all 256 CUs?
//  2. dot2:  v_dot2c_f32_bf16 issue rate per SIMD.
//  3. gemv_rw: the row-per-wave VALU GEMV candidate (x in registers, weights row-major, no split-K across
//            workgroups) at the large-v2 decoder shapes, graph-replayed chains over rotating weight matrices.
// Every timing is the mean per launch of a captured chain of dependent (same-stream) launches, replayed once warm.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

static float time_chain(hipStream_t s, int iters, const std::function<void(int)> &launch) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best * 1e3f / iters;
}

// ------------------------------------------------------------------ 1. pull ------------------
template <bool NT>
__global__ __launch_bounds__(256) void pull_kernel(const char *base, long bytes_per_wg, float *sink) {
    const u32x4 *src = (const u32x4 *)(base + (long)blockIdx.x * bytes_per_wg);
    const long n16 = bytes_per_wg >> 4;
    unsigned acc = 0;
    for (long i = threadIdx.x; i < n16; i += 256 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long j = i + (long)u * 256;
            const u32x4 *p = src + (j < n16 ? j : i);
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
    }
    if (acc == 0x12345678u) sink[0] = 1.f;  // never true in practice: keeps the loads alive
}

__global__ void trivial_kernel(float *p) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 123.f) p[0] += 1.f;
}

// ------------------------------------------------------------------ 2. dot2 ------------------
__global__ __launch_bounds__(256) void dot2_kernel(float *out, int iters) {
    unsigned a = threadIdx.x * 0x3f803f80u, b = 0x3f803f80u;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc[u] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc[u], false);
        a += 0x00010001u;
    }
    float s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u];
    if (s == 1.2345f) out[0] = s;
}

// ------------------------------------------------------------------ 3. gemv_rw ---------------
// Row-per-wave GEMV for decode groups: out[b][n] = sum_k A[b][k] W[n][k], K split in KW slices of KS = DW * 128
// elements (one slice per wave along K; KW > 1 reduces through LDS in slice order).  A lane owns DW dwords (2 bf16
// each) of a slice: 16-byte chunk j at elements j*512 + lane*8, the remaining DW % 4 dwords at (DW/4)*512 + lane*2*(DW%4).
// The activations of 8 batch rows live in registers (8 * DW dwords), weights of RPW rows too; larger batches loop over
// blocks of 8 rows with the weights held.  Products on v_dot2c_f32_bf16; cross-lane sums by a reduce-scatter whose
// tree is the same for every batch row and batch size (bit-level batch invariance).
template <int DW>
__device__ __forceinline__ void load_frag(unsigned (&v)[DW], const bf16_t *p, int lane) {
    constexpr int N4 = DW / 4, REM = DW % 4;
#pragma unroll
    for (int j = 0; j < N4; ++j) {
        const u32x4 t = *(const u32x4 *)(p + j * 512 + lane * 8);
        v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3];
    }
    if (REM == 1) v[4 * N4] = *(const unsigned *)(p + N4 * 512 + lane * 2);
    if (REM == 2) {
        const u32x2 t = *(const u32x2 *)(p + N4 * 512 + lane * 4);
        v[4 * N4] = t[0]; v[4 * N4 + 1] = t[1];
    }
    if (REM == 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) v[4 * N4 + i] = *(const unsigned *)(p + N4 * 512 + lane * 6 + 2 * i);
    }
}
template <int DW>
__device__ __forceinline__ void load_frag_nt(unsigned (&v)[DW], const bf16_t *p, int lane) {
    constexpr int N4 = DW / 4, REM = DW % 4;
#pragma unroll
    for (int j = 0; j < N4; ++j) {
        const u32x4 t = __builtin_nontemporal_load((const u32x4 *)(p + j * 512 + lane * 8));
        v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3];
    }
    if (REM == 1) v[4 * N4] = __builtin_nontemporal_load((const unsigned *)(p + N4 * 512 + lane * 2));
    if (REM == 2) {
        const u32x2 t = __builtin_nontemporal_load((const u32x2 *)(p + N4 * 512 + lane * 4));
        v[4 * N4] = t[0]; v[4 * N4 + 1] = t[1];
    }
    if (REM == 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) v[4 * N4 + i] = __builtin_nontemporal_load((const unsigned *)(p + N4 * 512 + lane * 6 + 2 * i));
    }
}

__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}

// 8 per-lane partials (one per batch row) -> every lane ends with the 64-lane total of row (lane >> 3)
__device__ __forceinline__ float reduce_scatter8(const float (&v)[8], int lane) {
    float a[4], b2[2], c;
    {
        const bool hi = lane & 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = hi ? v[i + 4] : v[i], send = hi ? v[i] : v[i + 4];
            a[i] = keep + __shfl_xor(send, 32);
        }
    }
    {
        const bool hi = lane & 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = hi ? a[i + 2] : a[i], send = hi ? a[i] : a[i + 2];
            b2[i] = keep + __shfl_xor(send, 16);
        }
    }
    {
        const bool hi = lane & 8;
        const float keep = hi ? b2[1] : b2[0], send = hi ? b2[0] : b2[1];
        c = keep + __shfl_xor(send, 8);
    }
    c += __shfl_xor(c, 4);
    c += __shfl_xor(c, 2);
    c += __shfl_xor(c, 1);
    return c;
}

struct RwArgs {
    const bf16_t *W;   // [N][K] row-major
    const bf16_t *A;   // [B][K] bf16 activations
    const float *c1, *c2;
    float *out;        // [B][N]
    float *x;          // MODE 2: residual [B][N] (+=)
    bf16_t *xb;        // MODE 2: bf16 copy
    int B, N, K;
};

// MODE 0: out = A W^T.   MODE 1: LayerNorm folded: out = rstd*(A Wg^T - mean*c1) + c2, statistics from the registers.
// MODE 2: KW slices, residual epilogue (x += acc + c2; xb = bf16(x)).
template <int DW, int RPW, int KW, int MODE>
__global__ __launch_bounds__(512) void gemv_rw_kernel(RwArgs p) {
    __shared__ float red[4][8][8];  // [slice][row][b]  (KW > 1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ks = KW > 1 ? wave % KW : 0;
    const int rw = KW > 1 ? wave / KW : wave;
    const int rwaves = (blockDim.x >> 6) / KW;
    const int row0 = (blockIdx.x * rwaves + rw) * RPW;
    constexpr int KS = DW * 128;
    unsigned w[RPW][DW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int n = row0 + r < p.N ? row0 + r : p.N - 1;
        load_frag_nt<DW>(w[r], p.W + (long)n * p.K + ks * KS, lane);
    }
    unsigned xa[8][DW];
    const int bl = lane >> 3;
    for (int b0 = 0; b0 < p.B; b0 += 8) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int bc = b0 + b < p.B ? b0 + b : p.B - 1;
            load_frag<DW>(xa[b], p.A + (long)bc * p.K + ks * KS, lane);
        }
        float tot[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float acc[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < DW; ++i) a = dot2(xa[b][i], w[r][i], a);
                acc[b] = a;
            }
            tot[r] = reduce_scatter8(acc, lane);
        }
        float mean = 0.f, rstd = 1.f;
        if (MODE == 1) {
            float s1[8], s2[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int i = 0; i < DW; ++i) {
                    a1 = dot2(xa[b][i], 0x3f803f80u, a1);
                    a2 = dot2(xa[b][i], xa[b][i], a2);
                }
                s1[b] = a1; s2[b] = a2;
            }
            const float S1 = reduce_scatter8(s1, lane), S2 = reduce_scatter8(s2, lane);
            mean = S1 / (float)p.K;
            float var = S2 / (float)p.K - mean * mean;
            var = var > 0.f ? var : 0.f;
            rstd = rsqrtf(var + 1e-5f);
        }
        const int b = b0 + bl;
        if (KW == 1) {
            if ((lane & 7) == 0 && b < p.B) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int n = row0 + r;
                    if (n < p.N) {
                        float v = tot[r];
                        if (MODE == 1) v = rstd * (v - mean * p.c1[n]) + p.c2[n];
                        p.out[(long)b * p.N + n] = v;
                    }
                }
            }
        } else {
            if ((lane & 7) == 0) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) red[ks][rw * RPW + r][bl] = tot[r];
            }
            __syncthreads();
            const int t = threadIdx.x;
            if (t < rwaves * RPW * 8) {
                const int r = t >> 3, bb = t & 7;
                const int n = (blockIdx.x * rwaves) * RPW + r;
                float v = red[0][r][bb];
#pragma unroll
                for (int s = 1; s < KW; ++s) v += red[s][r][bb];
                if (n < p.N && b0 + bb < p.B) {
                    const long o = (long)(b0 + bb) * p.N + n;
                    const float xn = p.x[o] + v + p.c2[n];
                    p.x[o] = xn;
                    const __bf16 h = (__bf16)xn;
                    p.xb[o] = __builtin_bit_cast(bf16_t, h);
                }
            }
            __syncthreads();
        }
    }
}

static float bf2f(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static bf16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <int DW, int RPW, int KW, int MODE>
static void run_rw(hipStream_t s, const char *name, int B, int N, int K, int rwaves, int n_mats, bool check) {
    const size_t wn = (size_t)N * K;
    bf16_t *dW, *dA, *dxb;
    float *dout, *dc1, *dc2, *dx;
    CK(hipMalloc(&dW, wn * 2 * n_mats + 4096));
    CK(hipMalloc(&dA, (size_t)128 * K * 2 + 4096));
    CK(hipMalloc(&dout, (size_t)128 * N * 4));
    CK(hipMalloc(&dx, (size_t)128 * N * 4));
    CK(hipMalloc(&dxb, (size_t)128 * N * 2));
    CK(hipMalloc(&dc1, N * 4));
    CK(hipMalloc(&dc2, N * 4));
    std::vector<bf16_t> hW(wn), hA((size_t)128 * K);
    std::vector<float> hc1(N), hc2(N);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) / 16777216.0f) - 0.5f; };
    for (auto &v : hW) v = f2bf(0.05f * rnd());
    for (auto &v : hA) v = f2bf(2.0f * rnd() + 0.3f);
    for (int n = 0; n < N; ++n) {
        float c = 0;
        for (int k = 0; k < K; ++k) c += bf2f(hW[(size_t)n * K + k]);
        hc1[n] = c;
        hc2[n] = 0.01f * rnd();
    }
    for (int m = 0; m < n_mats; ++m) CK(hipMemcpy(dW + wn * m, hW.data(), wn * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc1, hc1.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc2, hc2.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dx, 0, (size_t)128 * N * 4));
    RwArgs a;
    a.A = dA; a.c1 = dc1; a.c2 = dc2; a.out = dout; a.x = dx; a.xb = dxb; a.B = B; a.N = N; a.K = K;
    const int rows_per_wg = rwaves * RPW;
    const int grid = (N + rows_per_wg - 1) / rows_per_wg;
    const int threads = rwaves * KW * 64;
    if (check) {
        a.W = dW;
        CK(hipMemsetAsync(dx, 0, (size_t)128 * N * 4, s));
        gemv_rw_kernel<DW, RPW, KW, MODE><<<grid, threads, 0, s>>>(a);
        CK(hipStreamSynchronize(s));
        std::vector<float> got((size_t)B * N);
        CK(hipMemcpy(got.data(), MODE == 2 ? dx : dout, got.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int b = 0; b < B; ++b) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < K; ++k) { const double v = bf2f(hA[(size_t)b * K + k]); s1 += v; s2 += v * v; }
            const double mean = s1 / K, rstd = 1.0 / sqrt(s2 / K - mean * mean + 1e-5);
            for (int n = 0; n < N; n += 7) {
                double acc = 0;
                for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)b * K + k]) * bf2f(hW[(size_t)n * K + k]);
                double ref = acc;
                if (MODE == 1) ref = rstd * (acc - mean * hc1[n]) + hc2[n];
                if (MODE == 2) ref = acc + hc2[n];
                const double e = fabs(ref - got[(size_t)b * N + n]);
                if (e > maxerr) maxerr = e;
                if (fabs(ref) > maxref) maxref = fabs(ref);
            }
        }
        printf("  check %-8s B=%3d: max |err| %.3g (max |ref| %.3g)\n", name, B, maxerr, maxref);
    }
    const float us = time_chain(s, 256, [&](int i) {
        RwArgs b = a;
        b.W = dW + wn * (i % n_mats);
        gemv_rw_kernel<DW, RPW, KW, MODE><<<grid, threads, 0, s>>>(b);
    });
    printf("gemv_rw %-8s B=%3d N=%5d K=%4d grid=%4d x %3d thr  mats=%2d : %6.2f us  (%6.0f GB/s weights)\n", name, B, N, K, grid,
           threads, n_mats, us, wn * 2 / 1e3 / us);
    CK(hipFree(dW)); CK(hipFree(dA)); CK(hipFree(dout)); CK(hipFree(dx)); CK(hipFree(dxb)); CK(hipFree(dc1)); CK(hipFree(dc2));
}


// ------------------------------------------------------------------ 4. gemv_mf ---------------
// Lean MFMA GEMV for decode groups of any size: one workgroup per 16-row weight tile (fragment-tiled layout: k-step s of
// tile t = the contiguous KiB at ((t * K/32 + s) * 64 + lane) * 8), K split over the NW waves (SPW k-steps each, all weight
// loads issued first), batch rows in blocks of 16 looped INSIDE the workgroup with the weights held in registers and the
// next block's activation fragments requested before the current block's products.  Activations arrive as bf16 (the
// producer writes the bf16 copy of the residual), LayerNorm is folded: out = rstd * (x Wg^T - mean * c1) + c2 with the row
// statistics taken from the very fragments the wave multiplies.  Cross-wave sums through LDS in wave order.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct MfArgs {
    const bf16_t *W;   // tiled
    const bf16_t *A;   // [B][K] bf16
    const float *c1, *c2;
    float *out;        // [B][N] (MODE 0/1)
    float *x;          // MODE 2: residual
    bf16_t *xb;        // MODE 2: bf16 copy of the updated residual
    int B, N, K;
};

template <int SPW, int MODE>
__global__ __launch_bounds__(1024) void gemv_mf_kernel(MfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NW = blockDim.x >> 6;
    float *red = (float *)smem;                   // [2][NW][64][4]
    float *st = red + 2 * NW * 256;               // [2][NW][16][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrow = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x, n0 = tile * 16;
    const bf16_t *wp = p.W + (((long)tile * (p.K >> 5) + (long)wave * SPW) * 64 + lane) * 8;
    u32x4 wf[SPW];
#pragma unroll
    for (int u = 0; u < SPW; ++u) wf[u] = *(const u32x4 *)(wp + u * 512);
    const int kbase = wave * SPW * 32 + kq * 8;
    const int n = n0 + nrow;
    const int nc = n < p.N ? n : p.N - 1;
    float c1v = 0.f, c2v = 0.f;
    if (wave == 0) {
        c2v = p.c2[nc];
        if (MODE == 1) c1v = p.c1[nc];
    }
    u32x4 af[SPW];
    {
        const int rowc = nrow < p.B ? nrow : p.B - 1;
        const bf16_t *ap = p.A + (long)rowc * p.K + kbase;
#pragma unroll
        for (int u = 0; u < SPW; ++u) af[u] = *(const u32x4 *)(ap + u * 32);
    }
    int par = 0;
    for (int b0 = 0; b0 < p.B; b0 += 16, par ^= 1) {
        float xold[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 2 && wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + kq * 4 + r;
                xold[r] = p.x[(long)(b < p.B ? b : p.B - 1) * p.N + nc];
            }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[u]), __builtin_bit_cast(bf16x8, wf[u]), acc, 0, 0, 0);
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 = dot2(af[u][j], 0x3f803f80u, s1);
                    s2 = dot2(af[u][j], af[u][j], s2);
                }
            }
        }
        // next block's fragments: requested now, consumed after the reduction / epilogue of this block
        if (b0 + 16 < p.B) {
            const int rn = b0 + 16 + nrow;
            const bf16_t *ap = p.A + (long)(rn < p.B ? rn : p.B - 1) * p.K + kbase;
#pragma unroll
            for (int u = 0; u < SPW; ++u) af[u] = *(const u32x4 *)(ap + u * 32);
        }
        float *redp = red + par * NW * 256, *stp = st + par * NW * 32;
        if (MODE == 1) {
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (lane < 16) { stp[(wave * 16 + lane) * 2] = s1; stp[(wave * 16 + lane) * 2 + 1] = s2; }
        }
        if (NW > 1 || MODE == 1) {
            *(f32x4 *)(redp + (wave * 64 + lane) * 4) = acc;
            __syncthreads();
        }
        if (wave == 0) {
            for (int w = 1; w < NW; ++w) acc += *(const f32x4 *)(redp + (w * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bl = kq * 4 + r, b = b0 + bl;
                float v = acc[r];
                if (MODE == 1) {
                    float S1 = 0.f, S2 = 0.f;
                    for (int w = 0; w < NW; ++w) { S1 += stp[(w * 16 + bl) * 2]; S2 += stp[(w * 16 + bl) * 2 + 1]; }
                    const float mean = S1 / (float)p.K;
                    float var = S2 / (float)p.K - mean * mean;
                    var = var > 0.f ? var : 0.f;
                    const float rstd = rsqrtf(var + 1e-5f);
                    v = rstd * (v - mean * c1v) + c2v;
                } else {
                    v += c2v;
                }
                if (b < p.B && n < p.N) {
                    if (MODE == 2) {
                        const float xn = xold[r] + v;
                        p.x[(long)b * p.N + n] = xn;
                        const __bf16 h = (__bf16)xn;
                        p.xb[(long)b * p.N + n] = __builtin_bit_cast(bf16_t, h);
                    } else {
                        p.out[(long)b * p.N + n] = v;
                    }
                }
            }
        }
    }
}

static size_t tiled_off(size_t n, size_t k, size_t K) {
    return (((n >> 4) * (K >> 5) + (k >> 5)) * 64 + (n & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7);
}

template <int SPW, int MODE>
static void run_mf(hipStream_t s, const char *name, int B, int N, int K, int n_mats, bool check) {
    const int NW = K / 32 / SPW;
    const size_t wn = (size_t)N * K;
    bf16_t *dW, *dA, *dxb;
    float *dout, *dc1, *dc2, *dx;
    CK(hipMalloc(&dW, wn * 2 * n_mats + 4096));
    CK(hipMalloc(&dA, (size_t)128 * K * 2 + 4096));
    CK(hipMalloc(&dout, (size_t)128 * N * 4));
    CK(hipMalloc(&dx, (size_t)128 * N * 4));
    CK(hipMalloc(&dxb, (size_t)128 * N * 2));
    CK(hipMalloc(&dc1, N * 4));
    CK(hipMalloc(&dc2, N * 4));
    std::vector<bf16_t> hW(wn), hWt(wn), hA((size_t)128 * K);
    std::vector<float> hc1(N), hc2(N);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) / 16777216.0f) - 0.5f; };
    for (auto &v : hW) v = f2bf(0.05f * rnd());
    for (auto &v : hA) v = f2bf(2.0f * rnd() + 0.3f);
    for (int n = 0; n < N; ++n) {
        float c = 0;
        for (int k = 0; k < K; ++k) { c += bf2f(hW[(size_t)n * K + k]); hWt[tiled_off(n, k, K)] = hW[(size_t)n * K + k]; }
        hc1[n] = c;
        hc2[n] = 0.01f * rnd();
    }
    for (int m = 0; m < n_mats; ++m) CK(hipMemcpy(dW + wn * m, hWt.data(), wn * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc1, hc1.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc2, hc2.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dx, 0, (size_t)128 * N * 4));
    MfArgs a;
    a.A = dA; a.c1 = dc1; a.c2 = dc2; a.out = dout; a.x = dx; a.xb = dxb; a.B = B; a.N = N; a.K = K;
    const int grid = N / 16, threads = NW * 64;
    const size_t lds = (size_t)2 * NW * 1024 + 2 * NW * 128;
    if (check) {
        a.W = dW;
        CK(hipMemsetAsync(dx, 0, (size_t)128 * N * 4, s));
        gemv_mf_kernel<SPW, MODE><<<grid, threads, lds, s>>>(a);
        CK(hipStreamSynchronize(s));
        std::vector<float> got((size_t)B * N);
        CK(hipMemcpy(got.data(), MODE == 2 ? dx : dout, got.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int b = 0; b < B; ++b) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < K; ++k) { const double v = bf2f(hA[(size_t)b * K + k]); s1 += v; s2 += v * v; }
            const double mean = s1 / K, rstd = 1.0 / sqrt(s2 / K - mean * mean + 1e-5);
            for (int n = 0; n < N; n += 7) {
                double acc = 0;
                for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)b * K + k]) * bf2f(hW[(size_t)n * K + k]);
                double ref = acc + hc2[n];
                if (MODE == 1) ref = rstd * (acc - mean * hc1[n]) + hc2[n];
                const double e = fabs(ref - got[(size_t)b * N + n]);
                if (e > maxerr) maxerr = e;
                if (fabs(ref) > maxref) maxref = fabs(ref);
            }
        }
        printf("  check %-8s B=%3d: max |err| %.3g (max |ref| %.3g)\n", name, B, maxerr, maxref);
    }
    const float us = time_chain(s, 256, [&](int i) {
        MfArgs b = a;
        b.W = dW + wn * (i % n_mats);
        gemv_mf_kernel<SPW, MODE><<<grid, threads, lds, s>>>(b);
    });
    printf("gemv_mf %-8s B=%3d N=%5d K=%4d grid=%4d x %4d thr  mats=%2d : %6.2f us  (%6.0f GB/s weights)\n", name, B, N, K, grid,
           threads, n_mats, us, wn * 2 / 1e3 / us);
    CK(hipFree(dW)); CK(hipFree(dA)); CK(hipFree(dout)); CK(hipFree(dx)); CK(hipFree(dxb)); CK(hipFree(dc1)); CK(hipFree(dc2));
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float *sink;
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(sink, 0, 256));
    {
        const float us = time_chain(s, 512, [&](int) { trivial_kernel<<<256, 64, 0, s>>>(sink); });
        printf("launch floor (graph, trivial 256 WG): %.2f us\n", us);
    }
    // ---- 1. pull
    if (!getenv("LAB_SKIP_PULL")) {
        const long total_max = 16L << 20;
        const int n_buf = 48;   // 48 x 16 MiB = 768 MiB >> 256 MiB MALL
        char *buf;
        CK(hipMalloc(&buf, total_max * n_buf));
        CK(hipMemset(buf, 1, total_max * n_buf));
        const long totals[] = {3276800, 13107200};
        const int wgs[] = {40, 80, 160, 256, 320, 512, 1024, 2048};
        for (long total : totals)
            for (int nt = 0; nt < 2; ++nt)
                for (int cold = 0; cold < 2; ++cold) {
                    printf("pull %5.2f MB %s %s :", total / 1e6, nt ? "nt   " : "plain", cold ? "cold(HBM)" : "warm     ");
                    for (int n_wg : wgs) {
                        const long per = (total / n_wg + 15) & ~15L;
                        const float us = time_chain(s, 240, [&](int i) {
                            const char *b = buf + (cold ? (long)(i % n_buf) * total_max : 0);
                            if (nt) pull_kernel<true><<<n_wg, 256, 0, s>>>(b, per, sink);
                            else pull_kernel<false><<<n_wg, 256, 0, s>>>(b, per, sink);
                        });
                        printf(" wg%-4d %5.2fus", n_wg, us);
                    }
                    printf("\n");
                }
        CK(hipFree(buf));
    }
    // ---- 2. dot2 rate: 256 WGs x 256 threads (1 wave per SIMD), 8 independent accumulators
    {
        const int iters = 4096;
        const float us = time_chain(s, 8, [&](int) { dot2_kernel<<<256, 256, 0, s>>>(sink, iters); });
        const double per_simd = (double)iters * 8;   // dot2 instructions per wave (one wave per SIMD)
        printf("dot2: %.2f us for %d x 8 per wave -> %.2f ns per wave-instruction (%.2f cycles at 2.4 GHz)\n", us, iters,
               us * 1e3 / per_simd, us * 1e3 / per_simd * 2.4);
        const float us2 = time_chain(s, 8, [&](int) { dot2_kernel<<<512, 256, 0, s>>>(sink, iters); });
        printf("dot2 (2 waves per SIMD): %.2f us -> %.2f ns per wave-instruction per SIMD\n", us2, us2 * 1e3 / (2 * per_simd));
    }
    // ---- 4. gemv_mf at the large-v2 decoder shapes
    for (int B : {8, 16, 56, 96}) {
        run_mf<10, 1>(s, "ln_qkv", B, 3840, 1280, 32, true);
        run_mf<10, 2>(s, "attn_out", B, 1280, 1280, 32, true);
        run_mf<10, 1>(s, "ln_fc1", B, 5120, 1280, 32, true);
        run_mf<10, 2>(s, "fc2", B, 1280, 5120, 32, true);
        run_mf<10, 2>(s, "out_warm", B, 1280, 1280, 1, false);
        run_mf<5, 2>(s, "out_8w", B, 1280, 1280, 32, false);
        run_mf<5, 1>(s, "qkv_8w", B, 3840, 1280, 32, false);
    }
    if (getenv("LAB_SKIP_RW")) return 0;
    // ---- 3. gemv_rw at the large-v2 decoder shapes (d = 1280: DW = 10)
    for (int B : {8, 16, 56, 96}) {
        const bool chk = true;
        run_rw<10, 3, 1, 1>(s, "ln_qkv", B, 3840, 1280, 5, 32, chk);
        run_rw<10, 1, 1, 0>(s, "attn_out", B, 1280, 1280, 5, 32, chk);
        run_rw<10, 4, 1, 1>(s, "ln_fc1", B, 5120, 1280, 5, 32, chk);
        run_rw<10, 5, 4, 2>(s, "fc2", B, 1280, 5120, 1, 32, chk);
        run_rw<10, 1, 1, 0>(s, "out_warm", B, 1280, 1280, 5, 1, false);
    }
    // other model widths (tiny.en 384, base 512, small 768), B = 8
    run_rw<3, 3, 1, 1>(s, "qkv384", 8, 1152, 384, 3, 8, true);
    run_rw<4, 3, 1, 1>(s, "qkv512", 8, 1536, 512, 2, 8, true);
    run_rw<6, 3, 1, 1>(s, "qkv768", 8, 2304, 768, 3, 8, true);
    run_rw<6, 3, 4, 2>(s, "fc2_768", 8, 768, 3072, 1, 8, true);
    run_rw<1, 3, 1, 1>(s, "qkv128", 8, 384, 128, 1, 8, true);
    return 0;
}
