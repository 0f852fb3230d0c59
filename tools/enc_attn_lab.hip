// enc_attn_lab.hip -- standalone ablation lab for the encoder attention kernel (tools/, not product): the kernel text of
// csrc/enc_kernels.hip with pieces switched off by a template mask, timed at large-v2 geometry.  Results of an ablated
// variant are WRONG by construction; only the times mean something.
//   ABL bits: 1 no QK^T MFMAs, 2 no exp, 4 no PV MFMAs, 8 no global->LDS staging, 16 no barrier, 32 never rescale
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/enc_attn_lab.hip -o tools/build/enc_attn_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
constexpr int KS_STRIDE = 144;  // bytes per K row in LDS (128 + 16 pad): conflict-free b128
constexpr int VS_STRIDE = 136;  // bytes per V^T row in LDS (128 + 8 pad): conflict-free b64
constexpr float ATT_RESCALE_THR = 8.0f;   // log2 units: p <= 2^8 before the reference moves

template <bool SUM_MFMA, int ABL, int OCC>
__global__ __launch_bounds__(256, OCC) void enc_attn_kernel(const bf16_t *__restrict__ qk,
                                                          const bf16_t *__restrict__ vt,
                                                          bf16_t *__restrict__ att, int H, int S,
                                                          int S_pad, int d, int n_q, int n_bh) {
    __shared__ __attribute__((aligned(16))) char ks[2][64 * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) char vs[2][64 * VS_STRIDE];
    // XCD-aware workgroup map.  The n_q query blocks of one (chunk, head) pair all stream the same 384 KB of K / V^T;
    // dispatch places workgroup w on XCD w % 8 (observed, not guaranteed: a wrong guess only costs speed), and each XCD
    // has a private L2 -- so the pairs are dealt to the XCDs (pair % 8) and the n_q blocks of a pair are CONSECUTIVE
    // workgroups of that XCD: K / V^T is fetched from HBM once per pair instead of once per XCD that happens to hold one
    // of its query blocks (round 1, grid (n_q, pairs): 537 MB fetched per launch for 93 MB of Q + K + V^T).
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int bh = (j / n_q) * 8 + xcd, qblk = j % n_q;
    if (bh >= n_bh) return;  // workgroup-uniform (pair count padded to a multiple of 8)
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hf = lane >> 5;
    const int q = qblk * 128 + wave * 32 + ql;
    const int qc = q < S ? q : S - 1;
    const long ld = 2L * d;

    // Q^T B-fragments (pre-scaled queries): lane (col q, k = hf*8 + 16*ks .. +8)
    bf16x8 qf[4];
    {
        const bf16_t *qp = qk + ((long)b * S + qc) * ld + h * 64 + hf * 8;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *(const bf16x8 *)(qp + 16 * s4);
    }

    // tile staging: 512 16-byte chunks per operand; thread handles chunks tid and tid+256.  Addresses are a wave-uniform
    // base (SGPRs) + a 32-bit per-thread byte offset that advances by a constant per tile.
    const char *kbase = (const char *)(qk + (long)b * S * ld + d + h * 64);
    const char *vbase = (const char *)(vt + (long)(b * H + h) * 64 * S_pad);
    const int row0 = tid >> 3, c8 = tid & 7, row1 = row0 + 32;
    const unsigned kstep = (unsigned)(64 * ld * 2);            // bytes between consecutive 64-key tiles of K
    unsigned ko0 = (unsigned)(row0 * ld * 2 + c8 * 16), ko1 = (unsigned)(row1 * ld * 2 + c8 * 16);
    unsigned vo0 = (unsigned)(row0 * S_pad * 2 + c8 * 16), vo1 = (unsigned)(row1 * S_pad * 2 + c8 * 16);
    uint4 kr0, kr1, vr0, vr1;
#define ATT_GLOAD_NEXT()                                                    \
    do {                                                                    \
        ko0 += kstep; ko1 += kstep; vo0 += 128u; vo1 += 128u;               \
        kr0 = *(const uint4 *)(kbase + ko0);                                \
        kr1 = *(const uint4 *)(kbase + ko1);                                \
        vr0 = *(const uint4 *)(vbase + vo0);                                \
        vr1 = *(const uint4 *)(vbase + vo1);                                \
    } while (0)
#define ATT_LSTORE(buf)                                                                            \
    do {                                                                                           \
        *(uint4 *)(ks[buf] + row0 * KS_STRIDE + c8 * 16) = kr0;                                    \
        *(uint4 *)(ks[buf] + row1 * KS_STRIDE + c8 * 16) = kr1;                                    \
        *(uint2 *)(vs[buf] + row0 * VS_STRIDE + c8 * 16) = make_uint2(vr0.x, vr0.y);               \
        *(uint2 *)(vs[buf] + row0 * VS_STRIDE + c8 * 16 + 8) = make_uint2(vr0.z, vr0.w);           \
        *(uint2 *)(vs[buf] + row1 * VS_STRIDE + c8 * 16) = make_uint2(vr1.x, vr1.y);               \
        *(uint2 *)(vs[buf] + row1 * VS_STRIDE + c8 * 16 + 8) = make_uint2(vr1.z, vr1.w);           \
    } while (0)

    f32x16 oacc[2], lacc, cneg;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        oacc[0][r] = 0.f;
        oacc[1][r] = 0.f;
        lacc[r] = 0.f;
        cneg[r] = 0.f;
    }
    float ref = 0.f;   // the running reference r (log2 units) of this lane's query; -r lives in cneg
    float lsum_v = 0.f;   // !SUM_MFMA: the row sum of this lane's half of the keys, f32 VALU adds
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    const int ntiles = (S + 63) / 64;
    kr0 = *(const uint4 *)(kbase + ko0);
    kr1 = *(const uint4 *)(kbase + ko1);
    vr0 = *(const uint4 *)(vbase + vo0);
    vr1 = *(const uint4 *)(vbase + vo1);
    ATT_LSTORE(0);
    __syncthreads();
    // One 64-key tile.  FIRST: the reference is taken from this tile's maximum (unconditionally: l >= 1 from then on);
    // LAST: keys >= S masked (the only tile that can hold any).
    auto tile = [&](int j, auto first_tag, auto last_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        const int buf = j & 1;
        if (!LAST && !(ABL & 8)) ATT_GLOAD_NEXT();
        // ---- S^T - r = K Q^T + (-r) : two 32-kv blocks, the reference enters as the C operand -----------
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const char *kp = ks[buf] + (kb * 32 + ql) * KS_STRIDE + hf * 16;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const bf16x8 kf = *(const bf16x8 *)(kp + s4 * 32);
                if (!(ABL & 1)) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s4], s4 == 0 ? cneg : st[kb], 0, 0, 0);
                else if (s4 == 0) { st[kb] = cneg; st[kb][0] += (float)kf[0]; }
            }
        }
        if constexpr (LAST) {  // mask kv >= S
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = j * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (kv >= S) st[kb][r] = -1e30f;
                }
        }
        // ---- lagged reference: does any score of the tile exceed r by more than 2^THR? ----------------
        float mloc = st[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kb][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        if ((ABL & 32) ? FIRST : (FIRST || __builtin_amdgcn_ballot_w64(mloc > ATT_RESCALE_THR) != 0ull)) {  // wave-uniform; rare after the first tile
            const float delta = FIRST ? mloc : fmaxf(mloc, 0.f);   // the reference only moves up
            ref += delta;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] -= delta;
            if (!FIRST) {   // (first tile: O and l are still zero)
                const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    oacc[0][r] *= alpha;
                    oacc[1][r] *= alpha;
                    if (SUM_MFMA) lacc[r] *= alpha;
                }
                if (!SUM_MFMA) lsum_v *= alpha;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) cneg[r] = -ref;
        }
        // ---- P^T = exp2(S^T - r) -> bf16;  O^T += V^T P^T;  l += 1^T P^T -------------------------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                f32x8 p8;
#pragma unroll
                for (int i = 0; i < 8; ++i) p8[i] = (ABL & 2) ? st[kb][8 * j2 + i] : __builtin_amdgcn_exp2f(st[kb][8 * j2 + i]);
                const bf16x8 pf = __builtin_convertvector(p8, bf16x8);  // 4 x v_cvt_pk_bf16_f32
                if (!SUM_MFMA) lsum_v += ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
                const int kvoff = kb * 32 + 16 * j2 + 4 * hf;  // + {0..3} and + 8 + {0..3}
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    const char *vp = vs[buf] + (eb * 32 + ql) * VS_STRIDE + kvoff * 2;
                    const bf16x4 lo = *(const bf16x4 *)vp;
                    const bf16x4 hi = *(const bf16x4 *)(vp + 16);
                    bf16x8 vf;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        vf[i] = lo[i];
                        vf[4 + i] = hi[i];
                    }
                    if (!(ABL & 4)) oacc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[eb], 0, 0, 0);
                    else oacc[eb][0] += (float)vf[0] + (float)pf[0];
                }
                if (SUM_MFMA) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
            }
        if (!LAST && !(ABL & 8)) ATT_LSTORE(buf ^ 1);
        if (!(ABL & 16)) __syncthreads();
    };
    if (ntiles == 1) {
        tile(0, std::true_type{}, std::true_type{});
    } else {
        tile(0, std::true_type{}, std::false_type{});
        for (int j = 1; j < ntiles - 1; ++j) tile(j, std::false_type{}, std::false_type{});
        tile(ntiles - 1, std::false_type{}, std::true_type{});
    }
    // ---- finalize: O^T[e][q] / l -> att[b*S + q][h*64 + e] (every row of the ones-product holds the query's sum) -----
    if (q < S) {
        const float inv = 1.0f / (SUM_MFMA ? lacc[0] : lsum_v + __shfl_xor(lsum_v, 32));
        bf16_t *op = att + ((long)b * S + q) * d + h * 64;
#pragma unroll
        for (int eb = 0; eb < 2; ++eb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e0 = eb * 32 + 8 * g + 4 * hf;
                const unsigned lo = pack2(oacc[eb][4 * g + 0] * inv, oacc[eb][4 * g + 1] * inv);
                const unsigned hi = pack2(oacc[eb][4 * g + 2] * inv, oacc[eb][4 * g + 3] * inv);
                *(uint2 *)(op + e0) = make_uint2(lo, hi);
            }
    }
}


template <bool SM, int ABL, int OCC>
float run(const bf16_t *qk, const bf16_t *vt, bf16_t *att, int B, int H, int S, int S_pad, int d, int iters) {
    const int n_q = (S + 127) / 128, n_bh = B * H;
    const int grid = (n_bh + 7) / 8 * 8 * n_q;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) enc_attn_kernel<SM, ABL, OCC><<<grid, 256>>>(qk, vt, att, H, S, S_pad, d, n_q, n_bh);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) enc_attn_kernel<SM, ABL, OCC><<<grid, 256>>>(qk, vt, att, H, S, S_pad, d, n_q, n_bh);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, H = 20, S = 1500, S_pad = 1536, d = 1280;
    const size_t M = (size_t)B * S;
    std::vector<bf16_t> hq((M + 64) * 2 * d), hv((size_t)B * H * 64 * S_pad);
    unsigned x = 1;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((float)(x >> 8) / 16777216.0f - 0.5f) * 2.0f; };
    auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); };
    for (auto &v : hq) v = bf(rnd() * 1.5f);
    for (auto &v : hv) v = bf(rnd());
    bf16_t *qk, *vt, *att;
    hipMalloc(&qk, hq.size() * 2); hipMalloc(&vt, hv.size() * 2); hipMalloc(&att, M * d * 2);
    hipMemcpy(qk, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, hv.data(), hv.size() * 2, hipMemcpyHostToDevice);
    const double gf = 4.0 * B * S * (double)S * d / 1e9;
#define RUN(SM, ABL, OCC, what) { float us = run<SM, ABL, OCC>(qk, vt, att, B, H, S, S_pad, d, 20); \
        printf("%-58s %8.1f us  (%5.0f TF/s-equivalent)\n", what, us, gf / us * 1e-3 * 1e3); }
    RUN(true, 0, 3, "full kernel, ones-MFMA row sum, 3 waves/SIMD");
    RUN(false, 0, 3, "full kernel, VALU row sum, 3 waves/SIMD");
    RUN(false, 0, 2, "full kernel, VALU row sum, 2 waves/SIMD");
    RUN(false, 32, 3, "never rescale after the first tile");
    RUN(false, 2, 3, "no exp (cvt only)");
    RUN(false, 1, 3, "no QK^T MFMAs");
    RUN(false, 4, 3, "no PV MFMAs");
    RUN(false, 5, 3, "no MFMAs at all");
    RUN(false, 8, 3, "no global->LDS staging (same tile re-read)");
    RUN(false, 16, 3, "no barrier (racy)");
    RUN(false, 24, 3, "no staging, no barrier");
    RUN(false, 26, 3, "no staging, no barrier, no exp");
    RUN(false, 31, 3, "no staging / barrier / exp / MFMAs (LDS reads + max + cvt)");
    return 0;
}
