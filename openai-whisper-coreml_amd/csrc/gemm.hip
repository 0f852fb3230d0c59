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

#include <atomic>
#include <type_traits>

#include "model.h"

// No implicit FMA contraction in this file: the two tile shapes (and their different epilogue code paths) must round
// every output identically -- whether `gelu(v) + pos` or `c + (acc + bias)` became an FMA used to depend on the
// surrounding code (tests: the 256 tile is bitwise equal to the 128 tile, kernel by kernel and on a whole model).
// Every FMA that is wanted is written as fmaf().
#pragma clang fp contract(off)

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

// GELU(x) = x * Phi(x) with Phi from erfc(z) ~= P(t) exp(-z^2), t = 1 / (1 + 0.3275911 z) (Abramowitz & Stegun 7.1.26,
// |error| <= 1.5e-7 on erf): |GELU error| <= 4.3e-7 absolute over [-12, 12], i.e. >= 20x below the bf16 rounding of
// every activation that is not itself below 1e-4.  Using erfc on the negative side keeps the tail relatively accurate
// (no 1 - erf cancellation).  17 VALU operations, no branches: the libm erff costs ~3x that in the fc1 epilogue.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float h = 0.5f * t * poly * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
    return x * (x < 0.f ? h : 1.0f - h);
}
// The same function on TWO values at once (round 6): every full-rate operation as a packed-f32 instruction (v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32: two lanes' worth of work per issue slot), the two quarter-rate ones (v_rcp_f32, v_exp_f32) and the
// select per value.  Operation for operation the sequence of gelu_erf above, IEEE per component: the same bits (the tile
// shapes' epilogues mix the two forms; the bitwise tile tests hold them together).  The fc1 epilogue is VALU work with the
// matrix pipe idle (profiles/r06_pmc_encoder_stalls.txt: ~36 % of a 256 x 256 tile's time).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2 z = ax * (f32x2){0.70710678118654752440f, 0.70710678118654752440f};
    const f32x2 den = __builtin_elementwise_fma((f32x2){0.3275911f, 0.3275911f}, z, (f32x2){1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f32x2 poly = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(t, poly, (f32x2){1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(t, poly, (f32x2){-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(t, poly, (f32x2){0.254829592f, 0.254829592f});
    const f32x2 e = -(z * z) * (f32x2){1.4426950408889634f, 1.4426950408889634f};
    const f32x2 ex = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    const f32x2 h = (f32x2){0.5f, 0.5f} * t * poly * ex;
    const f32x2 g = {x.x < 0.f ? h.x : 1.0f - h.x, x.y < 0.f ? h.y : 1.0f - h.y};
    return x * g;
}
// f32 -> bf16, round to nearest even: v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
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

// Epilogue of one wave's MI x NJ accumulator fragments.  Fragment (i, j): this lane holds column
// n = ncol0 + j*16 and rows mrow0 + i*16 + {0..3}.  Everything that depends only on the row (the batched
// row -> address map, which needs an integer division) or only on the column (bias, head / feature split)
// is computed once per row / column, not per element: the epilogue is VALU work that nothing overlaps when a
// CU holds a single workgroup.
template <int EPI, int MI, int NJ>
__device__ __forceinline__ void store_tile(const GemmDev &p, const f32x4 (&acc)[MI][NJ], int mrow0, int ncol0) {
    float bv[NJ];
    long coff[NJ];   // column part of the destination offset (EPI_XKV; V^T part of EPI_QKV_ENC)
    bool vpart[NJ];  // EPI_QKV_ENC: this column belongs to the value projection
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = ncol0 + j * 16;
        const int nc = n < p.N ? n : p.N - 1;
        bv[j] = p.bias ? p.bias[nc] : 0.f;
        coff[j] = 0;
        vpart[j] = false;
        if (EPI == EPI_XKV) {  // n -> (kv, h, e); out[kv][b][h][s][e]
            const int kv = nc / p.d_model, hn = nc - kv * p.d_model, h = hn >> 6, e = hn & 63;
            coff[j] = ((long)(kv * p.batch) * p.n_head + h) * p.seq * 64 + e;
        } else if (EPI == EPI_QKV_ENC) {  // value columns -> V^T [b][h][e][seq_pad]
            vpart[j] = nc >= 2 * p.d_model;
            const int hn = nc - 2 * p.d_model, h = hn >> 6, e = hn & 63;
            coff[j] = ((long)h * 64 + e) * p.seq_pad;
        }
    }
    const unsigned rpb = (unsigned)p.c_rpb, seq = (unsigned)(p.seq > 0 ? p.seq : 1);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mb = mrow0 + i * 16;  // first of 4 consecutive rows (mb % 4 == 0)
        if (mb >= p.M) continue;
        if (EPI == EPI_QKV_ENC) {
            // 4 consecutive frames of one chunk -> one 8-B store.  Only when the 4 frames start a 4-key group of the chunk
            // (sq % 4 == 0: wm_att_vt_pos permutes whole 4-key groups); otherwise -- a chunk length that is not a multiple
            // of 4 puts later chunks' rows at sq % 4 != 0 -- the per-element path below places every frame on its own
            const unsigned b = (unsigned)mb / seq, sq = (unsigned)mb - b * seq;
            const long roff = (long)b * p.n_head * 64 * p.seq_pad + wm_att_vt_pos(sq);   // sq % 4 == 0: the 4 frames stay together
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (!vpart[j] || ncol0 + j * 16 >= p.N) continue;
                const f32x4 &c = acc[i][j];
                if ((sq & 3u) == 0 && sq + 3 < seq && mb + 3 < p.M) {
                    unsigned lo = (unsigned)f2bf(c[0] + bv[j]) | ((unsigned)f2bf(c[1] + bv[j]) << 16);
                    unsigned hi = (unsigned)f2bf(c[2] + bv[j]) | ((unsigned)f2bf(c[3] + bv[j]) << 16);
                    *(uint2 *)(p.vt + roff + coff[j]) = make_uint2(lo, hi);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned m = (unsigned)mb + r;
                        if ((int)m >= p.M) continue;
                        const unsigned b2 = m / seq, s2 = m - b2 * seq;
                        p.vt[(long)b2 * p.n_head * 64 * p.seq_pad + wm_att_vt_pos(s2) + coff[j]] = f2bf(c[r] + bv[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned m = (unsigned)mb + r;
            if ((int)m >= p.M) continue;
            long roff;
            if (EPI == EPI_XKV) {
                const unsigned b = m / seq, sq = m - b * seq;
                roff = ((long)b * p.n_head * p.seq + sq) * 64;
            } else {
                const unsigned q = m / rpb, rem = m - q * rpb;
                roff = (long)q * p.c_bstride + (long)rem * p.c_rstride;
            }
            const float *pos_row = EPI == EPI_CONV2_F32 ? p.pos + (long)(m % rpb) * p.N : nullptr;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = ncol0 + j * 16;
                if (n >= p.N) continue;
                const float v = acc[i][j][r] + bv[j];
                if (EPI == EPI_XKV) {
                    ((bf16_t *)p.C)[roff + coff[j]] = f2bf(v);
                } else if (EPI == EPI_QKV_ENC) {
                    if (!vpart[j]) ((bf16_t *)p.C)[roff + n] = f2bf(n < p.d_model ? v * WM_ENC_QSCALE : v);
                } else if (EPI == EPI_BIAS_BF16) {
                    ((bf16_t *)p.C)[roff + n] = f2bf(v);
                } else if (EPI == EPI_GELU_BF16) {
                    ((bf16_t *)p.C)[roff + n] = f2bf(gelu_erf(v));
                } else if (EPI == EPI_RESID_F32) {
                    ((float *)p.C)[roff + n] += v;
                } else if (EPI == EPI_CONV2_F32) {
                    ((float *)p.C)[roff + n] = gelu_erf(v) + pos_row[n];
                } else {  // EPI_F32
                    ((float *)p.C)[roff + n] = v;
                }
            }
        }
    }
}

// LDS-staged epilogue for the bf16-output epilogues: each wave transposes its fragments through a private LDS region
// (64 rows x 64 columns per pass, 144-byte row stride) so that the global stores are 16 bytes per lane, 8 lanes = one
// 128-byte line of a C row -- 16 store instructions per lane for a 128 x 64 sub-tile instead of 128 two-byte ones.
// Wave-local: no workgroup barrier (the caller guarantees every wave is done reading the main loop's LDS image).
// Requires N % 64 == 0 (a wave's 64 columns are entirely inside C and, for EPI_QKV_ENC / EPI_XKV, inside one head).
constexpr int STAGE_ROW_BYTES = 144;
constexpr int STAGE_WAVE_BYTES = 64 * STAGE_ROW_BYTES;  // 9216 B per wave

template <int EPI, int MI, int NJ>
__device__ __forceinline__ void store_tile_staged(const GemmDev &p, const f32x4 (&acc)[MI][NJ], int mwave0, int nwave0,
                                                  char *L, int lane) {
    static_assert(NJ == 4 && MI % 4 == 0, "a wave sub-tile is (MI x 16) x 64");
    static_assert(EPI == EPI_BIAS_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV_ENC || EPI == EPI_XKV, "bf16 outputs");
    if (nwave0 >= p.N) return;  // wave-uniform
    const int frow = lane & 15, fq = lane >> 4;
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = p.bias ? p.bias[nwave0 + j * 16 + frow] : 0.f;
    // wave-uniform column part of the destination
    long coff = nwave0;
    if (EPI == EPI_XKV) {  // the wave's 64 columns are one (kv, head): out[kv][b][h][s][64]
        const int kv = nwave0 / p.d_model, h = (nwave0 - kv * p.d_model) >> 6;
        coff = ((long)(kv * p.batch) * p.n_head + h) * p.seq * 64;
    }
    const unsigned rpb = (unsigned)p.c_rpb, seq = (unsigned)(p.seq > 0 ? p.seq : 1);
    const bool qpart = EPI == EPI_QKV_ENC && nwave0 < p.d_model;   // a wave's 64 columns are inside one of q | k | v
    const int rrow = lane >> 3, chunk = lane & 7;
#pragma unroll
    for (int ps = 0; ps < MI / 4; ++ps) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 v = {acc[ps * 4 + ii][j][r] + bv[j], acc[ps * 4 + ii][j][r + 1] + bv[j]};
                    if (EPI == EPI_GELU_BF16) v = gelu_erf2(v);
                    if (EPI == EPI_QKV_ENC && qpart) v = v * (f32x2){WM_ENC_QSCALE, WM_ENC_QSCALE};   // wave-uniform: the query third, see model.h
                    *(bf16_t *)(L + (ii * 16 + fq * 4 + r) * STAGE_ROW_BYTES + (j * 16 + frow) * 2) = f2bf(v.x);
                    *(bf16_t *)(L + (ii * 16 + fq * 4 + r + 1) * STAGE_ROW_BYTES + (j * 16 + frow) * 2) = f2bf(v.y);
                }
        // row -> (batch, row-in-batch): ONE division per pass (the lane's first row), then 8 rows further per step
        // (16 divisions per lane per tile were ~1 us of the epilogue; rows-per-batch >= 8 is checked by the caller)
        const unsigned div = EPI == EPI_XKV ? seq : rpb;
        unsigned m = (unsigned)(mwave0 + ps * 64 + rrow);
        unsigned q = m / div, rem = m - q * div;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = t * 8 + rrow;
            const uint4 val = *(const uint4 *)(L + row * STAGE_ROW_BYTES + chunk * 16);
            if ((int)m < p.M) {
                const long roff = EPI == EPI_XKV ? ((long)q * p.n_head * p.seq + rem) * 64
                                                 : (long)q * p.c_bstride + (long)rem * p.c_rstride;
                *(uint4 *)((bf16_t *)p.C + roff + coff + chunk * 8) = val;
            }
            m += 8;
            rem += 8;
            if (rem >= div) { rem -= div; ++q; }
        }
    }
}

// The same for the f32 residual update C += A.W^T + bias: 32 rows x 64 columns per pass (272-byte row stride), then
// every lane reads 4 consecutive floats, adds them to 16 bytes of C and writes them back: 32 read-modify-writes of
// 16 bytes per lane for a 128 x 64 sub-tile instead of 128 of 4 bytes.
constexpr int STAGE_F32_ROW_BYTES = 272;

template <int MI, int NJ>
__device__ __forceinline__ void resid_tile_staged(const GemmDev &p, const f32x4 (&acc)[MI][NJ], int mwave0, int nwave0,
                                                  char *L, int lane) {
    static_assert(NJ == 4 && MI % 2 == 0, "a wave sub-tile is (MI x 16) x 64");
    if (nwave0 >= p.N) return;  // wave-uniform
    const int frow = lane & 15, fq = lane >> 4;
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = p.bias ? p.bias[nwave0 + j * 16 + frow] : 0.f;
    const unsigned rpb = (unsigned)p.c_rpb;
    const unsigned q_last = (unsigned)(p.M - 1) / rpb, rem_last = (unsigned)(p.M - 1) - q_last * rpb;  // (uniform)
    const int rrow = lane >> 4, c4 = lane & 15;
    constexpr int NP = MI / 2;  // passes of 32 rows
    // The read-modify-write is load LATENCY (a dependent HBM round trip per pass): the eight 16-byte loads of a pass are
    // requested together, AND one pass ahead -- the loads of pass ps + 1 are in flight while pass ps is transposed through
    // LDS, added and stored (two register sets; the main loop's operand registers are dead by now).  Rows past M re-read
    // row M - 1 and store nothing (selects, no branches: the loads of a pass must stay one batch).
    float4 c[2][8];
    auto request = [&](int ps, float4 (&cc)[8]) {
        // row -> (batch, row-in-batch): one division per pass, then 4 rows further per step (rows-per-batch >= 8)
        unsigned m = (unsigned)(mwave0 + ps * 32 + rrow);
        unsigned q = m / rpb, rem = m - q * rpb;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bool ok = (int)m < p.M;
            const unsigned mq = ok ? q : q_last, mr = ok ? rem : rem_last;
            cc[t] = *(const float4 *)((const float *)p.C + (long)mq * p.c_bstride + (long)mr * p.c_rstride + nwave0 + c4 * 4);
            m += 4;
            rem += 4;
            if (rem >= rpb) { rem -= rpb; ++q; }
        }
    };
    request(0, c[0]);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        if (ps + 1 < NP) request(ps + 1, c[(ps + 1) & 1]);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *(float *)(L + (ii * 16 + fq * 4 + r) * STAGE_F32_ROW_BYTES + (j * 16 + frow) * 4) =
                        acc[ps * 2 + ii][j][r] + bv[j];
        unsigned m = (unsigned)(mwave0 + ps * 32 + rrow);
        unsigned q = m / rpb, rem = m - q * rpb;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = t * 4 + rrow;
            const float4 v = *(const float4 *)(L + row * STAGE_F32_ROW_BYTES + c4 * 16);
            if ((int)m < p.M) {
                float4 r = c[ps & 1][t];
                r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
                *(float4 *)((float *)p.C + (long)q * p.c_bstride + (long)rem * p.c_rstride + nwave0 + c4 * 4) = r;
            }
            m += 4;
            rem += 4;
            if (rem >= rpb) { rem -= rpb; ++q; }
        }
    }
}

// XCD-aware, bijective workgroup -> tile map: the 8 XCDs (private L2s) each get a contiguous range of tiles,
// walked in GM x tiles_n groups so that the tiles an XCD runs concurrently share A and W panels in its L2.
__device__ __forceinline__ void tile_of_workgroup(const GemmDev &p, int &tm, int &tn) {
    const int nwg = p.tiles_m * p.tiles_n;
    int wg = blockIdx.x;
    const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    const int GM = p.group_m;
    const int per_group = GM * p.tiles_n;
    const int grp = wg / per_group, in_grp = wg % per_group;
    const int first_m = grp * GM;
    const int gsz = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmDev p) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][TILE_BYTES];  // [buf][A|B]

    int tm, tn;
    tile_of_workgroup(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- staging addresses: 4 passes x (32 rows x 8 chunks of 16 B) per operand ---------
    // thread -> (row = pass*32 + tid/8, physical chunk = tid%8); it fetches logical chunk
    // (pchunk ^ (row & 7)) so that the lane-linear LDS image is the swizzled one.
    const int srow = tid >> 3, pch = tid & 7;
    const bf16_t *a_src[4];
    const bf16_t *w_src[4];
    const unsigned arpb = p.a_rpb > 0x7fffffffL ? 0x7fffffffu : (unsigned)p.a_rpb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + srow;
        const int lch = pch ^ (row & 7);
        unsigned m = (unsigned)(m0 + row);
        if (m > (unsigned)(p.M - 1)) m = (unsigned)(p.M - 1);  // clamp: tail rows are masked in the epilogue
        const unsigned aq = m / arpb, ar = m - aq * arpb;  // 32-bit: see the 256 kernel
        a_src[i] = p.A + (long)aq * p.a_bstride + (long)ar * p.a_rstride + lch * 8;
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
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV_ENC || EPI == EPI_XKV) {
        // bf16 outputs leave through LDS (the loop's last __syncthreads() means nobody reads the operand tiles any more);
        // the V^T third of the QKV projection keeps its transposed register path
        const int nwave0 = n0 + wc * 64;
        if (p.N % 64 == 0 && (EPI == EPI_XKV ? p.seq : p.c_rpb) >= 8 &&
            !(EPI == EPI_QKV_ENC && nwave0 >= 2 * p.d_model)) {  // wave-uniform
            store_tile_staged<EPI, 4, 4>(p, acc, m0 + wr * 64, nwave0, &lds[0][0][0] + wave * STAGE_WAVE_BYTES, lane);
            return;
        }
    }
    if constexpr (EPI == EPI_RESID_F32) {
        if (p.N % 64 == 0 && p.c_rpb >= 8) {
            resid_tile_staged<4, 4>(p, acc, m0 + wr * 64, n0 + wc * 64, &lds[0][0][0] + wave * STAGE_WAVE_BYTES, lane);
            return;
        }
    }
    store_tile<EPI, 4, 4>(p, acc, m0 + wr * 64 + fq * 4, n0 + wc * 64 + frow);
}

// =================================================================================================
// 64 x 64 x 64 tile for products that cannot fill the chip with 128 x 128 tiles (round 6): ONE 30 s chunk (M = 1500) -- the
// reference's own flow (Whisper.swift:23-31, one chunk of Whisper-small) -- has 72 tiles of 128 x 128 for its N = 768 products
// (out-projection, fc2 with K = 3072, conv2) on 256 CUs; as 64 x 64 tiles they are 288 workgroups.  Same structure as the
// 128 tile (4 waves as 2 x 2, LDS-DMA double buffer, XOR swizzle), each wave owns 32 x 32 = 2 x 2 fragments; 2 x 16 KiB of LDS,
// so several workgroups share a CU.  Every output element is accumulated over K in the same order by the same MFMA and
// finished by the same epilogue arithmetic: bitwise equal to the other tiles (test).
constexpr int BM3 = 64, BN3 = 64;
constexpr int TILE3_BYTES = BM3 * BK * 2;   // 8 KiB per operand tile
constexpr int STAGE3_BYTES = 2 * TILE3_BYTES;  // A + W of one K-tile
constexpr int NST3 = 4;                      // K-tiles in flight (ring of 4 x 16 KiB)

#define WM_DSR3(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

// A single-chunk product is LATENCY-bound per K-tile (4 MFMAs per wave against one HBM / L2 round trip), and with ~1
// workgroup per CU nothing else hides it: K-tiles are streamed THREE ahead of the one being multiplied (LDS-DMA into a
// 4-slot ring, counted s_waitcnt vmcnt(8), one barrier per K-tile; the loads past the end of K re-request the last tile so
// the count stays constant).  ds_reads in inline asm, as in the 256 kernel: hipcc would otherwise drain vmcnt(0) in front
// of every LDS read that may alias an outstanding LDS-DMA.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm64_bf16_kernel(GemmDev p) {
    __shared__ __attribute__((aligned(1024))) char lds[NST3 * STAGE3_BYTES];
    int tm, tn;
    tile_of_workgroup(p, tm, tn);
    const int m0 = tm * BM3, n0 = tn * BN3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // staging: 2 passes x (32 rows x 8 chunks of 16 B) per operand; thread -> (row = pass*32 + tid/8, physical chunk tid%8)
    const int srow = tid >> 3, pch = tid & 7;
    const bf16_t *a_src[2];
    const bf16_t *w_src[2];
    const unsigned arpb = p.a_rpb > 0x7fffffffL ? 0x7fffffffu : (unsigned)p.a_rpb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + srow;
        const int lch = pch ^ (row & 7);
        unsigned m = (unsigned)(m0 + row);
        if (m > (unsigned)(p.M - 1)) m = (unsigned)(p.M - 1);  // clamp: tail rows are masked in the epilogue
        const unsigned aq = m / arpb, ar = m - aq * arpb;
        a_src[i] = p.A + (long)aq * p.a_bstride + (long)ar * p.a_rstride + lch * 8;
        long n = n0 + row;
        if (n > p.N - 1) n = p.N - 1;
        w_src[i] = p.W + n * (long)p.K + lch * 8;
    }
    const int nk = p.K / BK;
    auto stage = [&](int slot, int kt) {   // 4 LDS-DMA instructions per wave
        const long ko = (long)(kt < nk ? kt : nk - 1) * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char *da = lds + slot * STAGE3_BYTES + (i * 32 + wave * 8) * 128;   // wave-uniform base; the hardware adds lane * 16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a_src[i] + ko),
                                             (__attribute__((address_space(3))) void *)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_src[i] + ko),
                                             (__attribute__((address_space(3))) void *)(da + TILE3_BYTES), 16, 0, 0);
        }
    };
    // fragment read addresses (slot 0; the slot and the operand are immediates): row r, k-step ks -> r * 128 + ((ks*4+fq) ^ (r & 7)) * 16
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    unsigned aa[2][2], wa[2][2];   // [fragment][k-step]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ra = wr * 32 + i * 16 + frow, rb = wc * 32 + i * 16 + frow;
            aa[i][ks] = lds0 + (unsigned)(ra * 128 + (((ks * 4 + fq) ^ (ra & 7)) << 4));
            wa[i][ks] = lds0 + (unsigned)(rb * 128 + (((ks * 4 + fq) ^ (rb & 7)) << 4));
        }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    stage(1, 1);
    stage(2, 2);
    auto step = [&](int kt, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int SO = SLOT * STAGE3_BYTES;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's pieces of K-tile kt have landed (kt + 1, kt + 2 may be in flight)
        __builtin_amdgcn_s_barrier();                        // ... everybody's; and everybody is done reading K-tile kt - 1
        stage((SLOT + 3) % NST3, kt + 3);                    // into the slot K-tile kt - 1 lived in
        bf16x8 af[2][2], bfr[2][2];
        WM_DSR3(af[0][0], aa[0][0], SO); WM_DSR3(af[1][0], aa[1][0], SO);
        WM_DSR3(bfr[0][0], wa[0][0], SO + TILE3_BYTES); WM_DSR3(bfr[1][0], wa[1][0], SO + TILE3_BYTES);
        WM_DSR3(af[0][1], aa[0][1], SO); WM_DSR3(af[1][1], aa[1][1], SO);
        WM_DSR3(bfr[0][1], wa[0][1], SO + TILE3_BYTES); WM_DSR3(bfr[1][1], wa[1][1], SO + TILE3_BYTES);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(af[0][0]), "+v"(af[1][0]), "+v"(bfr[0][0]), "+v"(bfr[1][0]), "+v"(af[0][1]), "+v"(af[1][1]),
                       "+v"(bfr[0][1]), "+v"(bfr[1][1])::"memory");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)   // (k ascending within the tile: the order every tile shape accumulates in)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bfr[j][ks], acc[i][j], 0, 0, 0);
    };
    int kt = 0;
    for (; kt + 4 <= nk; kt += 4) {
        step(kt, std::integral_constant<int, 0>{});
        step(kt + 1, std::integral_constant<int, 1>{});
        step(kt + 2, std::integral_constant<int, 2>{});
        step(kt + 3, std::integral_constant<int, 3>{});
    }
    if (kt < nk) { step(kt, std::integral_constant<int, 0>{}); ++kt; }   // (workgroup-uniform remainders)
    if (kt < nk) { step(kt, std::integral_constant<int, 1>{}); ++kt; }
    if (kt < nk) { step(kt, std::integral_constant<int, 2>{}); ++kt; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested tail tiles: nothing may land in LDS after the kernel's end
    store_tile<EPI, 2, 2>(p, acc, m0 + wr * 32 + fq * 4, n0 + wc * 32 + frow);
}

// =================================================================================================
// The 128 x 128 tile as a 3-stage LDS-DMA pipeline (round 6): same tile, same wave layout, same epilogues as
// gemm_bf16_kernel above, but K-tiles are streamed TWO ahead of the one being multiplied (ring of 3 x 32 KiB, counted
// s_waitcnt vmcnt(8), one barrier per K-tile, inline-asm ds_reads) instead of double-buffered with a full drain per
// K-tile.  For grids of at most one workgroup per CU (160-256 tiles: the N = d products of two 30 s chunks at large-v2,
// the QKV product of Whisper-small x 1) nothing else hides the LDS-DMA round trip; 96 KiB of LDS = one workgroup per CU, so
// larger grids keep the two-per-CU double-buffer kernel.  Same accumulation order: bitwise equal (test).
constexpr int NST2 = 3;
constexpr int STAGE2_BYTES = 2 * TILE_BYTES;   // A + W of one K-tile: 32 KiB

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm128p_bf16_kernel(GemmDev p) {
    extern __shared__ __attribute__((aligned(1024))) char lds_dyn[];   // NST2 * STAGE2_BYTES
    char *lds = lds_dyn;
    int tm, tn;
    tile_of_workgroup(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int srow = tid >> 3, pch = tid & 7;
    const bf16_t *a_src[4];
    const bf16_t *w_src[4];
    const unsigned arpb = p.a_rpb > 0x7fffffffL ? 0x7fffffffu : (unsigned)p.a_rpb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + srow;
        const int lch = pch ^ (row & 7);
        unsigned m = (unsigned)(m0 + row);
        if (m > (unsigned)(p.M - 1)) m = (unsigned)(p.M - 1);  // clamp: tail rows are masked in the epilogue
        const unsigned aq = m / arpb, ar = m - aq * arpb;
        a_src[i] = p.A + (long)aq * p.a_bstride + (long)ar * p.a_rstride + lch * 8;
        long n = n0 + row;
        if (n > p.N - 1) n = p.N - 1;
        w_src[i] = p.W + n * (long)p.K + lch * 8;
    }
    const int nk = p.K / BK;
    auto stage = [&](int slot, int kt) {   // 8 LDS-DMA instructions per wave
        const long ko = (long)(kt < nk ? kt : nk - 1) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            char *da = lds + slot * STAGE2_BYTES + (i * 32 + wave * 8) * 128;   // wave-uniform base; the hardware adds lane * 16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a_src[i] + ko),
                                             (__attribute__((address_space(3))) void *)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_src[i] + ko),
                                             (__attribute__((address_space(3))) void *)(da + TILE_BYTES), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    unsigned aa[4][2], wa[4][2];   // [fragment][k-step], slot 0 (slots 1 / 2: + 32 KiB as an immediate / in a second address set:
                                   // a ds_read immediate is 16 bits)
    unsigned aa2[4][2], wa2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ra = wr * 64 + i * 16 + frow, rb = wc * 64 + i * 16 + frow;
            aa[i][ks] = lds0 + (unsigned)(ra * 128 + (((ks * 4 + fq) ^ (ra & 7)) << 4));
            wa[i][ks] = lds0 + (unsigned)(rb * 128 + (((ks * 4 + fq) ^ (rb & 7)) << 4));
            aa2[i][ks] = aa[i][ks] + 2u * STAGE2_BYTES;
            wa2[i][ks] = wa[i][ks] + 2u * STAGE2_BYTES;
        }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    stage(1, 1);
    auto step = [&](int kt, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int SO = SLOT == 2 ? 0 : SLOT * STAGE2_BYTES;
        auto &A_ = SLOT == 2 ? aa2 : aa;
        auto &W_ = SLOT == 2 ? wa2 : wa;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's pieces of K-tile kt have landed (kt + 1 may be in flight)
        __builtin_amdgcn_s_barrier();                        // ... everybody's; and everybody is done reading K-tile kt - 1
        stage((SLOT + 2) % NST2, kt + 2);                    // into the slot K-tile kt - 1 lived in
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
            WM_DSR3(af[0], A_[0][ks], SO); WM_DSR3(af[1], A_[1][ks], SO); WM_DSR3(af[2], A_[2][ks], SO); WM_DSR3(af[3], A_[3][ks], SO);
            WM_DSR3(bfr[0], W_[0][ks], SO + TILE_BYTES); WM_DSR3(bfr[1], W_[1][ks], SO + TILE_BYTES);
            WM_DSR3(bfr[2], W_[2][ks], SO + TILE_BYTES); WM_DSR3(bfr[3], W_[3][ks], SO + TILE_BYTES);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3])::"memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
        step(kt, std::integral_constant<int, 0>{});
        step(kt + 1, std::integral_constant<int, 1>{});
        step(kt + 2, std::integral_constant<int, 2>{});
    }
    if (kt < nk) { step(kt, std::integral_constant<int, 0>{}); ++kt; }   // (workgroup-uniform remainders)
    if (kt < nk) { step(kt, std::integral_constant<int, 1>{}); ++kt; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested tail tiles have landed ...
    __builtin_amdgcn_s_barrier();                        // ... and nobody reads the operand tiles any more: LDS is the epilogues' now
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV_ENC || EPI == EPI_XKV) {
        const int nwave0 = n0 + wc * 64;
        if (p.N % 64 == 0 && (EPI == EPI_XKV ? p.seq : p.c_rpb) >= 8 &&
            !(EPI == EPI_QKV_ENC && nwave0 >= 2 * p.d_model)) {  // wave-uniform
            store_tile_staged<EPI, 4, 4>(p, acc, m0 + wr * 64, nwave0, lds + wave * STAGE_WAVE_BYTES, lane);
            return;
        }
    }
    if constexpr (EPI == EPI_RESID_F32) {
        if (p.N % 64 == 0 && p.c_rpb >= 8) {
            resid_tile_staged<4, 4>(p, acc, m0 + wr * 64, n0 + wc * 64, lds + wave * STAGE_WAVE_BYTES, lane);
            return;
        }
    }
    store_tile<EPI, 4, 4>(p, acc, m0 + wr * 64 + fq * 4, n0 + wc * 64 + frow);
}

// =================================================================================================
// 256 x 256 x 64 tile for the large encoder products: 512 threads = 8 wave64 as 2(M) x 4(N), each wave
// owns 128 x 64 of C (8 x 4 fragments, 128 accumulator registers); one workgroup per CU (128 KiB LDS).
//
// A K-tile (64 KiB in LDS, two buffers) is staged as four 16-KiB UNITS grouped by WHEN they are consumed:
//     S0 = A rows {0..63, 128..191}  (m-fragments 0-3 of both wave rows)      read in phase 1
//     S1 = W rows {wc*64 + 0..31}    (n-fragments 0-1 of the four wave columns) read in phase 1
//     S2 = W rows {wc*64 + 32..63}   (n-fragments 2-3)                          read in phase 2
//     S3 = A rows {64..127, 192..255} (m-fragments 4-7)                         read in phase 3
// and computed as four phases of 16 MFMAs (one 64 x 32 quadrant of the wave's C x K = 64):
//     P1 (A0,B0)   P2 (A0,B1)   P3 (A1,B1)   P4 (A1,B0: registers only)
// Every phase also issues ONE unit of the stream S0(t),S1(t),S2(t),S3(t),S0(t+1),... six units ahead of
// the phase that reads it (global_load_lds, 2 x 16 B per lane), and waits with a COUNTED s_waitcnt vmcnt(8):
// four units stay in flight across the barriers, the unit read by the next phase has landed.  A unit is
// re-staged no earlier than two phases after its last ds_read.  The two wave rows run one barrier apart
// (wr = 1 enters through an extra s_barrier), so on every SIMD one wave issues ds_reads / DMA while the
// other runs its MFMA cluster.  The ds_reads are inline asm: hipcc would otherwise put a vmcnt(0) in front
// of every LDS read that may alias an outstanding LDS-DMA and serialise the pipeline.
constexpr int BM2 = 256, BN2 = 256;
constexpr int UNIT_BYTES = 128 * BK * 2;   // 16 KiB
constexpr int BUF_BYTES = 4 * UNIT_BYTES;  // 64 KiB per K-tile

#define WM_DSR(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define WM_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int EPI>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(2, 2))) void gemm256_bf16_kernel(
    GemmDev p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF_BYTES];

    int tm, tn;
    tile_of_workgroup(p, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- staging sources: a unit is 2 passes of 64 unit-rows x 8 chunks of 16 B -----------------------
    // thread -> unit-row u = pass*64 + tid/8, physical chunk tid%8, fetching logical chunk (pch ^ (u & 7)).
    const int srow = tid >> 3, pch = tid & 7;
    const int lch8 = (pch ^ (srow & 7)) * 8;
    const bf16_t *a_src[2][2];  // [sub][pass]
    const bf16_t *w_src[2][2];
    const unsigned arpb = p.a_rpb > 0x7fffffffL ? 0x7fffffffu : (unsigned)p.a_rpb;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned m = (unsigned)(m0 + i * 128 + sub * 64 + srow);  // pass i = wave row i
            if (m > (unsigned)(p.M - 1)) m = (unsigned)(p.M - 1);     // clamp: tail rows are masked in the epilogue
            // 32-bit row -> (batch, row) division: the 64-bit one is ~300 VALU instructions, and four of them per thread
            // were 1.4 us of a 28-us K = 1280 tile (measured with in-kernel timestamps)
            const unsigned aq = m / arpb, ar = m - aq * arpb;
            a_src[sub][i] = p.A + (long)aq * p.a_bstride + (long)ar * p.a_rstride + lch8;
            long n = n0 + (2 * i + (srow >> 5)) * 64 + sub * 32 + (srow & 31);  // unit-row -> (wave col, row)
            if (n > p.N - 1) n = p.N - 1;
            w_src[sub][i] = p.W + n * (long)p.K + lch8;
        }
    auto issue = [&](const bf16_t *s0, const bf16_t *s1, int lds_off, long ko) {
        char *d0 = lds + lds_off + wave * 1024;  // wave-uniform base; the hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s0 + ko),
                                         (__attribute__((address_space(3))) void *)d0, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s1 + ko),
                                         (__attribute__((address_space(3))) void *)(d0 + 8192), 16, 0, 0);
    };
    // stream index -> (K-tile, unit): unit 0 = S0, 1 = S1, 2 = S2, 3 = S3
#define WM_ISSUE(unit, tile)                                                                                   \
    do {                                                                                                       \
        const int t_ = (tile);                                                                                 \
        const int off_ = (t_ & 1) * BUF_BYTES + (unit) * UNIT_BYTES;                                           \
        const long ko_ = (long)t_ * BK;                                                                        \
        if ((unit) == 0) issue(a_src[0][0], a_src[0][1], off_, ko_);                                           \
        else if ((unit) == 1) issue(w_src[0][0], w_src[0][1], off_, ko_);                                      \
        else if ((unit) == 2) issue(w_src[1][0], w_src[1][1], off_, ko_);                                      \
        else issue(a_src[1][0], a_src[1][1], off_, ko_);                                                       \
    } while (0)

    // ---- fragment read addresses (32-bit LDS offsets): unit-row u = wr*64 + i*16 + frow (A) or
    // wc*32 + j*16 + frow (W); the swizzle term only depends on frow, the k-step toggles chunk bit 2.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const unsigned sw0 = (unsigned)((fq ^ (frow & 7)) << 4);
    const unsigned a_rd = lds0 + (unsigned)((wr * 64 + frow) * 128) + sw0;
    const unsigned b_rd = lds0 + (unsigned)((wc * 32 + frow) * 128) + sw0;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4][2], b0[2][2], b1[2][2];  // [fragment][k-step]

#define WM_MFMA_QUAD(mq, bb, jb)                                                                               \
    do {                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
                    acc[(mq) * 4 + i][(jb) + j] =                                                              \
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bb[j][ks], acc[(mq) * 4 + i][(jb) + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
    } while (0)

    // One phase.  P: 0..3; ISSUE: stage stream unit (P + 2) & 3 of K-tile t + 1 (P < 2) or t + 2; VM: vmcnt.
#define WM_PHASE(P, ISSUE, VM, t)                                                                              \
    do {                                                                                                       \
        const unsigned bo_ = (unsigned)(((t) & 1) * BUF_BYTES);                                                \
        const unsigned ar0_ = a_rd + bo_, ar1_ = (a_rd ^ 64u) + bo_;                                           \
        const unsigned br0_ = b_rd + bo_, br1_ = (b_rd ^ 64u) + bo_;                                           \
        if ((P) == 0) {                                                                                        \
            WM_DSR(b0[0][0], br0_, UNIT_BYTES); WM_DSR(b0[0][1], br1_, UNIT_BYTES);                            \
            WM_DSR(b0[1][0], br0_, UNIT_BYTES + 2048); WM_DSR(b0[1][1], br1_, UNIT_BYTES + 2048);              \
            WM_DSR(a[0][0], ar0_, 0); WM_DSR(a[0][1], ar1_, 0);                                                \
            WM_DSR(a[1][0], ar0_, 2048); WM_DSR(a[1][1], ar1_, 2048);                                          \
            WM_DSR(a[2][0], ar0_, 4096); WM_DSR(a[2][1], ar1_, 4096);                                          \
            WM_DSR(a[3][0], ar0_, 6144); WM_DSR(a[3][1], ar1_, 6144);                                          \
        } else if ((P) == 1) {                                                                                 \
            WM_DSR(b1[0][0], br0_, 2 * UNIT_BYTES); WM_DSR(b1[0][1], br1_, 2 * UNIT_BYTES);                    \
            WM_DSR(b1[1][0], br0_, 2 * UNIT_BYTES + 2048); WM_DSR(b1[1][1], br1_, 2 * UNIT_BYTES + 2048);      \
        } else if ((P) == 2) {                                                                                 \
            WM_DSR(a[0][0], ar0_, 3 * UNIT_BYTES); WM_DSR(a[0][1], ar1_, 3 * UNIT_BYTES);                      \
            WM_DSR(a[1][0], ar0_, 3 * UNIT_BYTES + 2048); WM_DSR(a[1][1], ar1_, 3 * UNIT_BYTES + 2048);        \
            WM_DSR(a[2][0], ar0_, 3 * UNIT_BYTES + 4096); WM_DSR(a[2][1], ar1_, 3 * UNIT_BYTES + 4096);        \
            WM_DSR(a[3][0], ar0_, 3 * UNIT_BYTES + 6144); WM_DSR(a[3][1], ar1_, 3 * UNIT_BYTES + 6144);        \
        }                                                                                                      \
        if (ISSUE) WM_ISSUE(((P) + 2) & 3, (t) + ((P) < 2 ? 1 : 2));                                           \
        WM_VMCNT(VM);                                                                                          \
        __builtin_amdgcn_s_barrier();                                                                          \
        if ((P) == 0) {                                                                                        \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                \
                         : "+v"(b0[0][0]), "+v"(b0[0][1]), "+v"(b0[1][0]), "+v"(b0[1][1]), "+v"(a[0][0]),      \
                           "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]),          \
                           "+v"(a[3][0]), "+v"(a[3][1])::"memory");                                            \
            WM_MFMA_QUAD(0, b0, 0);                                                                            \
        } else if ((P) == 1) {                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                \
                         : "+v"(b1[0][0]), "+v"(b1[0][1]), "+v"(b1[1][0]), "+v"(b1[1][1])::"memory");          \
            WM_MFMA_QUAD(0, b1, 2);                                                                            \
        } else if ((P) == 2) {                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                \
                         : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]),          \
                           "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1])::"memory");                             \
            WM_MFMA_QUAD(1, b1, 2);                                                                            \
        } else {                                                                                               \
            WM_MFMA_QUAD(1, b0, 0);                                                                            \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
    } while (0)

    const int nk = p.K / BK;  // >= 2 (checked by the launcher)
    // prologue: stream units 0..5 = K-tile 0 complete + S0, S1 of K-tile 1
    WM_ISSUE(0, 0); WM_ISSUE(1, 0); WM_ISSUE(2, 0); WM_ISSUE(3, 0); WM_ISSUE(0, 1); WM_ISSUE(1, 1);
    WM_VMCNT(8);  // S0(0), S1(0) have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs one barrier behind row 0
    int t = 0;
    for (; t < nk - 2; ++t) {
        WM_PHASE(0, true, 8, t);
        WM_PHASE(1, true, 8, t);
        WM_PHASE(2, true, 8, t);
        WM_PHASE(3, true, 8, t);
    }
    // last two K-tiles: the stream ends, the counted waits shrink with it
    WM_PHASE(0, true, 8, t);
    WM_PHASE(1, true, 8, t);
    WM_PHASE(2, false, 6, t);
    WM_PHASE(3, false, 4, t);
    ++t;
    WM_PHASE(0, false, 2, t);
    WM_PHASE(1, false, 0, t);
    WM_PHASE(2, false, 0, t);
    WM_PHASE(3, false, 0, t);
    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with row 1's last barrier

    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV_ENC || EPI == EPI_XKV) {
        // bf16 outputs leave through LDS: after the barrier above every wave has finished reading the operand tiles
        const int nwave0 = n0 + wc * 64;
        if (p.N % 64 == 0 && (EPI == EPI_XKV ? p.seq : p.c_rpb) >= 8 &&
            !(EPI == EPI_QKV_ENC && nwave0 >= 2 * p.d_model)) {  // wave-uniform
            store_tile_staged<EPI, 8, 4>(p, acc, m0 + wr * 128, nwave0, lds + wave * STAGE_WAVE_BYTES, lane);
            return;
        }
    }
    if constexpr (EPI == EPI_RESID_F32) {
        if (p.N % 64 == 0 && p.c_rpb >= 8) {
            resid_tile_staged<8, 4>(p, acc, m0 + wr * 128, n0 + wc * 64, lds + wave * STAGE_WAVE_BYTES, lane);
            return;
        }
    }
    store_tile<EPI, 8, 4>(p, acc, m0 + wr * 128 + fq * 4, n0 + wc * 64 + frow);
#undef WM_PHASE
#undef WM_MFMA_QUAD
#undef WM_ISSUE
}

}  // namespace

// A/B and parity probes force one of the two tile shapes (the wmdbg_set_gemm_tile hook lives in debug_hooks.cpp)
int wm_gemm_set_tile_override(int tile) {
    if (tile != 0 && tile != 64 && tile != 128 && tile != 256) return WM_ERR_INVALID;
    g_wm_tuning.gemm_tile = tile;
    return WM_OK;
}

template <int EPI>
static int launch_gemm(const GemmDev &p, int tile, bool pipe128, int grid, hipStream_t s) {
    if (tile == 256) gemm256_bf16_kernel<EPI><<<grid, 512, 0, s>>>(p);
    else if (tile == 128 && pipe128) {
        static std::atomic<int> attr_set[64];   // per device: the kernel's dynamic-LDS allowance (96 KiB) is set once
        int dev = 0;
        WM_HIP(hipGetDevice(&dev));
        if (!attr_set[dev & 63].load(std::memory_order_acquire)) {
            WM_HIP(hipFuncSetAttribute((const void *)gemm128p_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, NST2 * STAGE2_BYTES));
            attr_set[dev & 63].store(1, std::memory_order_release);
        }
        gemm128p_bf16_kernel<EPI><<<grid, 256, NST2 * STAGE2_BYTES, s>>>(p);
    } else if (tile == 128) gemm_bf16_kernel<EPI><<<grid, 256, 0, s>>>(p);
    else gemm64_bf16_kernel<EPI><<<grid, 256, 0, s>>>(p);
    return WM_OK;
}

int wm_gemm(wm_ctx *ctx, const GemmArgs &g) {
    WM_REQUIRE(g.K % BK == 0 && g.K >= BK, WM_ERR_INVALID, "gemm: K=%d must be a multiple of %d", g.K, BK);
    WM_REQUIRE(g.M > 0 && g.N > 0 && g.M < (1 << 30) && g.a_rpb > 0 && g.c_rpb > 0, WM_ERR_INVALID, "gemm: bad problem size");
    GemmDev p;
    p.A = g.A; p.a_rpb = g.a_rpb; p.a_bstride = g.a_bstride; p.a_rstride = g.a_rstride;
    p.W = g.W; p.bias = g.bias; p.C = g.C;
    p.c_rpb = g.c_rpb; p.c_bstride = g.c_bstride; p.c_rstride = g.c_rstride;
    p.M = g.M; p.N = g.N; p.K = g.K; p.pos = g.pos; p.vt = g.vt;
    p.d_model = g.d_model; p.n_head = g.n_head; p.seq = g.seq; p.seq_pad = g.seq_pad; p.batch = g.batch;
    // Tile choice: the 256 x 256 staggered-phase kernel once it can put a workgroup on most CUs (one per CU); the 128 x 128
    // kernel (two per CU) below that; the 64 x 64 kernel when even 128 x 128 tiles leave most of the chip idle (fewer than 128
    // tiles: one 30 s chunk at N = d -- small: 72 -> 288 workgroups; the 128 tile keeps products whose 128-grid fills the chip).
    // (g_wm_tuning.gemm_tile: the debug library's tests / probes.)
    const int env_tile = g_wm_tuning.gemm_tile;
    const long tiles256 = (long)((g.M + BM2 - 1) / BM2) * ((g.N + BN2 - 1) / BN2);
    const long tiles128 = (long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    // (128 .. 159 tiles of 128 x 128 run on the pipelined 128 kernel, one per CU: Whisper-small's cross-K/V products at one chunk,
    // 144 tiles, 0.679 -> 0.632 ms per language-identification call against 576 tiles of 64 x 64; 72 tiles: the 64 tile wins)
    int tile = (g.K >= 2 * BK && tiles256 >= 160) ? 256 : (tiles128 >= 128 ? 128 : 64);
    if (env_tile == 64 || env_tile == 128) tile = env_tile;
    if (env_tile == 256 && g.K >= 2 * BK) tile = 256;
    const int bm = tile == 256 ? BM2 : tile == 128 ? BM : BM3, bn = tile == 256 ? BN2 : tile == 128 ? BN : BN3;
    p.tiles_m = (g.M + bm - 1) / bm;
    p.tiles_n = (g.N + bn - 1) / bn;
    const int grid = p.tiles_m * p.tiles_n;
    // 128 tile: the 3-stage pipeline (96 KiB of LDS: ONE workgroup per CU) while the grid fits one round of the chip that way,
    // the two-per-CU double buffer above that (measured, encoder per call: large-v2 x 2 chunks 7.64 -> 6.51 ms with its 240-tile
    // N = d products on the pipeline; a 288- or 360-tile product is two rounds at one per CU and loses 3-6 %:
    // profiles/r06_gemm128_pipeline_ab.txt).  (g_wm_tuning.gemm128_pipe: probes; 1 = never, 2 = always)
    const bool pipe128 = g_wm_tuning.gemm128_pipe == 2 || (g_wm_tuning.gemm128_pipe == 0 && grid <= ctx->n_cus);
    p.group_m = g_wm_tuning.gemm_gm > 0 ? g_wm_tuning.gemm_gm : 4;  // measured at large-v2, B = 8: GM 1 / 4 / 8 / 16 -> fc1 341 / 328 / 335 / 335 us
    static const char *names[] = {"gemm_bias_bf16", "gemm_gelu_bf16", "gemm_resid_f32", "gemm_conv2_f32",
                                  "gemm_qkv_enc", "gemm_xkv", "gemm_f32"};
    WmProfScope ps(&ctx->prof, names[g.epi], ctx->stream);
    hipStream_t st = ctx->stream;
    switch (g.epi) {
        case EPI_BIAS_BF16: WM_TRY(launch_gemm<EPI_BIAS_BF16>(p, tile, pipe128, grid, st)); break;
        case EPI_GELU_BF16: WM_TRY(launch_gemm<EPI_GELU_BF16>(p, tile, pipe128, grid, st)); break;
        case EPI_RESID_F32: WM_TRY(launch_gemm<EPI_RESID_F32>(p, tile, pipe128, grid, st)); break;
        case EPI_CONV2_F32: WM_TRY(launch_gemm<EPI_CONV2_F32>(p, tile, pipe128, grid, st)); break;
        case EPI_QKV_ENC: WM_TRY(launch_gemm<EPI_QKV_ENC>(p, tile, pipe128, grid, st)); break;
        case EPI_XKV: WM_TRY(launch_gemm<EPI_XKV>(p, tile, pipe128, grid, st)); break;
        case EPI_F32: WM_TRY(launch_gemm<EPI_F32>(p, tile, pipe128, grid, st)); break;
        default: wm_set_error("gemm: bad epilogue %d", g.epi); return WM_ERR_INVALID;
    }
    WM_HIP(hipGetLastError());
    return WM_OK;
}
