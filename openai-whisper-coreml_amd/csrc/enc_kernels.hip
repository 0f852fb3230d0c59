// enc_kernels.hip -- encoder-side kernels other than the GEMM (SURVEY.md section 2:
// K6 LayerNorm, K8 encoder flash attention, plus the mel re-layout feeding conv1).
#include <type_traits>

#include "model.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) float f32x8;

__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
}

// ------------------------------------------------------------------ LayerNorm ----------
// One wave64 per row, f32 statistics (eps 1e-5), two passes over registers.  HBM-bound:
// reads d*4 B, writes d*2 B (bf16 GEMM operand) and/or d*4 B.
template <int MAXP>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x,
                                                        const float *__restrict__ g,
                                                        const float *__restrict__ b, int rows, int d,
                                                        bf16_t *__restrict__ ob, float *__restrict__ of) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    // 16 bytes per lane per load (one 1 KiB request per wave-instruction), all MAXP loads of the row in flight at once
    const float4 *xr = (const float4 *)(x + (size_t)row * d);
    const int np = d >> 2;  // float4 quads per row (d % 4 == 0, checked by the launcher)
    float4 v[MAXP];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int idx = i * 64 + lane;
        v[i] = (idx < np) ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int idx = i * 64 + lane;
        if (idx < np) {
            const float a = v[i].x - mean, c = v[i].y - mean, e = v[i].z - mean, f = v[i].w - mean;
            q += (a * a + c * c) + (e * e + f * f);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)d + 1e-5f);
    const float4 *g4 = (const float4 *)g, *b4 = (const float4 *)b;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int idx = i * 64 + lane;
        if (idx < np) {
            const float4 gg = g4[idx], bb = b4[idx];
            const float y0 = (v[i].x - mean) * rstd * gg.x + bb.x;
            const float y1 = (v[i].y - mean) * rstd * gg.y + bb.y;
            const float y2 = (v[i].z - mean) * rstd * gg.z + bb.z;
            const float y3 = (v[i].w - mean) * rstd * gg.w + bb.w;
            if (ob) ((uint2 *)(ob + (size_t)row * d))[idx] = make_uint2(pack2(y0, y1), pack2(y2, y3));
            if (of) ((float4 *)(of + (size_t)row * d))[idx] = make_float4(y0, y1, y2, y3);
        }
    }
}

// ------------------------------------------------------------------ mel re-layout -------
// mel f32 [B][C][3000] (the reference's encoder input layout, Whisper.swift:25) ->
// bf16 [B][3002][C] time-major with zero rows 0 and 3001 (zeroed once at allocation), so
// conv1's 3-tap window of frame t is the contiguous run mel_t[b][t .. t+2][:].
__global__ __launch_bounds__(256) void mel_time_major_kernel(const float *__restrict__ mel, int C,
                                                             bf16_t *__restrict__ out) {
    __shared__ float tile[128][65];
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int c = i >> 6, t = i & 63;
        tile[c][t] = (t0 + t < WM_N_FRAMES) ? mel[((size_t)b * C + c) * WM_N_FRAMES + t0 + t] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int t = i / C, c = i % C;
        if (t0 + t < WM_N_FRAMES) out[((size_t)b * 3002 + 1 + t0 + t) * C + c] = f2bf(tile[c][t]);
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ in,
                                                          bf16_t *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = f2bf(in[i]);
}

// ------------------------------------------------------------------ encoder attention ---
// Non-causal multi-head attention, head_dim 64, flash style (online softmax, fp32
// statistics, scores never leave registers).  One workgroup = 4 waves = 128 query rows of
// one (chunk, head); each wave owns 32 query rows.
//
// CDNA4-specific structure: the score tile is computed TRANSPOSED, S^T = K Q^T, with
// v_mfma_f32_32x32x16_bf16, so that lane l holds scores of ONE query (column l & 31):
// the softmax row reductions are 31 in-register max/add plus a single exchange with lane
// l ^ 32 instead of butterfly shuffles.  P^T then feeds the second product directly as the
// B operand of O^T = V^T P^T (no transpose, no LDS round trip for P); V is stored
// transposed in HBM by the QKV GEMM epilogue so the A operand rows are contiguous, and the
// kv order inside a k-step is whatever order the accumulator registers hold (both operands
// use the same order, so the sum is unchanged).
constexpr int KS_STRIDE = 144;  // bytes per K row in LDS (128 + 16 pad): conflict-free b128
constexpr int VS_STRIDE = 136;  // bytes per V^T row in LDS (128 + 8 pad): conflict-free b64

__global__ __launch_bounds__(256, 4) void enc_attn_kernel(const bf16_t *__restrict__ qk,
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
    const float c = 0.125f * 1.44269504088896340736f;  // hd^-0.5 * log2(e), hd = 64

    // Q^T B-fragments: lane (col q, k = hf*8 + 16*ks .. +8)
    bf16x8 qf[4];
    {
        const bf16_t *qp = qk + ((long)b * S + qc) * ld + h * 64 + hf * 8;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *(const bf16x8 *)(qp + 16 * s4);
    }

    // tile staging: 512 16-byte chunks per operand; thread handles chunks tid and tid+256.  Addresses are a wave-uniform
    // base (SGPRs) + a 32-bit per-thread byte offset (a chunk's K rows span < 8 MB): four 64-bit per-thread pointers and
    // their 64-bit increments per tile cost 8-10 VGPRs at the 128-VGPR cap (the kernel used to spill 4).
    const char *kbase = (const char *)(qk + (long)b * S * ld + d + h * 64);
    const char *vbase = (const char *)(vt + (long)(b * H + h) * 64 * S_pad);
    const int row0 = tid >> 3, c8 = tid & 7, row1 = row0 + 32;
    const unsigned kstep = (unsigned)(64 * ld * 2);            // bytes between consecutive 64-key tiles of K
    const unsigned ko0 = (unsigned)(row0 * ld * 2 + c8 * 16), ko1 = (unsigned)(row1 * ld * 2 + c8 * 16);
    const unsigned vo0 = (unsigned)(row0 * S_pad * 2 + c8 * 16), vo1 = (unsigned)(row1 * S_pad * 2 + c8 * 16);
    uint4 kr0, kr1, vr0, vr1;
#define ATT_GLOAD(j)                                                        \
    do {                                                                    \
        const unsigned kj_ = (unsigned)(j) * kstep, vj_ = (unsigned)(j) * 128u; \
        kr0 = *(const uint4 *)(kbase + (ko0 + kj_));                        \
        kr1 = *(const uint4 *)(kbase + (ko1 + kj_));                        \
        vr0 = *(const uint4 *)(vbase + (vo0 + vj_));                        \
        vr1 = *(const uint4 *)(vbase + (vo1 + vj_));                        \
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

    f32x16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        oacc[0][r] = 0.f;
        oacc[1][r] = 0.f;
    }
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles = (S + 63) / 64;
    ATT_GLOAD(0);
    ATT_LSTORE(0);
    __syncthreads();
    // One 64-key tile.  The softmax, not the MFMA, bounds this kernel: per wave and tile 16 MFMAs (512 cycles) against ~33
    // v_exp_f32 (measured ~2.2x the cost of an FMA) + ~110 other VALU operations, and a wave cannot overlap its own MFMA
    // and VALU phases (4 waves per SIMD do).  So the loop carries as little VALU work as it can: the kv >= S mask lives in
    // the last tile's own copy of the body (if-converted into the loop it was 31 selects per tile), the exponent arguments
    // and the row sum use packed operations (half the instructions; v_pk_*_f32 are two-pass on gfx950, so they save issue
    // slots, not VALU time: enc_attention 134.6 -> 130.5 us at 8 chunks).
    auto tile = [&](int j, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int buf = j & 1;
        if (!LAST) ATT_GLOAD(j + 1);
        // ---- S^T = K Q^T : two 32-kv blocks ---------------------------------------------
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
            const char *kp = ks[buf] + (kb * 32 + ql) * KS_STRIDE + hf * 16;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const bf16x8 kf = *(const bf16x8 *)(kp + s4 * 32);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s4], st[kb], 0, 0, 0);
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
        // ---- online softmax for query column ql ------------------------------------------
        float mloc = st[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kb][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = -m_new * c;
        const f32x2 c2 = {c, c}, mc2 = {mc, mc};
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 sv = {st[kb][r], st[kb][r + 1]};
                const f32x2 ev = __builtin_elementwise_fma(sv, c2, mc2);  // v_pk_fma_f32: one FMA per score pair
                const f32x2 pv = {__builtin_amdgcn_exp2f(ev[0]), __builtin_amdgcn_exp2f(ev[1])};
                st[kb][r] = pv[0];
                st[kb][r + 1] = pv[1];
                ls2 += pv;  // v_pk_add_f32
            }
        float lsum = ls2[0] + ls2[1];
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {  // wave-uniform: the running maxima settle after a few tiles
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                oacc[0][r] *= alpha;
                oacc[1][r] *= alpha;
            }
        }
        // ---- O^T += V^T P^T --------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                f32x8 p8;
#pragma unroll
                for (int i = 0; i < 8; ++i) p8[i] = st[kb][8 * j2 + i];
                const bf16x8 pf = __builtin_convertvector(p8, bf16x8);  // 4 x v_cvt_pk_bf16_f32
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
                    oacc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[eb], 0, 0, 0);
                }
            }
        if (!LAST) ATT_LSTORE(buf ^ 1);
        __syncthreads();
    };
    for (int j = 0; j < ntiles - 1; ++j) tile(j, std::false_type{});
    tile(ntiles - 1, std::true_type{});
    // ---- finalize: O^T[e][q] / l -> att[b*S + q][h*64 + e] -------------------------------
    if (q < S) {
        const float inv = 1.0f / l_run;
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

}  // namespace

int wm_layernorm(wm_ctx *ctx, const float *x, const float *g, const float *b, int rows, int d,
                 bf16_t *out_bf16, float *out_f32) {
    WM_REQUIRE(d % 4 == 0 && d <= 1280, WM_ERR_INVALID, "layernorm: d=%d unsupported (a multiple of 4, <= 1280)", d);
    WmProfScope ps(&ctx->prof, "layernorm", ctx->stream);
    const int grid = (rows + 3) / 4;
    if (d <= 256)
        layernorm_kernel<1><<<grid, 256, 0, ctx->stream>>>(x, g, b, rows, d, out_bf16, out_f32);
    else if (d <= 768)
        layernorm_kernel<3><<<grid, 256, 0, ctx->stream>>>(x, g, b, rows, d, out_bf16, out_f32);
    else
        layernorm_kernel<5><<<grid, 256, 0, ctx->stream>>>(x, g, b, rows, d, out_bf16, out_f32);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int wm_mel_to_time_major(wm_ctx *ctx, const float *mel, int B, int n_mels, bf16_t *mel_t) {
    WM_REQUIRE(n_mels <= 128, WM_ERR_INVALID, "n_mels > 128");
    WmProfScope ps(&ctx->prof, "mel_time_major", ctx->stream);
    dim3 grid((WM_N_FRAMES + 63) / 64, B);
    mel_time_major_kernel<<<grid, 256, 0, ctx->stream>>>(mel, n_mels, mel_t);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int wm_f32_to_bf16(wm_ctx *ctx, const float *in, bf16_t *out, size_t n) {
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    f32_to_bf16_kernel<<<grid, 256, 0, ctx->stream>>>(in, out, n);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int wm_enc_attention(wm_ctx *ctx, const bf16_t *qk, const bf16_t *vt, bf16_t *att, int B, int H,
                     int S, int S_pad, int d) {
    WM_REQUIRE(d == H * 64, WM_ERR_INVALID, "attention: head_dim must be 64 (d=%d, H=%d)", d, H);
    WM_REQUIRE(S_pad % 64 == 0 && S_pad >= S, WM_ERR_INVALID, "attention: bad S_pad");
    WmProfScope ps(&ctx->prof, "enc_attention", ctx->stream);
    const int n_q = (S + 127) / 128, n_bh = B * H;
    const int grid = (n_bh + 7) / 8 * 8 * n_q;
    enc_attn_kernel<<<grid, 256, 0, ctx->stream>>>(qk, vt, att, H, S, S_pad, d, n_q, n_bh);
    WM_HIP(hipGetLastError());
    return WM_OK;
}
