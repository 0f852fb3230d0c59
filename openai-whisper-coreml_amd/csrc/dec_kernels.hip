// dec_kernels.hip -- the autoregressive decode step for gfx950 (SURVEY.md section 2 rows
// K10-K14): batch-of-B single-token decoder pass with a KV cache.
//
// Regime: B <= 16 sequences, so every matrix product is a "skinny" GEMM that streams each
// weight exactly once -- HBM-bound.  Design for CDNA4:
//   * dec_gemv: one workgroup per 16 output features; its NW waves split K; every lane
//     issues all of its 16-byte weight loads (up to 10 per lane = 10 KiB per wave in
//     flight) BEFORE touching the activations, straight into VGPRs (no LDS round trip for
//     data that is used once).  The product itself runs on the matrix pipe:
//     v_mfma_f32_16x16x32_bf16 with the batch padded to 16 rows -- the weight fragment a
//     lane loaded (8 consecutive k of one output row) IS the B operand, no shuffle.
//     LayerNorm, the attention-partial combine, bias, GELU, residual add, KV-cache append
//     and the logits arg-max are fused into the prologue / epilogue, so a decoder layer is
//     8 launches.
//   * dec_attention: single-query attention over the bf16 K/V cache, 8 lanes per 128-byte
//     row (coalesced), fp32 softmax, flash-decoding split over the 1500 encoder frames so
//     all 256 CUs stream; partial (m, l, o) triples are combined by the consumer GEMV.
#include "model.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
}
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ unsigned long long argmax_key(float v, int n) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
}

struct DecGemvDev {
    int B, N, K, KC;
    const bf16_t *W;
    const float *bias;
    const float *x, *ln_g, *ln_b;
    const bf16_t *a_bf16;
    const float *part;
    int nsplit;
    float *out_f32;
    bf16_t *out_bf16;
    bf16_t *kcache, *vcache;
    int pos, n_ctx, n_head;
    long ldo;
    unsigned long long *tilemax;  // [B][n_tiles] (DE_LOGITS)
    int n_tiles;
    int arg_first, arg_last;
};


// One group of G k-steps: issue all G weight loads (16 B per lane each, non-temporal: every
// weight byte is used once per step), then build the A fragments and run the MFMAs.
template <int AMODE, int G>
__device__ __forceinline__ void gemv_group(const DecGemvDev &p, const bf16_t *wp, int s0, int kbase, int kq,
                                           int nrow, bool live, float mean, float rstd, f32x4 &acc) {
    u32x4 wf[G];
#pragma unroll
    for (int u = 0; u < G; ++u) wf[u] = __builtin_nontemporal_load((const u32x4 *)(wp + (s0 + u) * 32));
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const int k = kbase + (s0 + u) * 32 + kq * 8;
        uint4 av = make_uint4(0, 0, 0, 0);
        if (live) {
            if (AMODE == DA_BF16) {
                av = *(const uint4 *)(p.a_bf16 + (long)nrow * p.K + k);
            } else if (AMODE == DA_LN) {
                const float4 x0 = *(const float4 *)(p.x + (long)nrow * p.K + k);
                const float4 x1 = *(const float4 *)(p.x + (long)nrow * p.K + k + 4);
                const float4 g0 = *(const float4 *)(p.ln_g + k), g1 = *(const float4 *)(p.ln_g + k + 4);
                const float4 b0 = *(const float4 *)(p.ln_b + k), b1 = *(const float4 *)(p.ln_b + k + 4);
                av.x = pack2((x0.x - mean) * rstd * g0.x + b0.x, (x0.y - mean) * rstd * g0.y + b0.y);
                av.y = pack2((x0.z - mean) * rstd * g0.z + b0.z, (x0.w - mean) * rstd * g0.w + b0.w);
                av.z = pack2((x1.x - mean) * rstd * g1.x + b1.x, (x1.y - mean) * rstd * g1.y + b1.y);
                av.w = pack2((x1.z - mean) * rstd * g1.z + b1.z, (x1.w - mean) * rstd * g1.w + b1.w);
            } else {  // DA_ATTN: combine flash-decoding partials (m, l, o[64]) of head k/64
                const int h = k >> 6, e = k & 63;
                const float *pp = p.part + ((long)(nrow * p.n_head + h) * p.nsplit) * 66;
                float M = -1e30f;
                for (int s = 0; s < p.nsplit; ++s) M = fmaxf(M, pp[s * 66]);
                float den = 0.f, num[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int s = 0; s < p.nsplit; ++s) {
                    const float w = __expf(pp[s * 66] - M);
                    den += w * pp[s * 66 + 1];
#pragma unroll
                    for (int i = 0; i < 8; ++i) num[i] += w * pp[s * 66 + 2 + e + i];
                }
                const float inv = 1.0f / den;
                av.x = pack2(num[0] * inv, num[1] * inv);
                av.y = pack2(num[2] * inv, num[3] * inv);
                av.z = pack2(num[4] * inv, num[5] * inv);
                av.w = pack2(num[6] * inv, num[7] * inv);
            }
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av),
                                                      __builtin_bit_cast(bf16x8, wf[u]), acc, 0, 0, 0);
    }
}

template <int AMODE, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void dec_gemv_kernel(DecGemvDev p) {
    __shared__ float stats[WM_DEC_MAXB][2];
    __shared__ float red[NW][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrow = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kbase = wave * p.KC;
    const int nsteps = p.KC >> 5;
    const bf16_t *wp = p.W + (long)(n0 + nrow) * p.K + kbase + kq * 8;

    // ---- LayerNorm statistics (fp32, eps 1e-5) for the B live rows ------------------------
    if (AMODE == DA_LN) {
        for (int b = wave; b < p.B; b += NW) {
            const float *xr = p.x + (long)b * p.K;
            float s = 0.f;
            for (int k = lane * 4; k < p.K; k += 256) {
                const float4 v = *(const float4 *)(xr + k);
                s += (v.x + v.y) + (v.z + v.w);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s / (float)p.K;
            float q = 0.f;
            for (int k = lane * 4; k < p.K; k += 256) {
                const float4 v = *(const float4 *)(xr + k);
                const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if (lane == 0) {
                stats[b][0] = mean;
                stats[b][1] = rsqrtf(q / (float)p.K + 1e-5f);
            }
        }
        __syncthreads();
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool live = nrow < p.B;  // A row = batch index
    float mean = 0.f, rstd = 0.f;
    if (AMODE == DA_LN && live) {
        mean = stats[nrow][0];
        rstd = stats[nrow][1];
    }

    int s0 = 0;
    if (nsteps % 10 == 0)
        for (; s0 + 10 <= nsteps; s0 += 10) gemv_group<AMODE, 10>(p, wp, s0, kbase, kq, nrow, live, mean, rstd, acc);
    for (; s0 + 4 <= nsteps; s0 += 4) gemv_group<AMODE, 4>(p, wp, s0, kbase, kq, nrow, live, mean, rstd, acc);
    for (; s0 < nsteps; ++s0) gemv_group<AMODE, 1>(p, wp, s0, kbase, kq, nrow, live, mean, rstd, acc);

    // ---- cross-wave (split-K) reduction through LDS ----------------------------------------
    if (NW > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < NW; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += red[w][lane][r];
    }

    // ---- epilogue (wave 0): D col n = lane & 15, rows b = kq*4 + r --------------------------
    const int n = n0 + nrow;
    const bool nvalid = n < p.N;
    const float bv = (nvalid && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = kq * 4 + r;
        const float v = acc[r] + bv;
        if (EPI == DE_LOGITS) {
            // arg-max over [arg_first, arg_last], first maximal index wins (Whisper.swift:38)
            unsigned long long key = 0ull;
            if (b < p.B && nvalid && n >= p.arg_first && n <= p.arg_last) key = argmax_key(v, n);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const unsigned long long ok = __shfl_xor(key, o);
                key = ok > key ? ok : key;
            }
            if (b < p.B && nrow == 0) p.tilemax[(long)b * p.n_tiles + blockIdx.x] = key;
            if (b < p.B && nvalid && p.out_f32) p.out_f32[(long)b * p.ldo + n] = v;
            continue;
        }
        if (b >= p.B || !nvalid) continue;
        if (EPI == DE_QKV) {
            const int d = p.N / 3;
            if (n < d) {
                p.out_f32[(long)b * d + n] = v;
            } else {
                const int hn = (n < 2 * d) ? n - d : n - 2 * d;
                bf16_t *c = (n < 2 * d) ? p.kcache : p.vcache;
                c[((long)(b * p.n_head + (hn >> 6)) * p.n_ctx + p.pos) * 64 + (hn & 63)] = f2bf(v);
            }
        } else if (EPI == DE_Q) {
            p.out_f32[(long)b * p.ldo + n] = v;
        } else if (EPI == DE_RESID) {
            p.out_f32[(long)b * p.ldo + n] += v;
        } else if (EPI == DE_GELU) {
            p.out_bf16[(long)b * p.ldo + n] = f2bf(gelu_erf(v));
        }
    }
}

// ------------------------------------------------------------------ token embedding ------
__global__ __launch_bounds__(256) void dec_embed_kernel(const int *__restrict__ tokens, int pos,
                                                        const bf16_t *__restrict__ emb,
                                                        const float *__restrict__ pemb, int d,
                                                        float *__restrict__ x) {
    const int b = blockIdx.x;
    const long tok = tokens[b];
    for (int j = threadIdx.x; j < d; j += 256)
        x[(long)b * d + j] = bf2f(emb[tok * d + j]) + pemb[(long)pos * d + j];
}

// ------------------------------------------------------------------ single-query attention
// grid (B*H, nsplit); 256 threads.  Keys [start, end) of this split; 8 lanes share a
// 128-byte K/V row (16 B each), 8 rows per wave-load.
__global__ __launch_bounds__(256) void dec_attn_kernel(const float *__restrict__ q,
                                                       const bf16_t *__restrict__ kc,
                                                       const bf16_t *__restrict__ vc, int H, int d,
                                                       int T_stride, int n_keys, int nsplit,
                                                       float *__restrict__ part) {
    __shared__ float sc[512];
    __shared__ float wred[4];
    __shared__ float wacc[4][64];
    const int bh = blockIdx.x, b = bh / H, h = bh % H, sp = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = lane >> 3, e8 = lane & 7;
    int chunk = (n_keys + nsplit - 1) / nsplit;
    chunk = (chunk + 7) & ~7;
    const int start = sp * chunk;
    int end = start + chunk;
    if (end > n_keys) end = n_keys;
    const int cnt = end > start ? end - start : 0;
    float *po = part + ((long)bh * nsplit + sp) * 66;

    float qe[8];
    {
        const float *qp = q + (long)b * d + h * 64 + e8 * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) qe[i] = qp[i] * 0.125f;  // hd^-0.5 (== hd^-0.25 on q and on k)
    }
    const bf16_t *kb = kc + (long)bh * T_stride * 64 + e8 * 8;
    const bf16_t *vb = vc + (long)bh * T_stride * 64 + e8 * 8;

    // ---- scores ---------------------------------------------------------------------------
    float mloc = -1e30f;
    for (int i0 = wave * 8; i0 < cnt; i0 += 32) {
        const int i = i0 + rg;
        float s = -1e30f;
        if (i < cnt) {
            const uint4 kv = *(const uint4 *)(kb + (long)(start + i) * 64);
            const unsigned w[4] = {kv.x, kv.y, kv.z, kv.w};
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a += qe[2 * j] * __uint_as_float(w[j] << 16);
                a += qe[2 * j + 1] * __uint_as_float(w[j] & 0xffff0000u);
            }
            s = a;
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (i < cnt) {
            if (e8 == 0) sc[i] = s;
            mloc = fmaxf(mloc, s);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, o));
    if (lane == 0) wred[wave] = mloc;
    __syncthreads();
    const float M = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    __syncthreads();
    // ---- exp + sum ------------------------------------------------------------------------
    float lloc = 0.f;
    for (int i = tid; i < cnt; i += 256) {
        const float pv = __expf(sc[i] - M);
        sc[i] = pv;
        lloc += pv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lloc += __shfl_xor(lloc, o);
    if (lane == 0) wred[wave] = lloc;
    __syncthreads();
    const float L = (wred[0] + wred[1]) + (wred[2] + wred[3]);
    // ---- o = sum_i p_i V[i] -----------------------------------------------------------------
    float oa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i0 = wave * 8; i0 < cnt; i0 += 32) {
        const int i = i0 + rg;
        if (i < cnt) {
            const float pv = sc[i];
            const uint4 vv = *(const uint4 *)(vb + (long)(start + i) * 64);
            const unsigned w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                oa[2 * j] += pv * __uint_as_float(w[j] << 16);
                oa[2 * j + 1] += pv * __uint_as_float(w[j] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        oa[i] += __shfl_xor(oa[i], 8);
        oa[i] += __shfl_xor(oa[i], 16);
        oa[i] += __shfl_xor(oa[i], 32);
    }
    if (rg == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) wacc[wave][e8 * 8 + i] = oa[i];
    }
    __syncthreads();
    if (tid < 64) po[2 + tid] = (wacc[0][tid] + wacc[1][tid]) + (wacc[2][tid] + wacc[3][tid]);
    if (tid == 0) {
        po[0] = cnt > 0 ? M : -1e30f;
        po[1] = cnt > 0 ? L : 0.f;
    }
}

// ------------------------------------------------------------------ arg-max -> next token
__global__ __launch_bounds__(256) void argmax_tokens_kernel(const unsigned long long *__restrict__ tilemax,
                                                            int n_tiles, int *__restrict__ cur,
                                                            int *__restrict__ history, int hist_stride,
                                                            int hist_pos, int *__restrict__ result,
                                                            int arg_first) {
    __shared__ unsigned long long wk[4];
    const int b = blockIdx.x;
    unsigned long long key = 0ull;
    for (int t = threadIdx.x; t < n_tiles; t += 256) {
        const unsigned long long k = tilemax[(long)b * n_tiles + t];
        key = k > key ? k : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(key, o);
        key = ok > key ? ok : key;
    }
    if ((threadIdx.x & 63) == 0) wk[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) key = wk[w] > key ? wk[w] : key;
        const int tok = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        if (cur) cur[b] = tok;
        if (history) history[(long)b * hist_stride + hist_pos] = tok;
        if (result) result[b] = tok - arg_first;
    }
}

// ------------------------------------------------------------------ synthetic weights ----
// Mirrors weights.synthetic_values(): Irwin-Hall(4) of 16-bit hash words, exact integer sum,
// one f32 multiply, optional bf16 rounding -> bit-identical to the numpy generator.
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void synth_fill_kernel(void *dst, int is_bf16, size_t n, unsigned key,
                                                         float scale, int layout, int conv_c, int kpad,
                                                         int kind) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v;
        if (kind == 2) {
            v = 1.0f;
        } else if (kind == 3) {
            v = 0.0f;
        } else {
            const unsigned h1 = hash32((unsigned)i ^ key);
            const unsigned h2 = hash32(h1 + 0x85ebca6bu);
            const int s = (int)(h1 & 0xffffu) + (int)(h1 >> 16) + (int)(h2 & 0xffffu) + (int)(h2 >> 16);
            v = __fmul_rn((float)(s - 131070), scale);
        }
        size_t o = i;
        if (layout == WL_CONV) {
            const size_t per = (size_t)conv_c * 3;
            const size_t oc = i / per, rem = i % per, c = rem / 3, tap = rem % 3;
            o = oc * kpad + tap * conv_c + c;
        }
        if (is_bf16)
            ((bf16_t *)dst)[o] = f2bf(v);
        else
            ((float *)dst)[o] = (kind == 0) ? bf2f(f2bf(v)) : v;  // matrices are bf16-representable everywhere
    }
}

template <int AMODE, int EPI>
int launch_gemv(wm_ctx *ctx, const DecGemvDev &p, int nw, int grid) {
    hipStream_t s = ctx->stream;
    switch (nw) {
        case 1: dec_gemv_kernel<AMODE, EPI, 1><<<grid, 64, 0, s>>>(p); break;
        case 2: dec_gemv_kernel<AMODE, EPI, 2><<<grid, 128, 0, s>>>(p); break;
        case 4: dec_gemv_kernel<AMODE, EPI, 4><<<grid, 256, 0, s>>>(p); break;
        case 8: dec_gemv_kernel<AMODE, EPI, 8><<<grid, 512, 0, s>>>(p); break;
        case 16: dec_gemv_kernel<AMODE, EPI, 16><<<grid, 1024, 0, s>>>(p); break;
        default: wm_set_error("dec_gemv: bad wave count %d", nw); return WM_ERR_INVALID;
    }
    WM_HIP(hipGetLastError());
    return WM_OK;
}

}  // namespace

static int pick_waves(int K) {
    int nw = 16;
    while (nw > 1 && (K / 256 < nw || K % (32 * nw) != 0)) nw >>= 1;
    return nw;
}

int wm_dec_gemv(wm_ctx *ctx, const DecGemvArgs &a) {
    WM_REQUIRE(a.B >= 1 && a.B <= WM_DEC_MAXB, WM_ERR_INVALID, "dec_gemv: B=%d out of range", a.B);
    WM_REQUIRE(a.K % 32 == 0, WM_ERR_INVALID, "dec_gemv: K=%d must be a multiple of 32", a.K);
    const int nw = pick_waves(a.K);
    DecGemvDev p;
    p.B = a.B; p.N = a.N; p.K = a.K; p.KC = a.K / nw;
    p.W = a.W; p.bias = a.bias; p.x = a.x; p.ln_g = a.ln_g; p.ln_b = a.ln_b; p.a_bf16 = a.a_bf16;
    p.part = a.part; p.nsplit = a.nsplit; p.out_f32 = a.out_f32; p.out_bf16 = a.out_bf16;
    p.kcache = a.kcache; p.vcache = a.vcache; p.pos = a.pos; p.n_ctx = a.n_ctx; p.n_head = a.n_head;
    p.ldo = a.ldo; p.tilemax = a.argmax; p.arg_first = a.arg_first; p.arg_last = a.arg_last;
    const int grid = (a.N + 15) / 16;
    p.n_tiles = grid;
    const int key = a.a_mode * 8 + a.epi;
    switch (key) {
        case DA_LN * 8 + DE_QKV: {
            WmProfScope ps(&ctx->prof, "dec_gemv_ln_qkv", ctx->stream);
            return launch_gemv<DA_LN, DE_QKV>(ctx, p, nw, grid);
        }
        case DA_LN * 8 + DE_Q: {
            WmProfScope ps(&ctx->prof, "dec_gemv_ln_q", ctx->stream);
            return launch_gemv<DA_LN, DE_Q>(ctx, p, nw, grid);
        }
        case DA_LN * 8 + DE_GELU: {
            WmProfScope ps(&ctx->prof, "dec_gemv_ln_fc1", ctx->stream);
            return launch_gemv<DA_LN, DE_GELU>(ctx, p, nw, grid);
        }
        case DA_LN * 8 + DE_LOGITS: {
            WmProfScope ps(&ctx->prof, "dec_gemv_ln_logits", ctx->stream);
            return launch_gemv<DA_LN, DE_LOGITS>(ctx, p, nw, grid);
        }
        case DA_ATTN * 8 + DE_RESID: {
            WmProfScope ps(&ctx->prof, "dec_gemv_attn_out", ctx->stream);
            return launch_gemv<DA_ATTN, DE_RESID>(ctx, p, nw, grid);
        }
        case DA_BF16 * 8 + DE_RESID: {
            WmProfScope ps(&ctx->prof, "dec_gemv_fc2", ctx->stream);
            return launch_gemv<DA_BF16, DE_RESID>(ctx, p, nw, grid);
        }
        case DA_BF16 * 8 + DE_Q: {
            WmProfScope ps(&ctx->prof, "dec_gemv_plain", ctx->stream);
            return launch_gemv<DA_BF16, DE_Q>(ctx, p, nw, grid);
        }
        default:
            wm_set_error("dec_gemv: unsupported mode pair (%d, %d)", a.a_mode, a.epi);
            return WM_ERR_INVALID;
    }
}

int wm_dec_embed(wm_ctx *ctx, const int *tokens, int B, int pos, const bf16_t *emb, const float *pemb,
                 int d, float *x, unsigned long long *) {
    WmProfScope ps(&ctx->prof, "dec_embed", ctx->stream);
    dec_embed_kernel<<<B, 256, 0, ctx->stream>>>(tokens, pos, emb, pemb, d, x);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int wm_dec_attention(wm_ctx *ctx, const float *q, const bf16_t *kc, const bf16_t *vc, int B, int H,
                     int T_stride, int n_keys, int nsplit, float *part) {
    WM_REQUIRE(n_keys >= 1 && (n_keys + nsplit - 1) / nsplit + 8 <= 512, WM_ERR_INVALID,
               "dec_attention: %d keys / %d splits exceeds the 512-key LDS tile", n_keys, nsplit);
    WmProfScope ps(&ctx->prof, nsplit > 1 ? "dec_attn_cross" : "dec_attn_self", ctx->stream);
    dim3 grid(B * H, nsplit);
    dec_attn_kernel<<<grid, 256, 0, ctx->stream>>>(q, kc, vc, H, H * 64, T_stride, n_keys, nsplit, part);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int wm_argmax_reduce(wm_ctx *ctx, const unsigned long long *tilemax, int n_tiles, int B, int *cur,
                     int *history, int hist_stride, int hist_pos, int *result, int arg_first) {
    WmProfScope ps(&ctx->prof, "argmax_reduce", ctx->stream);
    argmax_tokens_kernel<<<B, 256, 0, ctx->stream>>>(tilemax, n_tiles, cur, history, hist_stride, hist_pos,
                                                     result, arg_first);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int wm_fill_synthetic(wm_ctx *ctx, const WmTensor &t, uint32_t seed, int tensor_id) {
    if (t.kind == 4) return WM_OK;  // sinusoids are computed on the host (model.cpp)
    const unsigned key = [&] {
        unsigned x = seed + (unsigned)tensor_id * 0x9E3779B9u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return x;
    }();
    const double std = (t.kind == 0) ? 0.02 : 0.01;
    const float scale = (float)(std / 37837.22659);
    const int grid = (int)((t.n_elems + 255) / 256 < 16384 ? (t.n_elems + 255) / 256 : 16384);
    synth_fill_kernel<<<grid, 256, 0, ctx->stream>>>(t.ptr, t.is_bf16 ? 1 : 0, t.n_elems, key, scale, t.layout,
                                                     t.conv_c, t.conv_kpad, t.kind);
    WM_HIP(hipGetLastError());
    return WM_OK;
}
