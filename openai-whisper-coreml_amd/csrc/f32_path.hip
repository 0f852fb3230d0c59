// f32_path.hip -- the ALL-FP32 debug model path (libwhisper_mi355x_dbg.so only; never linked into the product).
//
// BASELINE.md's parity gate: "fp32 debug path must match to <= 1e-4 rel-L2"; SURVEY.md section 7 hard part 3: "keep an
// all-fp32 debug path to separate bugs from rounding".  wmdbg_set_precision(ctx, WM_F32) routes wm_encode and
// wm_decode_logits of that context through the kernels below: the SAME weights the product multiplies (the bf16 values in
// HBM, in their product layouts: conv taps permuted, decoder matrices fragment-tiled, Q|K|V fused), widened to f32 on
// load, with every activation, the K/V projections, the softmax and the accumulation in f32 -- the graphs
// whisper_to_cml.py:10-43 traces (`load_models` puts the model on the CPU in fp32, :6-8).  What this separates: a
// disagreement with the oracle that survives here is a bug in semantics, layout or indexing; one that disappears is bf16
// rounding of the product path.  Speed is irrelevant (plain FMA tiles, one query row per lane); the decoder is the
// stateless full-prefix form of the exported graph (no KV cache).
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/whisper_mi355x_debug.h"
#include "model.h"

namespace {

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

enum { F_STORE = 0, F_GELU = 1, F_RESID = 2, F_GELU_POS = 3 };

struct F32Gemm {
    const float *A;   // rows addressed as (m / a_rpb) * a_bstride + (m % a_rpb) * a_rstride (elements), K contiguous
    long a_rpb, a_bstride, a_rstride;
    const bf16_t *W;  // [N][K]: plain rows of ldw elements, or the WL_TILED fragment order (wm_tiled_offset)
    int w_tiled;
    long ldw;
    const float *bias;
    float *C;         // same row addressing
    long c_rpb, c_bstride, c_rstride;
    int M, N, K;
    int epi;
    const float *pos;  // F_GELU_POS: [c_rpb][N]
};

// C = epi(A W^T + bias): 64 x 64 tile, 16 x 16 threads x (4 x 4) outputs, K in steps of 16 through LDS, f32 FMA chains
// in ascending k.
__global__ __launch_bounds__(256) void gemm_f32_kernel(F32Gemm p) {
    __shared__ float As[16][68], Ws[16][68];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lr = t >> 2, lk = (t & 3) * 4;   // loader: row lr of the tile, k offset lk .. lk + 3
    int am = m0 + lr;
    am = am < p.M ? am : p.M - 1;
    const float *arow = p.A + (am / p.a_rpb) * p.a_bstride + (am % p.a_rpb) * p.a_rstride;
    int wn = n0 + lr;
    wn = wn < p.N ? wn : p.N - 1;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const float4 av = *(const float4 *)(arow + k0 + lk);
        const bf16_t *wp = p.w_tiled ? p.W + wm_tiled_offset((size_t)wn, (size_t)(k0 + lk), (size_t)p.K)
                                     : p.W + (long)wn * p.ldw + k0 + lk;
        const ushort4 wv = *(const ushort4 *)wp;
        As[lk + 0][lr] = av.x; As[lk + 1][lr] = av.y; As[lk + 2][lr] = av.z; As[lk + 3][lr] = av.w;
        Ws[lk + 0][lr] = bf2f(wv.x); Ws[lk + 1][lr] = bf2f(wv.y); Ws[lk + 2][lr] = bf2f(wv.z); Ws[lk + 3][lr] = bf2f(wv.w);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[k][ty * 4 + i];
                w[i] = Ws[k][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
        float *crow = p.C + (m / p.c_rpb) * p.c_bstride + (m % p.c_rpb) * p.c_rstride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float v = acc[i][j] + (p.bias ? p.bias[n] : 0.f);
            if (p.epi == F_GELU) v = gelu_exact(v);
            else if (p.epi == F_RESID) v = crow[n] + v;
            else if (p.epi == F_GELU_POS) v = gelu_exact(v) + p.pos[(long)(m % p.c_rpb) * p.N + n];
            crow[n] = v;
        }
    }
}

// Multi-head attention, head dim 64, f32 throughout: one query row per lane, keys / values in LDS tiles of 64 rows
// (every lane reads the same K / V element: an LDS broadcast), online softmax over sub-tiles of 16 keys.
// q / k / v / out: row r of batch b at base + (b * rows_per_batch + r) * ld + h * 64.  causal: key j <= query i.
__global__ __launch_bounds__(64) void attn_f32_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ v, float *__restrict__ out, int Tq, int Tk,
                                                      long ldq, long ldk, long ldv, long ldo, int causal) {
    __shared__ float4 ks[64][16], vs[64][16];
    const int lane = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 64 + lane, qc = qi < Tq ? qi : Tq - 1;
    float qr[64], o[64];
    {
        const float4 *qp = (const float4 *)(q + ((long)b * Tq + qc) * ldq + h * 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 x = qp[i];
            qr[4 * i] = x.x * 0.125f; qr[4 * i + 1] = x.y * 0.125f; qr[4 * i + 2] = x.z * 0.125f; qr[4 * i + 3] = x.w * 0.125f;
        }
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) o[i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const int k_end = causal ? min(Tk, blockIdx.x * 64 + 64) : Tk;   // workgroup-uniform: no key beyond the block's last query
    for (int j0 = 0; j0 < k_end; j0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = i * 64 + lane, r = idx >> 4, c = idx & 15;
            const int kr = j0 + r < Tk ? j0 + r : Tk - 1;
            ks[r][c] = *(const float4 *)(k + ((long)b * Tk + kr) * ldk + h * 64 + c * 4);
            vs[r][c] = *(const float4 *)(v + ((long)b * Tk + kr) * ldv + h * 64 + c * 4);
        }
        __syncthreads();
        for (int s0 = 0; s0 < 64; s0 += 16) {
            float s[16];
            float mx = m_run;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float4 kk = ks[s0 + jj][c];
                    a = fmaf(qr[4 * c], kk.x, a); a = fmaf(qr[4 * c + 1], kk.y, a);
                    a = fmaf(qr[4 * c + 2], kk.z, a); a = fmaf(qr[4 * c + 3], kk.w, a);
                }
                const int kj = j0 + s0 + jj;
                const bool ok = kj < Tk && (!causal || kj <= qc);
                s[jj] = ok ? a : -1e30f;
                mx = fmaxf(mx, s[jj]);
            }
            const float alpha = expf(m_run - mx);   // first sub-tile holds key 0, always admissible: mx is finite from then on
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 64; ++i) o[i] *= alpha;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float pj = s[jj] > -1e29f ? expf(s[jj] - mx) : 0.f;
                l_run += pj;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float4 vv = vs[s0 + jj][c];
                    o[4 * c] = fmaf(pj, vv.x, o[4 * c]); o[4 * c + 1] = fmaf(pj, vv.y, o[4 * c + 1]);
                    o[4 * c + 2] = fmaf(pj, vv.z, o[4 * c + 2]); o[4 * c + 3] = fmaf(pj, vv.w, o[4 * c + 3]);
                }
            }
            m_run = mx;
        }
    }
    if (qi < Tq) {
        const float inv = 1.0f / l_run;
        float4 *op = (float4 *)(out + ((long)b * Tq + qi) * ldo + h * 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) op[i] = make_float4(o[4 * i] * inv, o[4 * i + 1] * inv, o[4 * i + 2] * inv, o[4 * i + 3] * inv);
    }
}

// mel f32 [B][C][3000] -> f32 [B][3002][C] time-major (rows 0 and 3001 stay zero: conv padding 1)
__global__ void mel_t_f32_kernel(const float *__restrict__ mel, int C, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t c = i % C, t = (i / C) % 3000, b = i / ((size_t)C * 3000);
        out[(b * 3002 + 1 + t) * C + c] = mel[(b * C + c) * 3000 + t];
    }
}

// x[b*T + t][j] = token_embedding[tok[b][t]][j] + positional_embedding[t][j]
__global__ void embed_f32_kernel(const int *__restrict__ tok, int T, const bf16_t *__restrict__ emb,
                                 const float *__restrict__ pemb, int d, float *__restrict__ x) {
    const int row = blockIdx.x, t = row % T;
    const long token = tok[row];
    for (int j = threadIdx.x; j < d; j += blockDim.x)
        x[(long)row * d + j] = bf2f(emb[wm_tiled_offset((size_t)token, (size_t)j, (size_t)d)]) + pemb[(long)t * d + j];
}

struct Scratch {
    std::vector<void *> bufs;
    ~Scratch() {
        for (void *p : bufs) (void)hipFree(p);
    }
    int get(float **p, size_t n, hipStream_t s) {
        void *q = nullptr;
        WM_HIP(hipMalloc(&q, n * sizeof(float) + 256));
        bufs.push_back(q);
        WM_HIP(hipMemsetAsync(q, 0, n * sizeof(float) + 256, s));
        *p = (float *)q;
        return WM_OK;
    }
};

int gemm(wm_ctx *ctx, const float *A, long lda, const bf16_t *W, bool tiled, long ldw, const float *bias, float *C, long ldc,
         int M, int N, int K, int epi) {
    WM_REQUIRE(K % 16 == 0 && lda % 4 == 0, WM_ERR_INVALID, "f32 path: K %% 16 / lda %% 4");
    F32Gemm p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.a_rpb = (long)M + 1; p.a_rstride = lda;
    p.W = W; p.w_tiled = tiled ? 1 : 0; p.ldw = ldw; p.bias = bias;
    p.C = C; p.c_rpb = (long)M + 1; p.c_rstride = ldc;
    p.M = M; p.N = N; p.K = K; p.epi = epi;
    gemm_f32_kernel<<<dim3((N + 63) / 64, (M + 63) / 64), 256, 0, ctx->stream>>>(p);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int attention(wm_ctx *ctx, const float *q, const float *k, const float *v, float *out, int B, int H, int Tq, int Tk, long ldq,
              long ldk, long ldv, long ldo, bool causal) {
    attn_f32_kernel<<<dim3((Tq + 63) / 64, H, B), 64, 0, ctx->stream>>>(q, k, v, out, Tq, Tk, ldq, ldk, ldv, ldo, causal ? 1 : 0);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

// AudioEncoder.forward in f32 (oracle/whisper_ref.py encode; whisper_to_cml.py:10-23)
int f32_encode(wm_ctx *ctx, const float *d_mel, int B, float *d_xa) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m && m->finalized, WM_ERR_STATE, "model weights not finalised (wm_finalize)");
    WM_REQUIRE(B >= 1, WM_ERR_INVALID, "B must be >= 1");
    const wm_dims &D = m->dims;
    const int d = D.n_audio_state, H = D.n_audio_head, S = 1500, C = D.n_mels, M = B * S;
    hipStream_t s = ctx->stream;
    Scratch sc;
    float *mel_t, *h1, *x, *xn, *qkv, *att, *hid;
    WM_TRY(sc.get(&mel_t, (size_t)B * 3002 * C, s));
    WM_TRY(sc.get(&h1, (size_t)B * 3001 * d, s));
    WM_TRY(sc.get(&x, (size_t)M * d, s));
    WM_TRY(sc.get(&xn, (size_t)M * d, s));
    WM_TRY(sc.get(&qkv, (size_t)M * 3 * d, s));
    WM_TRY(sc.get(&att, (size_t)M * d, s));
    WM_TRY(sc.get(&hid, (size_t)M * 4 * d, s));
    mel_t_f32_kernel<<<4096, 256, 0, s>>>(d_mel, C, mel_t, (size_t)B * C * 3000);
    WM_HIP(hipGetLastError());
    {   // conv1 (k 3, p 1) + GELU: the window of frame t is the contiguous run mel_t[b][t .. t + 2][:]
        F32Gemm p;
        memset(&p, 0, sizeof(p));
        p.A = mel_t; p.a_rpb = 3000; p.a_bstride = 3002L * C; p.a_rstride = C;
        p.W = m->conv1_w; p.ldw = m->k1pad; p.bias = m->conv1_b;
        p.C = h1 + d; p.c_rpb = 3000; p.c_bstride = 3001L * d; p.c_rstride = d;
        p.M = B * 3000; p.N = d; p.K = 3 * C; p.epi = F_GELU;
        WM_REQUIRE(p.K % 16 == 0, WM_ERR_INVALID, "f32 path: 3 * n_mels must be a multiple of 16");
        gemm_f32_kernel<<<dim3((p.N + 63) / 64, (p.M + 63) / 64), 256, 0, s>>>(p);
        WM_HIP(hipGetLastError());
    }
    {   // conv2 (k 3, s 2, p 1) + GELU + positional embedding: the window of frame s is h1[b][2s .. 2s + 2][:]
        F32Gemm p;
        memset(&p, 0, sizeof(p));
        p.A = h1; p.a_rpb = S; p.a_bstride = 3001L * d; p.a_rstride = 2L * d;
        p.W = m->conv2_w; p.ldw = 3L * d; p.bias = m->conv2_b;
        p.C = x; p.c_rpb = S; p.c_bstride = (long)S * d; p.c_rstride = d;
        p.M = M; p.N = d; p.K = 3 * d; p.epi = F_GELU_POS; p.pos = m->enc_pos;
        gemm_f32_kernel<<<dim3((p.N + 63) / 64, (p.M + 63) / 64), 256, 0, s>>>(p);
        WM_HIP(hipGetLastError());
    }
    for (int i = 0; i < D.n_audio_layer; ++i) {
        const EncLayerW &L = m->enc[i];
        WM_TRY(wm_layernorm(ctx, x, L.ln1_g, L.ln1_b, M, d, nullptr, xn));
        WM_TRY(gemm(ctx, xn, d, L.wqkv, false, d, L.bqkv, qkv, 3 * d, M, 3 * d, d, F_STORE));
        WM_TRY(attention(ctx, qkv, qkv + d, qkv + 2 * d, att, B, H, S, S, 3 * d, 3 * d, 3 * d, d, false));
        WM_TRY(gemm(ctx, att, d, L.wo, false, d, L.bo, x, d, M, d, d, F_RESID));
        WM_TRY(wm_layernorm(ctx, x, L.ln2_g, L.ln2_b, M, d, nullptr, xn));
        WM_TRY(gemm(ctx, xn, d, L.w1, false, d, L.b1, hid, 4 * d, M, 4 * d, d, F_GELU));
        WM_TRY(gemm(ctx, hid, 4 * d, L.w2, false, 4 * d, L.b2, x, d, M, d, 4 * d, F_RESID));
    }
    WM_TRY(wm_layernorm(ctx, x, m->ln_post_g, m->ln_post_b, M, d, nullptr, d_xa));
    WM_HIP(hipStreamSynchronize(s));   // the scratch buffers are freed on return
    return WM_OK;
}

// TextDecoder.forward without kv_cache (oracle/whisper_ref.py decode_logits; whisper_to_cml.py:25-43) in f32
int f32_decode_logits(wm_ctx *ctx, const int32_t *host_tokens, int B, int T, const float *d_xa, float *d_logits) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m && m->finalized, WM_ERR_STATE, "model weights not finalised (wm_finalize)");
    const wm_dims &D = m->dims;
    const int d = D.n_text_state, H = D.n_text_head, S = 1500, V = D.n_vocab, M = B * T, MA = B * S;
    hipStream_t s = ctx->stream;
    Scratch sc;
    float *x, *xn, *qkv, *att, *hid, *qx, *xkv, *tokf;
    WM_TRY(sc.get(&x, (size_t)M * d, s));
    WM_TRY(sc.get(&xn, (size_t)M * d, s));
    WM_TRY(sc.get(&qkv, (size_t)M * 3 * d, s));
    WM_TRY(sc.get(&att, (size_t)M * d, s));
    WM_TRY(sc.get(&hid, (size_t)M * 4 * d, s));
    WM_TRY(sc.get(&qx, (size_t)M * d, s));
    WM_TRY(sc.get(&xkv, (size_t)MA * 2 * d, s));
    WM_TRY(sc.get(&tokf, (size_t)M, s));
    int *d_tok = (int *)tokf;
    WM_HIP(hipMemcpyAsync(d_tok, host_tokens, (size_t)M * sizeof(int), hipMemcpyHostToDevice, s));
    WM_HIP(hipStreamSynchronize(s));
    embed_f32_kernel<<<M, 256, 0, s>>>(d_tok, T, m->tok_emb, m->dec_pos, d, x);
    WM_HIP(hipGetLastError());
    for (int l = 0; l < D.n_text_layer; ++l) {
        const DecLayerW &L = m->dec[l];
        WM_TRY(wm_layernorm(ctx, x, L.ln1_g, L.ln1_b, M, d, nullptr, xn));
        WM_TRY(gemm(ctx, xn, d, L.wqkv, true, d, L.bqkv, qkv, 3 * d, M, 3 * d, d, F_STORE));
        WM_TRY(attention(ctx, qkv, qkv + d, qkv + 2 * d, att, B, H, T, T, 3 * d, 3 * d, 3 * d, d, true));
        WM_TRY(gemm(ctx, att, d, L.wo, true, d, L.bo, x, d, M, d, d, F_RESID));
        WM_TRY(wm_layernorm(ctx, x, L.lnx_g, L.lnx_b, M, d, nullptr, xn));
        WM_TRY(gemm(ctx, xn, d, L.wxq, true, d, L.bxq, qx, d, M, d, d, F_STORE));
        WM_TRY(gemm(ctx, d_xa, d, L.wxkv, false, d, L.bxkv, xkv, 2 * d, MA, 2 * d, d, F_STORE));
        WM_TRY(attention(ctx, qx, xkv, xkv + d, att, B, H, T, S, d, 2 * d, 2 * d, d, false));
        WM_TRY(gemm(ctx, att, d, L.wxo, true, d, L.bxo, x, d, M, d, d, F_RESID));
        WM_TRY(wm_layernorm(ctx, x, L.ln2_g, L.ln2_b, M, d, nullptr, xn));
        WM_TRY(gemm(ctx, xn, d, L.w1, true, d, L.b1, hid, 4 * d, M, 4 * d, d, F_GELU));
        WM_TRY(gemm(ctx, hid, 4 * d, L.w2, true, 4 * d, L.b2, x, d, M, d, 4 * d, F_RESID));
    }
    WM_TRY(wm_layernorm(ctx, x, m->ln_g, m->ln_b, M, d, nullptr, xn));
    WM_TRY(gemm(ctx, xn, d, m->tok_emb, true, d, nullptr, d_logits, V, M, V, d, F_STORE));
    WM_HIP(hipStreamSynchronize(s));
    return WM_OK;
}

const WmDebugHooks kF32Hooks = {f32_encode, f32_decode_logits};

}  // namespace

extern "C" int wmdbg_set_precision(wm_ctx *ctx, int precision) {
    WM_REQUIRE(ctx && ctx->model, WM_ERR_STATE, "wmdbg_set_precision: context has no model");
    WM_REQUIRE(precision == WM_F32 || precision == WM_BF16, WM_ERR_INVALID, "wmdbg_set_precision: WM_F32 or WM_BF16");
    ctx->dbg_hooks = precision == WM_F32 ? &kF32Hooks : nullptr;
    return WM_OK;
}
