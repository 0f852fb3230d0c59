// debug_hooks.cpp -- the wmdbg_* test / micro-benchmark hooks of include/whisper_mi355x_debug.h.
// NOT linked into libwhisper_mi355x.so: build.py links this translation unit, together with the product's objects, into
// libwhisper_mi355x_dbg.so, which only tests/ and tools/ load (kernel-level parity tests against the oracle, probes).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/whisper_mi355x_debug.h"
#include "model.h"

// ---------------------------------------------------------------- host-only test hooks
int wm_gemm_set_tile_override(int tile);
extern "C" int wmdbg_set_gemm_tile(int tile) { return wm_gemm_set_tile_override(tile); }

// The launch-shape experiment knobs (wm_internal.h WmTuning): by name, so that tools/ scripts need no struct mirror.
extern "C" int wmdbg_set_tuning(const char *key, int value) {
    WM_REQUIRE(key, WM_ERR_INVALID, "wmdbg_set_tuning: null key");
    struct { const char *name; int *field; } table[] = {
        {"gemv_tn", &g_wm_tuning.gemv_tn}, {"gemv_nblk", &g_wm_tuning.gemv_nblk}, {"gemv_no_ppw2", &g_wm_tuning.gemv_no_ppw2}, {"gemv_ppw2_nblk", &g_wm_tuning.gemv_ppw2_nblk},
        {"prefetch_max_b", &g_wm_tuning.prefetch_max_b}, {"xattn_split_below", &g_wm_tuning.xattn_split_below},
        {"xattn_wgs", &g_wm_tuning.xattn_wgs}, {"xattn_no_flat", &g_wm_tuning.xattn_no_flat},
        {"xattn_lds_pad", &g_wm_tuning.xattn_lds_pad}, {"xattn_splits", &g_wm_tuning.xattn_splits},
        {"gemm_tile", &g_wm_tuning.gemm_tile}, {"gemm128_pipe", &g_wm_tuning.gemm128_pipe}, {"gemm_gm", &g_wm_tuning.gemm_gm}, {"no_early_stop", &g_wm_tuning.no_early_stop},
        {"xattn_no_deep", &g_wm_tuning.xattn_no_deep}, {"xattn_never_short", &g_wm_tuning.xattn_never_short},         {"logits_tn", &g_wm_tuning.logits_tn}, {"enc_attn_mfma_sum", &g_wm_tuning.enc_attn_mfma_sum},
        {"group_chunks", &g_wm_tuning.group_chunks}, {"argmax_rows_per_wg", &g_wm_tuning.argmax_rows_per_wg}, {"xattn_fuse_q", &g_wm_tuning.xattn_fuse_q}, {"xattn_pair_wg_max_pairs", &g_wm_tuning.xattn_pair_wg_max_pairs},
        {"lane_parts", &g_wm_tuning.lane_parts}, {"lane_solo_cus", &g_wm_tuning.lane_solo_cus}, {"frontend_per_wave_twiddles", &g_wm_tuning.frontend_per_wave_twiddles},
    };
    if (strcmp(key, "reset") == 0) { g_wm_tuning = WmTuning(); return WM_OK; }
    for (auto &e : table)
        if (strcmp(key, e.name) == 0) { *e.field = value; return WM_OK; }
    wm_set_error("wmdbg_set_tuning: unknown key '%s'", key);
    return WM_ERR_INVALID;
}

extern "C" int wmdbg_group_count(int B, int lanes, int explicit_lanes) { return wm_group_count(B, lanes, explicit_lanes != 0, 0); }
extern "C" int wmdbg_lane_parts(int B, int lanes, int explicit_lanes, int n_text_state) {
    return wm_lane_parts(B, lanes, explicit_lanes != 0, n_text_state, 0);
}
extern "C" int wmdbg_cu_mask(int cu_lo, int cu_hi, uint32_t *mask8) { return wm_cu_mask(cu_lo, cu_hi, mask8); }

extern "C" int wmdbg_mel_filterbank(int n_mels, float *out) {
    WM_REQUIRE(out && n_mels > 0 && n_mels <= 256, WM_ERR_INVALID, "bad args");
    std::vector<float> f;
    wm_mel_filterbank(n_mels, f);
    memcpy(out, f.data(), f.size() * sizeof(float));
    return WM_OK;
}
extern "C" int wmdbg_mel80(float *out) {
    WM_REQUIRE(out, WM_ERR_INVALID, "null");
    memcpy(out, wm_mel80_table(), sizeof(float) * 80 * 201);
    return WM_OK;
}

// ------------------------------------------------------------------ per-kernel test hooks
static int up(void **d, const void *h, size_t bytes, hipStream_t s) {
    WM_HIP(hipMalloc(d, bytes + 512));
    WM_HIP(hipMemsetAsync(*d, 0, bytes + 512, s));
    if (h) WM_HIP(hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, s));
    return WM_OK;
}
static void to_bf16(const float *in, std::vector<bf16_t> &out, size_t n) {
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
        uint32_t u;
        memcpy(&u, &in[i], 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        out[i] = (bf16_t)(u >> 16);
    }
}
static void from_bf16(const std::vector<bf16_t> &in, float *out) {
    for (size_t i = 0; i < in.size(); ++i) {
        uint32_t u = (uint32_t)in[i] << 16;
        memcpy(&out[i], &u, 4);
    }
}

extern "C" int wmdbg_gemm(wm_ctx *ctx, const float *A, const float *W, const float *bias, float *C, int M, int N,
                          int K, int epi) {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(epi == EPI_F32 || epi == EPI_BIAS_BF16 || epi == EPI_GELU_BF16 || epi == EPI_RESID_F32, WM_ERR_INVALID,
               "wmdbg_gemm: epilogue %d not exposed", epi);
    std::vector<bf16_t> a16, w16;
    to_bf16(A, a16, (size_t)M * K);
    to_bf16(W, w16, (size_t)N * K);
    void *dA, *dW, *dB = nullptr, *dC;
    hipStream_t s = ctx->stream;
    WM_TRY(up(&dA, a16.data(), a16.size() * 2, s));
    WM_TRY(up(&dW, w16.data(), w16.size() * 2, s));
    if (bias) WM_TRY(up(&dB, bias, (size_t)N * 4, s));
    const bool f32out = (epi == EPI_F32 || epi == EPI_RESID_F32);
    WM_TRY(up(&dC, epi == EPI_RESID_F32 ? C : nullptr, (size_t)M * N * (f32out ? 4 : 2), s));
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t *)dA; g.a_rpb = (long)M + 1; g.a_rstride = K;
    g.W = (const bf16_t *)dW; g.bias = (const float *)dB; g.C = dC;
    g.c_rpb = (long)M + 1; g.c_rstride = N; g.M = M; g.N = N; g.K = K; g.epi = epi;
    int rc = wm_gemm(ctx, g);
    if (rc == WM_OK) {
        if (f32out) {
            WM_HIP(hipMemcpyAsync(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost, s));
            WM_HIP(hipStreamSynchronize(s));
        } else {
            std::vector<bf16_t> c16((size_t)M * N);
            WM_HIP(hipMemcpyAsync(c16.data(), dC, c16.size() * 2, hipMemcpyDeviceToHost, s));
            WM_HIP(hipStreamSynchronize(s));
            from_bf16(c16, C);
        }
    }
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC);
    if (dB) (void)hipFree(dB);
    return rc;
}

extern "C" int wmdbg_layernorm(wm_ctx *ctx, const float *x, const float *g, const float *b, int rows, int d,
                               float *out_f32, float *out_bf16_as_f32) {
    WM_TRY(wm_ctx_make_current(ctx));
    void *dx, *dg, *db, *of, *ob;
    hipStream_t s = ctx->stream;
    WM_TRY(up(&dx, x, (size_t)rows * d * 4, s));
    WM_TRY(up(&dg, g, (size_t)d * 4, s));
    WM_TRY(up(&db, b, (size_t)d * 4, s));
    WM_TRY(up(&of, nullptr, (size_t)rows * d * 4, s));
    WM_TRY(up(&ob, nullptr, (size_t)rows * d * 2, s));
    int rc = wm_layernorm(ctx, (const float *)dx, (const float *)dg, (const float *)db, rows, d, (bf16_t *)ob, (float *)of);
    if (rc == WM_OK) {
        std::vector<bf16_t> t((size_t)rows * d);
        WM_HIP(hipMemcpyAsync(out_f32, of, (size_t)rows * d * 4, hipMemcpyDeviceToHost, s));
        WM_HIP(hipMemcpyAsync(t.data(), ob, t.size() * 2, hipMemcpyDeviceToHost, s));
        WM_HIP(hipStreamSynchronize(s));
        from_bf16(t, out_bf16_as_f32);
    }
    (void)hipFree(dx); (void)hipFree(dg); (void)hipFree(db); (void)hipFree(of); (void)hipFree(ob);
    return rc;
}

// Encoder attention on host q, k, v given as f32 [B][S][H*64] each (rounded to bf16 inside).
extern "C" int wmdbg_enc_attention(wm_ctx *ctx, const float *q, const float *k, const float *v, int B, int H, int S,
                                   float *out) {
    WM_TRY(wm_ctx_make_current(ctx));
    const int d = H * 64, S_pad = ((S + 63) / 64) * 64;
    const size_t M = (size_t)B * S;
    std::vector<bf16_t> qk((M + 64) * 2 * d, 0), vt((size_t)B * H * 64 * S_pad, 0), tmp;
    auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); };
    for (size_t mrow = 0; mrow < M; ++mrow)
        for (int j = 0; j < d; ++j) {
            qk[mrow * 2 * d + j] = bf(q[mrow * d + j] * WM_ENC_QSCALE);   // the kernel's contract: pre-scaled queries
            qk[mrow * 2 * d + d + j] = bf(k[mrow * d + j]);
        }
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s)
            for (int j = 0; j < d; ++j)
                vt[((size_t)(b * H + j / 64) * 64 + j % 64) * S_pad + wm_att_vt_pos((unsigned)s)] = bf(v[((size_t)b * S + s) * d + j]);
    void *dqk, *dvt, *datt;
    hipStream_t st = ctx->stream;
    WM_TRY(up(&dqk, qk.data(), qk.size() * 2, st));
    WM_TRY(up(&dvt, vt.data(), vt.size() * 2, st));
    WM_TRY(up(&datt, nullptr, M * d * 2, st));
    int rc = wm_enc_attention(ctx, (const bf16_t *)dqk, (const bf16_t *)dvt, (bf16_t *)datt, B, H, S, S_pad, d);
    if (rc == WM_OK) {
        tmp.resize(M * d);
        WM_HIP(hipMemcpyAsync(tmp.data(), datt, tmp.size() * 2, hipMemcpyDeviceToHost, st));
        WM_HIP(hipStreamSynchronize(st));
        from_bf16(tmp, out);
    }
    (void)hipFree(dqk); (void)hipFree(dvt); (void)hipFree(datt);
    return rc;
}

// Skinny decode GEMV: out[B][N] = LN?(x)[B][K] . W[N][K]^T + bias, driven the way the decoder drives it: bf16 activations,
// LayerNorm folded into the weights (W' = bf16(W g), c1, c2 by wm_ln_fold on device) and the row statistics handed over as
// partial sums of the f32 input.
extern "C" int wmdbg_dec_gemv(wm_ctx *ctx, const float *x, const float *ln_g, const float *ln_b, const float *W,
                              const float *bias, float *out, int B, int N, int K) {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(B >= 1 && B <= WM_DEC_MAXB, WM_ERR_INVALID, "wmdbg_dec_gemv: B out of range");
    const int Npad = ((N + 15) / 16) * 16;
    std::vector<bf16_t> w16, x16;
    std::vector<float> wp((size_t)Npad * K, 0.f);   // fragment-tiled order (WL_TILED)
    for (size_t r = 0; r < (size_t)N; ++r)
        for (size_t k = 0; k < (size_t)K; ++k) wp[wm_tiled_offset(r, k, (size_t)K)] = W[r * K + k];
    to_bf16(wp.data(), w16, wp.size());
    {   // activations in the fragment-tiled order the decoder keeps them in (rows padded to 16)
        const int Bpad = ((B + 15) / 16) * 16;
        std::vector<float> xp((size_t)Bpad * K, 0.f);
        for (size_t b = 0; b < (size_t)B; ++b)
            for (size_t k = 0; k < (size_t)K; ++k) xp[wm_tiled_offset(b, k, (size_t)K)] = x[b * K + k];
        to_bf16(xp.data(), x16, xp.size());
    }
    void *dx16, *dg = nullptr, *db = nullptr, *dW, *dWf = nullptr, *dc1 = nullptr, *dc2 = nullptr, *dbias = nullptr, *dout;
    hipStream_t s = ctx->stream;
    WM_TRY(up(&dx16, x16.data(), x16.size() * 2, s));
    WM_TRY(up(&dW, w16.data(), w16.size() * 2, s));
    WM_TRY(up(&dout, nullptr, (size_t)B * N * 4, s));
    if (bias) WM_TRY(up(&dbias, bias, (size_t)N * 4, s));
    DecGemvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.N = N; a.K = K; a.W = (const bf16_t *)dW; a.c2 = (const float *)dbias; a.a = (const bf16_t *)dx16;
    a.out_f32 = (float *)dout; a.ldo = N; a.epi = DE_Q; a.pos_ptr = nullptr;
    void *dst = nullptr;
    int rc = WM_OK;
    if (ln_g) {
        WM_TRY(up(&dg, ln_g, (size_t)K * 4, s));
        WM_TRY(up(&db, ln_b, (size_t)K * 4, s));
        WM_TRY(up(&dWf, nullptr, w16.size() * 2, s));
        WM_TRY(up(&dc1, nullptr, (size_t)Npad * 4, s));
        WM_TRY(up(&dc2, nullptr, (size_t)Npad * 4, s));
        rc = wm_ln_fold(ctx, (const bf16_t *)dW, (const float *)dg, (const float *)db, (const float *)dbias, N, K,
                        (bf16_t *)dWf, (float *)dc1, (float *)dc2);
        // the producer's partial statistics per 16-row block: K/16 parts (part p = columns 16p .. 16p+15), host-computed
        const int nblk = (B + 15) / 16;
        std::vector<float> st((size_t)nblk * 2 * K, 0.f);   // block stride = 2 K floats, as in the decoder
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k) {
                const int part = k >> 4;
                const float v = x[(size_t)b * K + k];
                float *blk = st.data() + (size_t)(b >> 4) * 2 * K;
                blk[(part * 16 + (b & 15)) * 2] += v;
                blk[(part * 16 + (b & 15)) * 2 + 1] += v * v;
            }
        WM_TRY(up(&dst, st.data(), st.size() * 4, s));
        a.stats_in = (const float *)dst; a.stats_parts = K / 16;
        a.W = (const bf16_t *)dWf; a.c1 = (const float *)dc1; a.c2 = (const float *)dc2;
    }
    if (rc == WM_OK) rc = wm_dec_gemv(ctx, a);
    if (rc == WM_OK) {
        WM_HIP(hipMemcpyAsync(out, dout, (size_t)B * N * 4, hipMemcpyDeviceToHost, s));
        WM_HIP(hipStreamSynchronize(s));
    }
    void *fr[] = {dx16, dg, db, dW, dWf, dc1, dc2, dbias, dout, dst};
    for (void *p : fr)
        if (p) (void)hipFree(p);
    return rc;
}

// The residual product of a decoder layer (DE_RESID: attention out-projection K = d, fc2 K = 4d) at any decode-group size:
// this is the epilogue the two-parts-per-wave kernel serves above 16 rows when K splits over 16 waves (d = 768 / 1024 / 1280).
extern "C" int wmdbg_dec_gemv_resid(wm_ctx *ctx, const float *x, const float *W, const float *bias, float *resid,
                                    float *copy_bf16, float *stats, int B, int N, int K) {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(B >= 1 && B <= WM_DEC_MAXB && N % 16 == 0, WM_ERR_INVALID, "wmdbg_dec_gemv_resid: B out of range / N % 16");
    const int Bpad = ((B + 15) / 16) * 16;
    std::vector<bf16_t> w16, x16;
    {
        std::vector<float> wp((size_t)N * K, 0.f), xp((size_t)Bpad * K, 0.f);
        for (size_t r = 0; r < (size_t)N; ++r)
            for (size_t k = 0; k < (size_t)K; ++k) wp[wm_tiled_offset(r, k, (size_t)K)] = W[r * K + k];
        for (size_t b = 0; b < (size_t)B; ++b)
            for (size_t k = 0; k < (size_t)K; ++k) xp[wm_tiled_offset(b, k, (size_t)K)] = x[b * K + k];
        to_bf16(wp.data(), w16, wp.size());
        to_bf16(xp.data(), x16, xp.size());
    }
    void *dx16, *dW, *dbias = nullptr, *dres, *dcopy, *dst;
    hipStream_t s = ctx->stream;
    WM_TRY(up(&dx16, x16.data(), x16.size() * 2, s));
    WM_TRY(up(&dW, w16.data(), w16.size() * 2, s));
    WM_TRY(up(&dres, resid, (size_t)B * N * 4, s));
    WM_TRY(up(&dcopy, nullptr, (size_t)Bpad * N * 2, s));
    WM_TRY(up(&dst, nullptr, (size_t)(Bpad / 16) * 2 * N * 4, s));
    if (bias) WM_TRY(up(&dbias, bias, (size_t)N * 4, s));
    DecGemvArgs a;
    memset(&a, 0, sizeof(a));
    a.epi = DE_RESID; a.B = B; a.N = N; a.K = K; a.W = (const bf16_t *)dW; a.c2 = (const float *)dbias;
    a.a = (const bf16_t *)dx16; a.out_f32 = (float *)dres; a.out_bf16 = (bf16_t *)dcopy; a.ldo = N;
    a.stats_out = (float *)dst;
    int rc = wm_dec_gemv(ctx, a);
    if (rc == WM_OK) {
        std::vector<bf16_t> c16((size_t)Bpad * N), lin((size_t)B * N);
        std::vector<float> st((size_t)(Bpad / 16) * 2 * N);
        WM_HIP(hipMemcpyAsync(resid, dres, (size_t)B * N * 4, hipMemcpyDeviceToHost, s));
        WM_HIP(hipMemcpyAsync(c16.data(), dcopy, c16.size() * 2, hipMemcpyDeviceToHost, s));
        WM_HIP(hipMemcpyAsync(st.data(), dst, st.size() * 4, hipMemcpyDeviceToHost, s));
        WM_HIP(hipStreamSynchronize(s));
        for (size_t b = 0; b < (size_t)B; ++b)
            for (size_t n = 0; n < (size_t)N; ++n) lin[b * N + n] = c16[wm_tiled_offset(b, n, (size_t)N)];
        from_bf16(lin, copy_bf16);
        for (int b = 0; b < B; ++b) {   // block (b / 16): [N/16 parts][16][2]
            const float *blk = st.data() + (size_t)(b >> 4) * 2 * N;
            float s1 = 0.f, s2 = 0.f;
            for (int part = 0; part < N / 16; ++part) {
                s1 += blk[(part * 16 + (b & 15)) * 2];
                s2 += blk[(part * 16 + (b & 15)) * 2 + 1];
            }
            stats[b * 2] = s1;
            stats[b * 2 + 1] = s2;
        }
    }
    void *fr[] = {dx16, dW, dbias, dres, dcopy, dst};
    for (void *p : fr)
        if (p) (void)hipFree(p);
    return rc;
}

// Single-query attention: q f32 [B][H*64], k/v f32 [B][H][T][64] (rounded to bf16), first
// n_keys positions; `nsplit` (1, 2, 4, 8) workgroups share the 8 streams of a pair; out f32 [B][H*64].
extern "C" int wmdbg_dec_attention(wm_ctx *ctx, const float *q, const float *k, const float *v, int B, int H, int T,
                                   int n_keys, int nsplit, float *out) {
    WM_TRY(wm_ctx_make_current(ctx));
    std::vector<bf16_t> k16, v16;
    to_bf16(k, k16, (size_t)B * H * T * 64);
    to_bf16(v, v16, (size_t)B * H * T * 64);
    void *dq, *dk, *dv, *dp;
    hipStream_t s = ctx->stream;
    WM_TRY(up(&dq, q, (size_t)B * H * 64 * 4, s));
    WM_TRY(up(&dk, k16.data(), k16.size() * 2, s));
    WM_TRY(up(&dv, v16.data(), v16.size() * 2, s));
    WM_TRY(up(&dp, nullptr, (size_t)B * H * WM_MAXSPLIT * 66 * 4, s));
    void *datt;
    WM_TRY(up(&datt, nullptr, ((size_t)B + 15) / 16 * 16 * H * 64 * 2, s));
    // nsplit == 0 selects the decoder's self-attention kernel (one 4-wave workgroup per pair), nsplit == -1 the
    // cross-attention launch path (8-wave block-streaming kernel, capped grid)
    int rc = nsplit == 0 ? wm_dec_self_attention(ctx, (const float *)dq, (const bf16_t *)dk, (const bf16_t *)dv, B, H, T,
                                                 n_keys, nullptr, (bf16_t *)datt)
                         : wm_dec_attention(ctx, (const float *)dq, (const bf16_t *)dk, (const bf16_t *)dv, B, H, T, n_keys,
                                            nullptr, nsplit < 0 ? 1 : nsplit, (float *)dp, (bf16_t *)datt, nsplit < 0);
    if (rc == WM_OK) {
        const size_t Bpad = ((size_t)B + 15) / 16 * 16, dd = (size_t)H * 64;
        std::vector<bf16_t> o16(Bpad * dd), lin((size_t)B * dd);
        WM_HIP(hipMemcpyAsync(o16.data(), datt, o16.size() * 2, hipMemcpyDeviceToHost, s));
        WM_HIP(hipStreamSynchronize(s));
        for (size_t b = 0; b < (size_t)B; ++b)   // head outputs are stored in the out-projection's tiled A-operand order
            for (size_t k = 0; k < dd; ++k) lin[b * dd + k] = o16[wm_tiled_offset(b, k, dd)];
        from_bf16(lin, out);
    }
    (void)hipFree(datt);
    (void)hipFree(dq); (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(dp);
    return rc;
}

// ------------------------------------------------------------------ micro-benchmarks ------
// Time `iters` back-to-back launches of one decode kernel, cycling over `n_mats` distinct weight
// matrices / cache slices so nothing is served from L2 or MALL.  Returns average microseconds
// per launch (HIP events on the context's stream).

extern "C" int wmdbg_bench_dec_gemv(wm_ctx *ctx, int B, int N, int K, int ln, int resid, int n_mats, int iters,
                                    int nw_override, float *avg_us) {
    (void)nw_override;  // the K split is a function of K alone (wm_dec_gemv_split)
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(B >= 1 && B <= WM_DEC_MAXB, WM_ERR_INVALID, "wmdbg_bench_dec_gemv: B out of range");
    hipStream_t s = ctx->stream;
    const int Npad = ((N + 15) / 16) * 16;
    void *dW, *dx16, *dc, *dout, *dout16;
    WM_TRY(up(&dW, nullptr, (size_t)n_mats * Npad * K * 2, s));
    WM_TRY(up(&dx16, nullptr, (size_t)WM_DEC_MAXB * K * 2, s));
    WM_TRY(up(&dc, nullptr, (size_t)Npad * 4, s));
    WM_TRY(up(&dout, nullptr, (size_t)WM_DEC_MAXB * Npad * 4, s));
    WM_TRY(up(&dout16, nullptr, (size_t)WM_DEC_MAXB * Npad * 2, s));
    WM_HIP(hipMemsetAsync(dW, 0x3c, (size_t)n_mats * Npad * K * 2, s));
    DecGemvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.N = N; a.K = K; a.out_f32 = (float *)dout; a.ldo = Npad; a.epi = resid ? DE_RESID : DE_Q;
    a.a = (const bf16_t *)dx16; a.c2 = (const float *)dc;
    void *dstat;
    WM_TRY(up(&dstat, nullptr, (size_t)(WM_DEC_MAXB / 16) * 2 * (K > N ? K : N) * 4 + 4096, s));
    if (resid) { a.stats_out = (float *)dstat; a.out_bf16 = (bf16_t *)dout16; }
    if (ln) { a.c1 = (const float *)dc; a.stats_in = (const float *)dstat; a.stats_parts = K / 16; }
    hipEvent_t e0, e1;
    WM_HIP(hipEventCreate(&e0));
    WM_HIP(hipEventCreate(&e1));
    int rc = WM_OK;
    // capture the launch chain once, replay it: per-kernel time = true serialized duration
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    WM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters && rc == WM_OK; ++i) {
        a.W = (const bf16_t *)dW + (size_t)(i % n_mats) * Npad * K;
        rc = wm_dec_gemv(ctx, a);
    }
    WM_HIP(hipStreamEndCapture(s, &g));
    if (rc == WM_OK) {
        WM_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        WM_HIP(hipGraphLaunch(ge, s));
        WM_HIP(hipStreamSynchronize(s));
        WM_HIP(hipEventRecord(e0, s));
        WM_HIP(hipGraphLaunch(ge, s));
    }
    WM_HIP(hipEventRecord(e1, s));
    WM_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    WM_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1e3f / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    void *fr[] = {dW, dx16, dc, dout, dout16, dstat};
    for (void *p : fr) (void)hipFree(p);
    return rc;
}

extern "C" int wmdbg_bench_dec_attention(wm_ctx *ctx, int B, int H, int T, int n_keys, int nsplit, int n_slices,
                                         int iters, float *avg_us) {
    WM_TRY(wm_ctx_make_current(ctx));
    hipStream_t s = ctx->stream;
    const size_t slice = (size_t)B * H * T * 64;
    void *dk, *dv, *dq, *dp, *datt;
    WM_TRY(up(&dk, nullptr, slice * n_slices * 2, s));
    WM_TRY(up(&dv, nullptr, slice * n_slices * 2, s));
    WM_TRY(up(&dq, nullptr, (size_t)B * H * 64 * 4, s));
    WM_TRY(up(&dp, nullptr, (size_t)B * H * WM_MAXSPLIT * 66 * 4, s));
    WM_TRY(up(&datt, nullptr, ((size_t)B + 15) / 16 * 16 * H * 64 * 2, s));
    hipEvent_t e0, e1;
    WM_HIP(hipEventCreate(&e0));
    WM_HIP(hipEventCreate(&e1));
    int rc = WM_OK;
    for (int pass = 0; pass < 2 && rc == WM_OK; ++pass) {
        if (pass == 1) WM_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < iters && rc == WM_OK; ++i)
            rc = wm_dec_attention(ctx, (const float *)dq, (const bf16_t *)dk + (size_t)(i % n_slices) * slice,
                                  (const bf16_t *)dv + (size_t)(i % n_slices) * slice, B, H, T, n_keys, nullptr, nsplit,
                                  (float *)dp, (bf16_t *)datt, true);
    }
    WM_HIP(hipEventRecord(e1, s));
    WM_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    WM_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1e3f / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    void *fr[] = {dk, dv, dq, dp, datt};
    for (void *p : fr) (void)hipFree(p);
    return rc;
}

int wm_launch_trivial(wm_ctx *ctx, int *p, int grid);
// Dependent-launch floor of this machine: `iters` trivial kernels (each increments one HBM word,
// so they are truly serialised), eager stream launches vs one captured hipGraph replayed.
extern "C" int wmdbg_bench_launch_floor(wm_ctx *ctx, int iters, int grid, float *eager_us, float *graph_us) {
    WM_TRY(wm_ctx_make_current(ctx));
    hipStream_t s = ctx->stream;
    void *d;
    WM_TRY(up(&d, nullptr, 64, s));
    hipEvent_t e0, e1;
    WM_HIP(hipEventCreate(&e0));
    WM_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) WM_TRY(wm_launch_trivial(ctx, (int *)d, grid));
    WM_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) WM_TRY(wm_launch_trivial(ctx, (int *)d, grid));
    WM_HIP(hipEventRecord(e1, s));
    WM_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    WM_HIP(hipEventElapsedTime(&ms, e0, e1));
    *eager_us = ms * 1e3f / iters;
    hipGraph_t g;
    hipGraphExec_t ge;
    WM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) (void)wm_launch_trivial(ctx, (int *)d, grid);
    WM_HIP(hipStreamEndCapture(s, &g));
    WM_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    WM_HIP(hipGraphLaunch(ge, s));
    WM_HIP(hipStreamSynchronize(s));
    WM_HIP(hipEventRecord(e0, s));
    WM_HIP(hipGraphLaunch(ge, s));
    WM_HIP(hipEventRecord(e1, s));
    WM_HIP(hipStreamSynchronize(s));
    WM_HIP(hipEventElapsedTime(&ms, e0, e1));
    *graph_us = ms * 1e3f / iters;
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
    return WM_OK;
}

// Mean duration (us) of `iters` back-to-back launches of one encoder GEMM shape.  Operands as in the encoder:
// A ~ N(0,1), W ~ N(0, 0.02^2), bias ~ 0.01 N(0,1); the launches rotate over `n_w` weight matrices so that W comes
// from HBM as it does in the 32-layer encoder (n_w = 1: W stays cache-resident).
extern "C" int wmdbg_bench_gemm(wm_ctx *ctx, int M, int N, int K, int epi, int iters, int n_w, float *us) {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(epi == EPI_F32 || epi == EPI_BIAS_BF16 || epi == EPI_GELU_BF16 || epi == EPI_RESID_F32, WM_ERR_INVALID,
               "wmdbg_bench_gemm: epilogue %d not exposed", epi);
    WM_REQUIRE(n_w >= 1 && n_w <= 64, WM_ERR_INVALID, "wmdbg_bench_gemm: n_w out of range");
    hipStream_t s = ctx->stream;
    std::vector<bf16_t> a16((size_t)M * K), w16((size_t)N * K);
    uint32_t x = 12345u;
    auto gauss = [&]() {  // Irwin-Hall(4), unit variance
        float acc = 0.f;
        for (int i = 0; i < 4; ++i) { x = x * 1664525u + 1013904223u; acc += (float)(x >> 8) * (1.0f / 16777216.0f); }
        return (acc - 2.0f) * 1.7320508f;
    };
    auto f2bf = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (bf16_t)(u >> 16);
    };
    for (auto &v : a16) v = f2bf(gauss());
    for (auto &v : w16) v = f2bf(0.02f * gauss());
    std::vector<float> bias(N);
    for (auto &v : bias) v = 0.01f * gauss();
    void *dA, *dB, *dC;
    std::vector<void *> dW(n_w, nullptr);
    WM_TRY(up(&dA, a16.data(), a16.size() * 2, s));
    for (int i = 0; i < n_w; ++i) WM_TRY(up(&dW[i], w16.data(), w16.size() * 2, s));
    WM_TRY(up(&dB, bias.data(), (size_t)N * 4, s));
    WM_TRY(up(&dC, nullptr, (size_t)M * N * 4, s));
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t *)dA; g.a_rpb = (long)M + 1; g.a_rstride = K;
    g.bias = (const float *)dB; g.C = dC;
    g.c_rpb = (long)M + 1; g.c_rstride = N; g.M = M; g.N = N; g.K = K; g.epi = epi;
    hipEvent_t e0, e1;
    WM_HIP(hipEventCreate(&e0));
    WM_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) { g.W = (const bf16_t *)dW[i % n_w]; WM_TRY(wm_gemm(ctx, g)); }
    WM_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) { g.W = (const bf16_t *)dW[i % n_w]; WM_TRY(wm_gemm(ctx, g)); }
    WM_HIP(hipEventRecord(e1, s));
    WM_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    WM_HIP(hipEventElapsedTime(&ms, e0, e1));
    *us = ms * 1e3f / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    for (void *w : dW) (void)hipFree(w);
    return WM_OK;
}

int wm_launch_spin(hipStream_t s, int *p, int grid, int cycles);
// Do two independent branches of a captured hipGraph run concurrently?  Each branch is a chain of
// `iters` kernels that spin ~`us_each` microseconds on `grid` workgroups.  Returns wall time of one
// replay with 1 branch and with 2 branches.
extern "C" int wmdbg_bench_graph_branches(wm_ctx *ctx, int iters, int grid, int us_each, float *one_us, float *two_us) {
    WM_TRY(wm_ctx_make_current(ctx));
    hipStream_t s = ctx->stream, s2;
    WM_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void *d;
    WM_TRY(up(&d, nullptr, 256, s));
    const int cycles = us_each * 100;  // wall_clock64 ticks at 100 MHz
    hipEvent_t e0, e1, ef, ej;
    WM_HIP(hipEventCreate(&e0)); WM_HIP(hipEventCreate(&e1));
    WM_HIP(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); WM_HIP(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int nb = 1; nb <= 2; ++nb) {
        hipGraph_t g; hipGraphExec_t ge;
        WM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        if (nb == 2) { WM_HIP(hipEventRecord(ef, s)); WM_HIP(hipStreamWaitEvent(s2, ef, 0)); }
        for (int i = 0; i < iters; ++i) {
            WM_TRY(wm_launch_spin(s, (int *)d, grid, cycles));
            if (nb == 2) WM_TRY(wm_launch_spin(s2, (int *)d + 16, grid, cycles));
        }
        if (nb == 2) { WM_HIP(hipEventRecord(ej, s2)); WM_HIP(hipStreamWaitEvent(s, ej, 0)); }
        WM_HIP(hipStreamEndCapture(s, &g));
        WM_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        WM_HIP(hipGraphLaunch(ge, s));
        WM_HIP(hipStreamSynchronize(s));
        WM_HIP(hipEventRecord(e0, s));
        WM_HIP(hipGraphLaunch(ge, s));
        WM_HIP(hipEventRecord(e1, s));
        WM_HIP(hipStreamSynchronize(s));
        float ms = 0.f;
        WM_HIP(hipEventElapsedTime(&ms, e0, e1));
        *(nb == 1 ? one_us : two_us) = ms * 1e3f;
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ef); (void)hipEventDestroy(ej);
    (void)hipStreamDestroy(s2); (void)hipFree(d);
    return WM_OK;
}
