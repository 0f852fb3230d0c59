/*
 * whisper_mi355x_debug.h -- test hooks exported by libwhisper_mi355x_dbg.so (the product library,
 * libwhisper_mi355x.so, does not contain them).
 *
 * NOT part of the drop-in surface: these let tests/ drive individual HIP kernels (and a
 * few host-side helpers) through the same C ABI conventions, so each kernel can be
 * compared with the oracle in isolation.  All pointers are HOST pointers; the hooks
 * stage through HBM themselves.
 */
#ifndef WHISPER_MI355X_DEBUG_H
#define WHISPER_MI355X_DEBUG_H

#include "whisper_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-only: slaney mel filterbank as librosa.filters.mel(sr=16000, n_fft=400, n_mels)
 * builds it (the recipe behind export_m80.py:4's mel_filters.npz); out: [n_mels][201]. */
WM_API int wmdbg_mel_filterbank(int n_mels, float *out);
/* Host-only: the embedded copy of the reference's m80.npy (80*201 f32). */
WM_API int wmdbg_mel80(float *out);

/* ---- single-kernel hooks (GPU).  Matrices are given as f32 and rounded to bf16 inside,
 * exactly as the weights / activations are held in HBM. ---------------------------------- */
/* C[M][N] = A[M][K] . W[N][K]^T (+bias); epi: 6 = f32 out, 0 = bf16 out, 1 = gelu -> bf16,
 * 2 = C += (f32 residual).  K % 64 == 0.  C is f32 on the host in every case. */
WM_API int wmdbg_gemm(wm_ctx *ctx, const float *A, const float *W, const float *bias, float *C, int M, int N,
               int K, int epi);
/* LayerNorm over the last axis (eps 1e-5): f32 result and the bf16 result widened to f32. */
WM_API int wmdbg_layernorm(wm_ctx *ctx, const float *x, const float *g, const float *b, int rows, int d,
                    float *out_f32, float *out_bf16_as_f32);
/* Non-causal MHA, head_dim 64: q,k,v,out f32 [B][S][H*64]. */
WM_API int wmdbg_enc_attention(wm_ctx *ctx, const float *q, const float *k, const float *v, int B, int H, int S,
                        float *out);
/* Decode-step skinny GEMM: out[B][N] = (ln_g ? LayerNorm(x) : x) . W[N][K]^T + bias; B <= 16. */
WM_API int wmdbg_dec_gemv(wm_ctx *ctx, const float *x, const float *ln_g, const float *ln_b, const float *W,
                   const float *bias, float *out, int B, int N, int K);
/* The decoder's residual product (out-projection / fc2): resid[B][N] += x[B][K] . W[N][K]^T + bias, any B <= 128.
 * Also returns the bf16 copy of the updated residual (widened to f32) and its per-row (sum, sum of squares) rebuilt from
 * the per-tile LayerNorm partials the kernel leaves for the next folded GEMV: stats [B][2]. */
WM_API int wmdbg_dec_gemv_resid(wm_ctx *ctx, const float *x, const float *W, const float *bias, float *resid, float *copy_bf16,
                         float *stats, int B, int N, int K);
/* Single-query attention over a cache: q [B][H*64], k/v [B][H][T][64], keys 0..n_keys-1;
 * out = the bf16 head outputs widened to f32.  nsplit in 1..8. */
WM_API int wmdbg_dec_attention(wm_ctx *ctx, const float *q, const float *k, const float *v, int B, int H, int T,
                        int n_keys, int nsplit, float *out);

/* ---- micro-benchmarks: average microseconds per launch over `iters` back-to-back launches
 * that cycle over n_mats weight matrices / n_slices cache slices (defeats L2 / MALL reuse). */
WM_API int wmdbg_bench_dec_gemv(wm_ctx *ctx, int B, int N, int K, int ln, int resid, int n_mats, int iters,
                         int nw_override, float *avg_us);
WM_API int wmdbg_bench_dec_attention(wm_ctx *ctx, int B, int H, int T, int n_keys, int nsplit, int n_slices,
                              int iters, float *avg_us);

/* Dependent-launch floor: average microseconds per trivial kernel, eager vs hipGraph replay. */
WM_API int wmdbg_bench_launch_floor(wm_ctx *ctx, int iters, int grid, float *eager_us, float *graph_us);

/* Wall time (us) of one replay of a captured graph with ONE chain of `iters` spinning kernels vs TWO
 * independent chains: tells whether hipGraph runs parallel branches concurrently on this runtime. */
WM_API int wmdbg_bench_graph_branches(wm_ctx *ctx, int iters, int grid, int us_each, float *one_us, float *two_us);

/* Mean duration (us) of one encoder GEMM launch, C[M][N] = A[M][K] W[N][K]^T, back to back, on encoder-like operands
 * (A ~ N(0,1), W ~ N(0,0.02^2)); launches rotate over n_w copies of W (n_w large: W streams from HBM as in the model). */
WM_API int wmdbg_bench_gemm(wm_ctx *ctx, int M, int N, int K, int epi, int iters, int n_w, float *us);

/* Force the encoder GEMM tile: 128 (128 x 128, 4 waves), 256 (256 x 256, 8 waves, staggered phases) or 0 = automatic.
 * Process-wide; used by the parity tests and A/B probes to run every shape through both kernels. */
/* sub-chip lanes (round 6): CU-masked decode groups of a wm_transcribe_greedy call (0: none, 2: two half-chip groups) for a
 * call of B chunks on a model of decoder width n_text_state; the 256-bit CU mask of the CUs [cu_lo, cu_hi) of every XCD */
WM_API int wmdbg_lane_parts(int B, int lanes, int explicit_lanes, int n_text_state);
WM_API int wmdbg_cu_mask(int cu_lo, int cu_hi, uint32_t *mask8);
WM_API int wmdbg_set_gemm_tile(int tile);

/* The ALL-FP32 debug model path (BASELINE.md parity gate: "fp32 debug path must match to <= 1e-4 rel-L2"; csrc/f32_path.hip).
 * precision = WM_F32: wm_encode and wm_decode_logits of THIS context run with f32 activations, f32 K/V and f32 accumulation on
 * the very weights the product multiplies (the bf16 values in HBM, in their product layouts) -- separates bugs from rounding.
 * WM_BF16 restores the product kernels.  wm_detect_language / wm_transcribe_greedy are not affected. */
WM_API int wmdbg_set_precision(wm_ctx *ctx, int precision);

/* Launch-shape experiment knobs (csrc/wm_internal.h, struct WmTuning), by name: "gemv_tn", "gemv_nblk", "gemv_no_ppw2",
 * "prefetch_max_b", "xattn_split_below", "xattn_wgs", "xattn_no_flat", "xattn_lds_pad", "xattn_splits", "gemm_tile",
 * "gemm_gm", "no_early_stop", "xattn_no_deep", "xattn_never_short", "logits_tn", "enc_attn_mfma_sum"; key "reset" restores the product's rules.  Process-wide.  The PRODUCT library has no such
 * entry point and reads no environment variable for launch shapes (rounds 1-3 had WM_GEMV_*, WM_XATTN_*, WM_GEMM_*). */
WM_API int wmdbg_set_tuning(const char *key, int value);
/* The product's group policy as a pure function: decode groups of a wm_transcribe_greedy call of B chunks with `lanes` lanes
 * available; explicit_lanes != 0: the host set the lane count with wm_set_lanes (host only, no GPU). */
WM_API int wmdbg_group_count(int B, int lanes, int explicit_lanes);

#ifdef __cplusplus
}
#endif
#endif
