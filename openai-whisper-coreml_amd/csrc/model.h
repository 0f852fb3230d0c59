// model.h -- Whisper encoder/decoder state (weights, activations, KV caches) and the
// launch wrappers of the HIP kernels that implement boundary #2
// (Whisper/Whisper/Whisper.swift:17-40; graph contract whisper_to_cml.py:10-43; arithmetic
// = openai-whisper AudioEncoder / TextDecoder).
#pragma once
#include "wm_internal.h"

// ---------------------------------------------------------------- HBM layout ----------
// All matrix weights are bf16 [N][K] row-major (PyTorch Linear layout: K contiguous), so
// both MFMA operands are K-contiguous.  Vectors (biases, LayerNorm, positions) are f32.
struct EncLayerW {
    float *ln1_g, *ln1_b;
    bf16_t *wqkv;  // [3d][d]  rows: query | key | value
    float *bqkv;   // [3d]     (key bias = 0: openai-whisper's key Linear has no bias)
    bf16_t *wo;    // [d][d]
    float *bo;
    float *ln2_g, *ln2_b;
    bf16_t *w1;  // [4d][d]
    float *b1;
    bf16_t *w2;  // [d][4d]
    float *b2;
};
struct DecLayerW {
    float *ln1_g, *ln1_b;
    bf16_t *wqkv;  // [3d][d]
    float *bqkv;
    bf16_t *wo;
    float *bo;
    float *lnx_g, *lnx_b;  // cross_attn_ln
    bf16_t *wxq;           // [d][d]
    float *bxq;
    bf16_t *wxkv;  // [2d][d]  rows: key | value   (applied to the encoder output)
    float *bxkv;   // [2d]     (key half = 0)
    bf16_t *wxo;
    float *bxo;
    float *ln2_g, *ln2_b;
    bf16_t *w1;
    float *b1;
    bf16_t *w2;
    float *b2;
    // LayerNorm-folded copies built by wm_finalize (decode GEMV operands): W' = bf16(W g), c1[n] = sum_k W'[n][k],
    // c2[n] = bias[n] + sum_k beta[k] W[n][k] -- so that LN(x) W^T + b = rstd (x W'^T - mean c1) + c2
    bf16_t *wqkv_f, *wxq_f, *w1_f;
    float *qkv_c1, *qkv_c2, *xq_c1, *xq_c2, *fc1_c1, *fc1_c2;
};

// One registry entry per openai-whisper state-dict key: where its elements live in HBM.
// WL_CONV : [O][C][3] -> [O][Kpad], k = tap*C + c (conv taps made contiguous for the implicit GEMM)
// WL_TILED: [N][K] -> MFMA-fragment-major tiles for the decode GEMV: tile (n/16, k/32) is the
//           1 KiB a wave loads with ONE global_load_dwordx4 (lane = n%16 + 16*((k%32)/8), 8 k
//           per lane), tiles ordered n-tile major, k-step minor -- a wave streams contiguous KiBs.
enum WmLayout { WL_PLAIN = 0, WL_CONV = 1, WL_TILED = 2 };
__host__ __device__ static inline size_t wm_tiled_offset(size_t n, size_t k, size_t K) {
    return (((n >> 4) * (K >> 5) + (k >> 5)) * 64 + (n & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7);
}
struct WmTensor {
    std::string name;
    void *ptr = nullptr;  // destination of logical element 0
    bool is_bf16 = false;
    size_t n_elems = 0;
    int layout = WL_PLAIN;
    int conv_c = 0, conv_kpad = 0;
    int kind = 0;  // K_MATRIX..K_SINUSOID of weights.py (synthetic generator)
    bool set = false;
};

// openai-whisper's ApplyTimestampRules (whisper/decoding.py [3p]) evaluated inside the fused logits / arg-max kernels.
// Per sequence the rules reduce to an allowed TEXT range and an allowed TIMESTAMP range of token ids for the next
// position (rng), maintained by the arg-max kernel from a 4-int history (hist), plus one comparison: if the summed
// probability of the allowed timestamps exceeds the best allowed text token, a timestamp is forced -- the logits kernel
// therefore emits, per 16-column tile, the best text key, the best timestamp key and a (max, sum exp) partial of the
// timestamp columns.
struct WmTsDev {
    int *rng;                     // [B][4] text_lo, text_hi, ts_lo, ts_hi for the next position; null = rules off
    int *hist;                    // [B][4] n_sampled, last_is_ts, prev_is_ts, last_ts
    unsigned long long *key_ts;   // [B][n_tiles] best allowed timestamp per tile (tiles >= ts_begin / 16 only)
    float *lse;                   // [B][n_tiles][2] (max, sum exp(v - max)) over the allowed timestamps of the tile
    int ts_begin, eot, n_vocab, max_initial;  // max_initial: index of the largest first timestamp, < 0 = unlimited
};

// Early-stop state of a decode group (device view; done == null: off).  A row is DONE once it has emitted `eot`
// (eot >= 0) or produced budget[b] tokens (budget != null); from then on its tokens are `pad_tok`, it is dropped from
// the compact live list the attention kernels walk, and when the list is empty the host stops launching positions.
struct WmStopDev {
    int *done;          // [B] 0 / 1
    const int *budget;  // [B] tokens a row may generate (null: max_new for all)
    int *live_rows;     // [B] compact, ascending list of the rows that are not done
    int *n_live;        // [1]
    int eot, pad_tok;
};

struct WmModel {
    wm_dims dims;
    bool finalized = false;
    bool shares_weights = false;  // clone: weight pointers alias the parent context's (never freed here)
    int k1pad = 0;  // conv1 GEMM K (3*n_mels rounded up to 64)
    int vpad = 0;   // n_vocab rounded up to 16 (token embedding rows)
    // weights
    bf16_t *conv1_w = nullptr, *conv2_w = nullptr;
    float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr;
    std::vector<EncLayerW> enc;
    float *ln_post_g = nullptr, *ln_post_b = nullptr;
    bf16_t *tok_emb = nullptr;  // [vpad][d]
    float *dec_pos = nullptr;   // [n_text_ctx][d]
    std::vector<DecLayerW> dec;
    float *ln_g = nullptr, *ln_b = nullptr;
    bf16_t *emb_f = nullptr;    // [vpad][d] token embedding with the final LayerNorm's gamma folded in (logits GEMV)
    float *logit_c1 = nullptr, *logit_c2 = nullptr;  // [vpad]
    std::vector<void *> allocs;
    std::vector<WmTensor> tensors;
    std::map<std::string, int> index;
    // activations (sized for `cap_b` chunks)
    int cap_b = 0;
    bf16_t *mel_t = nullptr;  // [B][3002][n_mels]  time-major, zero rows 0 and 3001
    bf16_t *h1p = nullptr;    // [B][3001][d]       conv1 output, zero row 0
    float *x = nullptr;       // [B*1500][d]        encoder residual stream (f32)
    bf16_t *xn = nullptr;     // [B*1500][d]        LayerNorm output / encoder output (bf16)
    bf16_t *qk = nullptr;     // [B*1500 + 64][2d]  q | k
    bf16_t *vt = nullptr;     // [B][H][64][1536]   v transposed (s contiguous)
    bf16_t *att = nullptr;    // [B*1500][d]
    bf16_t *hid = nullptr;    // [B*1500][4d]
    float *xa_f32 = nullptr;  // [B*1500][d]        ln_post output (API output)
    float *mel_f32 = nullptr; // [B][n_mels][3000]  front-end output kept on device
    bf16_t *xkv = nullptr;    // [L][2][B][H][1500][64]  cross-attention K/V cache
    bf16_t *skv = nullptr;    // [L][2][B][H][n_text_ctx][64] self-attention K/V cache
    // decode-step buffers
    float *dx = nullptr;        // [B][d]    decoder residual stream (f32)
    bf16_t *dxb = nullptr;      // [B][d]    its bf16 copy, MEAN-CENTRED: the A operand of the LayerNorm-folded GEMVs (WL_TILED order)
    float *dmean = nullptr;     // [2][B]    ping-pong: the rows' LayerNorm means = the centring offsets of dxb
    float *dq = nullptr;        // [16][d]   query (self or cross)
    float *dpart = nullptr;     // [16][H][WM_MAXSPLIT][66] attention partials (m, l, o[64])
    float *dstats = nullptr;    // [B/16][d/16][16][2] LayerNorm partial statistics of the residual stream
    bf16_t *datt = nullptr;     // [B][d]    attention head outputs (bf16 A operand of the out-projection; WL_TILED order)
    bf16_t *dhid = nullptr;     // [B][4d]   GELU(fc1) (A operand of fc2; WL_TILED order)
    float *dlogits = nullptr;   // [B][vpad]
    unsigned long long *dargmax = nullptr;  // [16][vpad/16] per-tile packed (value, ~index) maxima
    int *dresult = nullptr;     // [16] arg-max result relative to arg_first
    int *dseq = nullptr;        // [n_text_ctx][16->B] token sequence (prompt, then generated), position-major
    int *dpos = nullptr;        // [1] current decode position (read by every decode kernel)
    int *darrive = nullptr;     // [1] arrival counter of the arg-max workgroups (zero between launches)
    // early stop (wm_transcribe_greedy with eot >= 0 or per-chunk token budgets): see WmStopDev
    int *ddone = nullptr, *dbudget = nullptr, *dlive = nullptr, *dnlive = nullptr;
    int *h_nlive = nullptr;     // pinned host ring: n_live after each burst of positions (the host polls it)
    // The decode group being enqueued SHARES the chip with other groups (other lanes of the call, other contexts' calls):
    // its cross-attention is launched as one short-lived workgroup per (sequence, head) pair instead of <= 256 persistent
    // ones.  Alone, the persistent shape streams faster (56 rows: 66.8 vs 70.5 us); next to other groups' kernels the
    // short-lived one lets their workgroups in every ~13 us instead of once per launch: driver command 2076 -> 2100-2134
    // audio-s/s, default run 2190 -> 2250 (profiles/r04_xattn_short_lived.txt).  A launch shape: same bits.
    bool xattn_shared = false;
    bool stop_on = false;       // the decode being enqueued uses the stop state (kernel arguments of the captured graph)
    bool budget_on = false;
    int stop_eot = -1;
    std::vector<int32_t> budget_host;  // wm_set_token_budgets: per-chunk budgets of the NEXT call (empty: none)
    // Captured decode steps: one hipGraph of ONE position, and one of WM_BURST consecutive positions (the arg-max kernel
    // advances the device-side position, so consecutive positions do not depend on the host: 1/8 of the graph launches).
    // Everything a capture bakes into its kernel arguments is in the key; a lane keeps the last few shapes it ran
    // (a server alternating between batch sizes, bench.py's groups of 56 / 48 chunks landing on different lanes from one
    // pass to the next) instead of re-capturing ~2300 launches every time the shape changes (measured: the capture is
    // host work of a few ms that hides behind the lane's own encoder, so this is tidiness, not throughput).
    struct GraphSet {
        int B = 0, n_prompt = 0, cap_b = 0, mask = 0, stop_key = 0;
        // [mode]: 0 = the group has the chip to itself, 1 = it shares it (xattn_shared: short-lived cross-attention
        // workgroups); chosen burst by burst from the number of decodes in flight on the device, captured on first use
        int burst[2] = {0, 0};
        hipGraph_t g1[2] = {nullptr, nullptr}, gk[2] = {nullptr, nullptr};
        hipGraphExec_t e1[2] = {nullptr, nullptr}, ek[2] = {nullptr, nullptr};
        void destroy() {
            for (int i = 0; i < 2; ++i) {
                if (e1[i]) (void)hipGraphExecDestroy(e1[i]);
                if (g1[i]) (void)hipGraphDestroy(g1[i]);
                if (ek[i]) (void)hipGraphExecDestroy(ek[i]);
                if (gk[i]) (void)hipGraphDestroy(gk[i]);
                e1[i] = ek[i] = nullptr;
                g1[i] = gk[i] = nullptr;
            }
        }
        unsigned long stamp = 0;   // last use (LRU eviction)
    };
    // The ONE decoder step of a language-identification call (Whisper.swift:33-40: <|startoftranscript|> -> arg-max over the
    // language ids): ~100 launches of 3-5 us each, i.e. as long on the host (eager: ~3.5 us per launch) as on the GPU --
    // captured once per shape and replayed (round 6: the reference's own flow, device-resident, 2.23 -> see profiles/).
    struct LidGraph {
        int B = 0, cap_b = 0, first = 0, last = 0, logits = 0;
        hipGraph_t g = nullptr;
        hipGraphExec_t e = nullptr;
        void destroy() {
            if (e) (void)hipGraphExecDestroy(e);
            if (g) (void)hipGraphDestroy(g);
            e = nullptr; g = nullptr; B = 0;
        }
    } lid_graph;
    std::vector<int32_t> lid_host;   // host staging of the call's <|startoftranscript|> row (outlives the async upload)
    static constexpr int kMaxGraphSets = 4;
    std::vector<GraphSet> graph_sets;
    int graph_cur = -1;            // the set of the decode being enqueued
    unsigned long graph_clock = 0;
    unsigned *dmask = nullptr;   // [2][vpad/32] suppressed-token bitmaps (wm_set_suppress); [1] = first generated token
    bool mask_on = false;
    std::vector<unsigned> mask_host;  // host copy of the every-position bitmap
    // timestamp rules (wm_set_timestamp_rules): per-sequence state and per-tile partials, see WmTsDev
    bool ts_on = false;
    int ts_begin = 0, ts_eot = 0, ts_max_initial = -1;
    int *dts_rng = nullptr, *dts_hist = nullptr;
    unsigned long long *dts_key = nullptr;
    float *dts_lse = nullptr;
    void *pcm_stage = nullptr;  // host-pointer staging for wm_transcribe_greedy
    size_t pcm_stage_bytes = 0;
    float *io_stage = nullptr;  // staging for host-pointer model calls
    size_t io_stage_bytes = 0;
};

// model.cpp
int wm_model_create(wm_ctx *ctx, const wm_dims *dims);
void wm_model_destroy(wm_ctx *ctx);
int wm_model_clone(wm_ctx *child, const wm_ctx *parent);
int wm_model_reserve(wm_ctx *ctx, int B);
int wm_model_set_tensor(wm_ctx *ctx, const char *name, const float *data, size_t n);
int wm_model_get_tensor(wm_ctx *ctx, const char *name, float *data, size_t n);
int wm_model_init_synthetic(wm_ctx *ctx, uint64_t seed, float matrix_gain);
int wm_model_finalize(wm_ctx *ctx);
// device-pointer cores
int wm_model_encode_dev(wm_ctx *ctx, const float *d_mel, int B, float *d_xa_out /*nullable*/);
int wm_model_cross_kv(wm_ctx *ctx, int B);                 // from m->xn (bf16 encoder output)
int wm_model_set_xa(wm_ctx *ctx, const float *d_xa, int B);  // f32 xa -> m->xn (bf16)
int wm_model_decode_begin(wm_ctx *ctx, int B);
// One decoder position for all B sequences, position *m->dpos (device memory).  Expects the
// embedded input of that position in m->dx (+ m->dstats): wm_model_embed_first for the first
// position, afterwards produced by wm_model_close_step.  Ends with logits -> per-tile arg-max over
// [arg_first, arg_last] (m->dargmax); want_logits additionally stores f32 logits in m->dlogits.
int wm_model_decode_step(wm_ctx *ctx, int B, bool want_logits, int arg_first, int arg_last, int mask_first_pos = -1,
                         bool use_ts = false);
// the device view of the context's timestamp-rule state (rng == null when the rules are off)
WmTsDev wm_model_ts_dev(const WmModel *m);
int wm_model_set_timestamp_rules(wm_ctx *ctx, int enable, int32_t ts_begin, int32_t eot, int32_t max_initial);
// mask_first_pos >= 0: apply the suppress bitmaps, the first-token one at decode position mask_first_pos
int wm_model_set_suppress(wm_ctx *ctx, const int32_t *ids, int n, const int32_t *first_ids, int n_first);
int wm_model_embed_first(wm_ctx *ctx, int B);
// arg-max reduce + write next token (positions >= n_prompt) + embed next position + advance *dpos
int wm_model_close_step(wm_ctx *ctx, int B, int n_prompt, bool write_seq, int *result, int arg_first, bool use_ts = false);
// the device view of the early-stop state (done == null when m->stop_on is false)
WmStopDev wm_model_stop_dev(const WmModel *m);
void wm_model_drop_graphs(WmModel *m);
int wm_model_set_pos(wm_ctx *ctx, int pos);

// ---------------------------------------------------------------- kernel launchers ----
// gemm.hip
enum GemmEpi {
    EPI_BIAS_BF16 = 0,  // C bf16 [.][ldc] = acc + bias
    EPI_GELU_BF16 = 1,  // C bf16 = gelu(acc + bias)
    EPI_RESID_F32 = 2,  // C f32 += acc + bias
    EPI_CONV2_F32 = 3,  // C f32 = gelu(acc + bias) + pos[m % rows_per_batch][n]
    EPI_QKV_ENC = 4,    // n < 2d: C bf16 [m][2d] (query columns n < d multiplied by WM_ENC_QSCALE);  n >= 2d: vt[b][h][e][s]
    EPI_XKV = 5,        // cross K/V cache scatter
    EPI_F32 = 6         // C f32 = acc + bias (debug / generic)
};
// The encoder's queries are stored PRE-SCALED by hd^-1/2 * log2(e) (hd = 64): one f32 multiply in the QKV epilogue before
// its single bf16 rounding, so that the attention kernel's score accumulators are exponents of 2 already (enc_kernels.hip).
#define WM_ENC_QSCALE (0.125f * 1.44269504088896340736f)
// Key position s of a chunk -> its column in the encoder's V^T buffer [b][h][e][seq_pad]: the two middle 4-key groups of
// every 16 keys are swapped, so that the 8 keys one lane of the attention kernel multiplies per MFMA k-step are 16
// contiguous bytes (enc_kernels.hip).
__host__ __device__ static inline unsigned wm_att_vt_pos(unsigned s) { return (s & ~12u) | ((s & 4u) << 1) | ((s & 8u) >> 1); }
struct GemmArgs {
    const bf16_t *A;   // rows addressed as (m / a_rpb) * a_bstride + (m % a_rpb) * a_rstride
    long a_rpb, a_bstride, a_rstride;
    const bf16_t *W;   // [N][K]
    const float *bias; // [N] or null
    void *C;
    long c_rpb, c_bstride, c_rstride;  // same row addressing for C (elements)
    int M, N, K;
    int epi;
    // epilogue extras
    const float *pos;  // EPI_CONV2_F32
    bf16_t *vt;        // EPI_QKV_ENC
    int d_model, n_head, seq, seq_pad, batch;  // EPI_QKV_ENC / EPI_XKV
};
int wm_gemm(wm_ctx *ctx, const GemmArgs &g);

// enc_kernels.hip
int wm_layernorm(wm_ctx *ctx, const float *x, const float *g, const float *b, int rows, int d,
                 bf16_t *out_bf16 /*nullable*/, float *out_f32 /*nullable*/);
int wm_mel_to_time_major(wm_ctx *ctx, const float *mel, int B, int n_mels, bf16_t *mel_t);
int wm_f32_to_bf16(wm_ctx *ctx, const float *in, bf16_t *out, size_t n);
int wm_enc_attention(wm_ctx *ctx, const bf16_t *qk, const bf16_t *vt, bf16_t *att, int B, int H,
                     int S, int S_pad, int d);

// dec_kernels.hip
constexpr int WM_DEC_MAXB = 128;  // decode group: up to eight batch blocks of 16 rows (the MFMA M dimension)
constexpr int WM_NLIVE_RING = 16;  // pinned host slots for the per-burst live-row counts (early stop)
constexpr int WM_MAXSPLIT = 8;  // stream partials of a (sequence, head) pair of the cross-attention (small batches)
enum DecEpi { DE_QKV = 0, DE_Q = 1, DE_RESID = 2, DE_GELU = 3, DE_LOGITS = 4 };
struct DecGemvArgs {
    int epi;
    int B, N, K;
    const bf16_t *W;    // [N (padded to 16)][K] in WL_TILED order; LayerNorm mode: the gamma-folded copy
    const float *c1;    // LayerNorm mode (non-null): column sums of the folded weights, see DecLayerW
    const float *c2;    // [N] bias (LayerNorm mode: + beta fold) or null
    const bf16_t *a;    // [B padded to 16][K] bf16 activations in WL_TILED order (wm_tiled_offset(b, k, K))
    const float *stats_in; // LayerNorm mode: [B/16][K/16][16][2] partial (sum, sum of squares) per row of the f32 residual,
    int stats_parts;       //                 from the producer of the residual (always K/16 parts; unused ones are zero)
    float *stats_out;      // DE_RESID: [B/16][N/16][16][2] partials of the updated residual (may be null)
    // mean-centring of the bf16 residual copy (dec_kernels.hip, DecGemvDev::mean_in): [B] f32 each, nullable.
    // DE_RESID reads the offsets; a LayerNorm-mode launch reads them and writes the rows' new means to mean_out
    // (the OTHER buffer of a ping-pong pair).
    const float *mean_in;
    float *mean_out;
    // outputs
    float *out_f32;        // DE_QKV: q [B][N/3]; DE_Q: [B][ldo]; DE_RESID: residual [B][ldo] (+=); DE_LOGITS: [B][ldo] or null
    bf16_t *out_bf16;      // DE_GELU: [B][ldo]; DE_RESID: bf16 copy of the updated residual (may be null); WL_TILED order
    bf16_t *kcache, *vcache;  // DE_QKV: this layer's [B][H][T][64]
    const int *pos_ptr;       // device-side decode position (DE_QKV appends at *pos_ptr)
    int n_ctx, n_head;
    long ldo;
    unsigned long long *argmax;  // DE_LOGITS: per-tile packed maxima [B][ceil(N/16)] over [arg_first, arg_last]
    int arg_first, arg_last;
    // DE_LOGITS: suppressed-token bitmaps (bit n of word n/32), [2][mask_words]: [0] every position, [1] the position
    // *pos_ptr == mask_first_pos only (first generated token); null = no filter
    const unsigned *mask;
    int mask_words, mask_first_pos;
    WmTsDev ts;  // DE_LOGITS: timestamp rules (ts.rng == null: off)
    // optional L2 warm-up of the NEXT skinny GEMV's weights ([pf_rows][pf_k] bf16, WL_TILED)
    const bf16_t *pf_ptr;
    int pf_rows, pf_k;
    int pf_head_major;     // the next launch is wm_dec_xattn_fq: (pairs per XCD) place every head's tiles on the XCD(s) that run it
};
int wm_dec_gemv(wm_ctx *ctx, const DecGemvArgs &a);
// cross_attn_ln + query projection fused INTO the cross-attention launch (96 .. 256 pairs, alone on the device): qa = the
// DE_Q LayerNorm-mode arguments the separate GEMV would get (out_f32 unused).  Same bits as the two launches.
bool wm_dec_xattn_fq_applies(int B, int H, int K, bool short_lived);
int wm_dec_xattn_fq(wm_ctx *ctx, const DecGemvArgs &qa, const bf16_t *kc, const bf16_t *vc, int B, int H, int T_stride,
                    int n_keys, bf16_t *att, const int *live_rows, const int *n_live, const bf16_t *pf_ptr = nullptr,
                    int pf_rows = 0, int pf_k = 0);
// waves per workgroup and k-steps per wave the GEMV uses for a given K (a function of K only)
int wm_dec_gemv_split(int K, int *spw);
// W' = bf16(W g) (WL_TILED in, WL_TILED out), c1 = row sums of W', c2 = bias + W beta; rows N .. pad16(N) give zeros
int wm_ln_fold(wm_ctx *ctx, const bf16_t *W, const float *g, const float *beta, const float *bias /*nullable*/, int N,
               int K, bf16_t *Wf, float *c1, float *c2);
// x[b] = token_embedding[seq[*pos_ptr][b]] + positional_embedding[*pos_ptr]
int wm_dec_embed(wm_ctx *ctx, const int *seq, const int *pos_ptr, int B, const bf16_t *emb, const float *pemb,
                 int d, float *x, bf16_t *xb, float *stats_out, float *mean_buf /*nullable*/);
// Single-query attention over a K/V cache [B][H][T_stride][64] -> bf16 head outputs att[B][H*64] in WL_TILED order.
// Keys 0 .. n-1 with n = *pos_ptr + 1 when pos_ptr != null, else n_keys.
int wm_dec_attn_splits(int B, int H);
int wm_dec_attention(wm_ctx *ctx, const float *q, const bf16_t *kc, const bf16_t *vc, int B, int H,
                     int T_stride, int n_keys, const int *pos_ptr, int nsplit, float *part, bf16_t *att,
                     bool cross, const bf16_t *pf_ptr = nullptr, int pf_rows = 0, int pf_k = 0,
                     const int *live_rows = nullptr, const int *n_live = nullptr, bool short_lived = false);
// The decoder's causal self-attention (<= 448 cached rows per pair): one 4-wave workgroup per (sequence, head).
int wm_dec_self_attention(wm_ctx *ctx, const float *q, const bf16_t *kc, const bf16_t *vc, int B, int H, int T_stride,
                          int n_keys, const int *pos_ptr, bf16_t *att, const bf16_t *pf_ptr = nullptr, int pf_rows = 0,
                          int pf_k = 0, const int *live_rows = nullptr, const int *n_live = nullptr);
// Close a decode step (one workgroup): reduce the per-tile packed maxima of a DE_LOGITS launch;
// chosen token of row b -> seq[(*pos_ptr + 1) * B + b] when that position is >= n_prompt;
// (token - arg_first) -> result[b]; embed the tokens of position *pos_ptr + 1 into x (+ LayerNorm
// partial statistics) when x != null; then *pos_ptr += 1.  seq / pos_ptr / result / x may be null.
// arrive: zero-initialised device counter, required above 16 rows (one workgroup per 16 rows; the last one to arrive
// advances the position).  fallback_tok: the token taken when nothing is admissible (all suppressed / NaN logits).
int wm_argmax_embed(wm_ctx *ctx, const unsigned long long *tilemax, int n_tiles, int B, int *seq, int *pos_ptr,
                    int n_prompt, int *result, int arg_first, const bf16_t *emb, const float *pemb, int d, int n_ctx,
                    float *x, bf16_t *xb, float *stats_out, const WmTsDev *ts = nullptr, int *arrive = nullptr,
                    int fallback_tok = 0, float *mean_buf = nullptr, const WmStopDev *stop = nullptr);
// start of a decode with early stop: no row done, every row live
int wm_stop_init(wm_ctx *ctx, const WmStopDev &stop, int B);
// initial timestamp-rule state of B sequences (before the first sampled token)
int wm_ts_init(wm_ctx *ctx, const WmTsDev &ts, int B);
int wm_range_softmax(wm_ctx *ctx, const float *logits, long ldo, int B, int first, int n, float *probs);
int wm_fill_synthetic(wm_ctx *ctx, const WmTensor &t, uint32_t seed, int tensor_id, float gain);
