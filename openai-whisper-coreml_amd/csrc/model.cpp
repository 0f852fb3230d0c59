// model.cpp -- weights registry, HBM buffers and the launch sequences of the encoder
// (openai-whisper AudioEncoder.forward, traced at whisper_to_cml.py:10-23), the
// cross-attention K/V projection, and one KV-cached decoder position
// (TextDecoder.forward, traced at whisper_to_cml.py:25-43).
#include "model.h"

#include <math.h>
#include <string.h>

static inline bf16_t host_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static inline float host_bf2f(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static int dalloc(WmModel *m, void **p, size_t bytes, hipStream_t s) {
    bytes = (bytes + 255) & ~(size_t)255;
    WM_HIP(hipMalloc(p, bytes + 256));  // + slack: tile loaders may over-read by < 256 B
    WM_HIP(hipMemsetAsync(*p, 0, bytes + 256, s));
    m->allocs.push_back(*p);
    return WM_OK;
}
template <typename T>
static int dalloc_t(WmModel *m, T **p, size_t n, hipStream_t s) {
    return dalloc(m, (void **)p, n * sizeof(T), s);
}

static void reg(WmModel *m, const std::string &name, void *ptr, bool is_bf16, size_t n, int kind,
                int layout = WL_PLAIN, int conv_c = 0, int kpad = 0) {
    WmTensor t;
    t.name = name; t.ptr = ptr; t.is_bf16 = is_bf16; t.n_elems = n; t.kind = kind;
    t.layout = layout; t.conv_c = conv_c; t.conv_kpad = kpad;
    m->index[name] = (int)m->tensors.size();
    m->tensors.push_back(t);
}

static int alloc_decode_buffers(WmModel *m, hipStream_t s) {
    const wm_dims &D = m->dims;
    const int d = D.n_text_state;
    // decode-step buffers (batch <= WM_DEC_MAXB)
    WM_TRY(dalloc_t(m, &m->dx, (size_t)WM_DEC_MAXB * d, s));
    WM_TRY(dalloc_t(m, &m->dxb, (size_t)WM_DEC_MAXB * d, s));
    WM_TRY(dalloc_t(m, &m->dmean, (size_t)2 * WM_DEC_MAXB, s));
    WM_TRY(dalloc_t(m, &m->dq, (size_t)WM_DEC_MAXB * d, s));
    WM_TRY(dalloc_t(m, &m->dpart, (size_t)WM_DEC_MAXB * D.n_text_head * WM_MAXSPLIT * 66, s));
    WM_TRY(dalloc_t(m, &m->datt, (size_t)WM_DEC_MAXB * d, s));
    WM_TRY(dalloc_t(m, &m->dstats, (size_t)(WM_DEC_MAXB / 16) * (d / 16) * 16 * 2, s));
    WM_TRY(dalloc_t(m, &m->dhid, (size_t)WM_DEC_MAXB * 4 * d, s));
    WM_TRY(dalloc_t(m, &m->dlogits, (size_t)WM_DEC_MAXB * m->vpad, s));
    WM_TRY(dalloc_t(m, &m->dargmax, (size_t)WM_DEC_MAXB * (m->vpad / 16), s));
    WM_TRY(dalloc_t(m, &m->dresult, WM_DEC_MAXB, s));
    WM_TRY(dalloc_t(m, &m->dseq, (size_t)WM_DEC_MAXB * (D.n_text_ctx + 1), s));
    WM_TRY(dalloc_t(m, &m->dpos, 4, s));
    WM_TRY(dalloc_t(m, &m->darrive, 4, s));
    WM_TRY(dalloc_t(m, &m->ddone, WM_DEC_MAXB, s));
    WM_TRY(dalloc_t(m, &m->dbudget, WM_DEC_MAXB, s));
    WM_TRY(dalloc_t(m, &m->dlive, WM_DEC_MAXB + 4, s));   // [live rows | n_live]: ONE pointer argument for the attention
    m->dnlive = m->dlive + WM_DEC_MAXB;                  // kernels (their first loads need everything in 14 dwords)
    if (!m->h_nlive) WM_HIP(hipHostMalloc((void **)&m->h_nlive, WM_NLIVE_RING * sizeof(int), hipHostMallocDefault));
    WM_TRY(dalloc_t(m, &m->dts_rng, (size_t)WM_DEC_MAXB * 4, s));
    WM_TRY(dalloc_t(m, &m->dts_hist, (size_t)WM_DEC_MAXB * 4, s));
    WM_TRY(dalloc_t(m, &m->dts_key, (size_t)WM_DEC_MAXB * (m->vpad / 16), s));
    WM_TRY(dalloc_t(m, &m->dts_lse, (size_t)WM_DEC_MAXB * (m->vpad / 16) * 2, s));
    WM_TRY(dalloc_t(m, &m->dmask, (size_t)2 * (m->vpad / 32), s));
    WM_HIP(hipMemsetAsync(m->dmask, 0, (size_t)2 * (m->vpad / 32) * 4, s));
    return WM_OK;
}

WmStopDev wm_model_stop_dev(const WmModel *m) {
    WmStopDev t;
    memset(&t, 0, sizeof(t));
    if (!m->stop_on) return t;
    t.done = m->ddone; t.budget = m->budget_on ? m->dbudget : nullptr; t.live_rows = m->dlive; t.n_live = m->dnlive;
    t.eot = m->stop_eot;
    t.pad_tok = m->stop_eot >= 0 ? m->stop_eot : 0;   // what a finished row keeps embedding: any valid id
    return t;
}

void wm_model_drop_graphs(WmModel *m) {
    for (WmModel::GraphSet &g : m->graph_sets) g.destroy();
    m->graph_sets.clear();
    m->graph_cur = -1;
    m->lid_graph.destroy();
}

WmTsDev wm_model_ts_dev(const WmModel *m) {
    WmTsDev t;
    memset(&t, 0, sizeof(t));
    if (!m->ts_on) return t;
    t.rng = m->dts_rng; t.hist = m->dts_hist; t.key_ts = m->dts_key; t.lse = m->dts_lse;
    t.ts_begin = m->ts_begin; t.eot = m->ts_eot; t.n_vocab = m->dims.n_vocab; t.max_initial = m->ts_max_initial;
    return t;
}

int wm_model_set_timestamp_rules(wm_ctx *ctx, int enable, int32_t ts_begin, int32_t eot, int32_t max_initial) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m, WM_ERR_STATE, "context has no model");
    if (enable) {
        const int V = m->dims.n_vocab;
        WM_REQUIRE(ts_begin > 0 && ts_begin < V && eot >= 0 && eot < ts_begin, WM_ERR_INVALID,
                   "timestamp rules: need 0 <= eot < timestamp_begin < n_vocab (%d)", V);
        // <|endoftext|> is the token of last resort of the rules (an opening timestamp may be followed by it, and it is
        // what the arg-max kernel falls back to when nothing else is admissible): it must not be suppressed
        if (!m->mask_host.empty())
            WM_REQUIRE(!((m->mask_host[eot >> 5] >> (eot & 31)) & 1u), WM_ERR_INVALID,
                       "timestamp rules: <|endoftext|> (%d) is in the suppress list", eot);
        if (m->ts_begin != ts_begin || m->ts_eot != eot || m->ts_max_initial != max_initial) {
            // these ids are baked into the captured decode graphs' kernel arguments: drop the stale captures
            wm_model_drop_graphs(m);
        }
        m->ts_begin = ts_begin; m->ts_eot = eot; m->ts_max_initial = max_initial;
    }
    m->ts_on = enable != 0;
    return WM_OK;
}

// SuppressTokens / SuppressBlank of openai-whisper's decoding.py as two bitmaps over the vocabulary.
int wm_model_set_suppress(wm_ctx *ctx, const int32_t *ids, int n, const int32_t *first_ids, int n_first) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m, WM_ERR_STATE, "context has no model");
    WM_REQUIRE(n >= 0 && n_first >= 0 && (n == 0 || ids) && (n_first == 0 || first_ids), WM_ERR_INVALID,
               "set_suppress: bad list");
    const int words = m->vpad / 32, V = m->dims.n_vocab;
    std::vector<unsigned> bits((size_t)2 * words, 0u);
    for (int i = 0; i < n; ++i) {
        WM_REQUIRE(ids[i] >= 0 && ids[i] < V, WM_ERR_INVALID, "set_suppress: token %d outside [0, %d)", ids[i], V);
        bits[ids[i] >> 5] |= 1u << (ids[i] & 31);
    }
    for (int i = 0; i < words; ++i) bits[words + i] = bits[i];  // first position: always-list OR first-list
    for (int i = 0; i < n_first; ++i) {
        WM_REQUIRE(first_ids[i] >= 0 && first_ids[i] < V, WM_ERR_INVALID, "set_suppress: token %d outside [0, %d)",
                   first_ids[i], V);
        bits[words + (first_ids[i] >> 5)] |= 1u << (first_ids[i] & 31);
    }
    size_t live0 = 0, live1 = 0;
    for (int t = 0; t < V; ++t) {
        live0 += !((bits[t >> 5] >> (t & 31)) & 1u);
        live1 += !((bits[words + (t >> 5)] >> (t & 31)) & 1u);
    }
    WM_REQUIRE(live0 > 0 && live1 > 0, WM_ERR_INVALID, "set_suppress: every token would be suppressed");
    if (m->ts_on)
        WM_REQUIRE(!((bits[m->ts_eot >> 5] >> (m->ts_eot & 31)) & 1u), WM_ERR_INVALID,
                   "set_suppress: <|endoftext|> (%d) must stay admissible while the timestamp rules are on", m->ts_eot);
    m->mask_host.assign(bits.begin(), bits.begin() + words);   // the every-position list (timestamp-rule validation)
    WM_HIP(hipMemcpyAsync(m->dmask, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    m->mask_on = (n + n_first) > 0;
    return WM_OK;
}

// kinds mirror weights.py: 0 matrix, 1 bias, 2 LN weight, 3 LN bias, 4 sinusoid
int wm_model_create(wm_ctx *ctx, const wm_dims *dims) {
    const wm_dims &D = *dims;
    WM_REQUIRE(D.n_mels == 80 || D.n_mels == 128, WM_ERR_INVALID, "n_mels must be 80 or 128");
    WM_REQUIRE(D.n_audio_ctx == 1500, WM_ERR_INVALID, "n_audio_ctx must be 1500 (30 s chunks)");
    WM_REQUIRE(D.n_audio_state == D.n_text_state, WM_ERR_INVALID, "audio/text widths must match");
    WM_REQUIRE(D.n_audio_state % 64 == 0 && D.n_audio_state <= 1280 && D.n_audio_state >= 64,
               WM_ERR_INVALID, "model width must be a multiple of 64 in [64, 1280]");
    WM_REQUIRE(D.n_audio_head * 64 == D.n_audio_state && D.n_text_head * 64 == D.n_text_state,
               WM_ERR_INVALID, "head_dim must be 64 (Whisper uses 64 at every size)");
    WM_REQUIRE(D.n_text_ctx >= 8 && D.n_text_ctx <= 448 && D.n_vocab >= 16, WM_ERR_INVALID, "bad text dims");
    WM_REQUIRE(D.n_audio_layer >= 1 && D.n_text_layer >= 1, WM_ERR_INVALID, "bad layer counts");
    WM_REQUIRE(wm_dec_gemv_split(D.n_text_state, nullptr) > 0 && wm_dec_gemv_split(4 * D.n_text_state, nullptr) > 0,
               WM_ERR_INVALID, "model width %d: the decode GEMV cannot split K = d / 4d over its waves", D.n_text_state);
    WmModel *m = new WmModel();
    ctx->model = m;
    m->dims = D;
    const int d = D.n_audio_state;
    hipStream_t s = ctx->stream;
    m->k1pad = ((3 * D.n_mels + 63) / 64) * 64;
    m->vpad = ((D.n_vocab + 15) / 16) * 16;

    WM_TRY(dalloc_t(m, &m->conv1_w, (size_t)d * m->k1pad, s));
    WM_TRY(dalloc_t(m, &m->conv1_b, d, s));
    WM_TRY(dalloc_t(m, &m->conv2_w, (size_t)d * 3 * d, s));
    WM_TRY(dalloc_t(m, &m->conv2_b, d, s));
    WM_TRY(dalloc_t(m, &m->enc_pos, (size_t)D.n_audio_ctx * d, s));
    reg(m, "encoder.conv1.weight", m->conv1_w, true, (size_t)d * D.n_mels * 3, 0, WL_CONV, D.n_mels, m->k1pad);
    reg(m, "encoder.conv1.bias", m->conv1_b, false, d, 1);
    reg(m, "encoder.conv2.weight", m->conv2_w, true, (size_t)d * d * 3, 0, WL_CONV, d, 3 * d);
    reg(m, "encoder.conv2.bias", m->conv2_b, false, d, 1);
    reg(m, "encoder.positional_embedding", m->enc_pos, false, (size_t)D.n_audio_ctx * d, 4);

    // lq / lkv / lo: HBM layout of the query, key+value and out matrices (decode GEMV operands are
    // stored fragment-tiled; GEMM operands row-major)
    auto attn_regs = [&](const std::string &p, bf16_t *wq, float *bq, bf16_t *wk, bf16_t *wv, float *bv,
                         bf16_t *wo, float *bo, float *lg, float *lb, const char *a, int lq, int lkv, int lo) {
        const size_t dd = (size_t)d * d;
        reg(m, p + "." + a + ".query.weight", wq, true, dd, 0, lq, 0, d);
        reg(m, p + "." + a + ".query.bias", bq, false, d, 1);
        reg(m, p + "." + a + ".key.weight", wk, true, dd, 0, lkv, 0, d);
        reg(m, p + "." + a + ".value.weight", wv, true, dd, 0, lkv, 0, d);
        reg(m, p + "." + a + ".value.bias", bv, false, d, 1);
        reg(m, p + "." + a + ".out.weight", wo, true, dd, 0, lo, 0, d);
        reg(m, p + "." + a + ".out.bias", bo, false, d, 1);
        reg(m, p + "." + a + "_ln.weight", lg, false, d, 2);
        reg(m, p + "." + a + "_ln.bias", lb, false, d, 3);
    };
    auto mlp_regs = [&](const std::string &p, bf16_t *w1, float *b1, bf16_t *w2, float *b2, float *lg, float *lb,
                        int lay) {
        reg(m, p + ".mlp.0.weight", w1, true, (size_t)4 * d * d, 0, lay, 0, d);
        reg(m, p + ".mlp.0.bias", b1, false, 4 * d, 1);
        reg(m, p + ".mlp.2.weight", w2, true, (size_t)4 * d * d, 0, lay, 0, 4 * d);
        reg(m, p + ".mlp.2.bias", b2, false, d, 1);
        reg(m, p + ".mlp_ln.weight", lg, false, d, 2);
        reg(m, p + ".mlp_ln.bias", lb, false, d, 3);
    };

    m->enc.resize(D.n_audio_layer);
    for (int i = 0; i < D.n_audio_layer; ++i) {
        EncLayerW &L = m->enc[i];
        WM_TRY(dalloc_t(m, &L.ln1_g, d, s)); WM_TRY(dalloc_t(m, &L.ln1_b, d, s));
        WM_TRY(dalloc_t(m, &L.wqkv, (size_t)3 * d * d, s)); WM_TRY(dalloc_t(m, &L.bqkv, 3 * d, s));
        WM_TRY(dalloc_t(m, &L.wo, (size_t)d * d, s)); WM_TRY(dalloc_t(m, &L.bo, d, s));
        WM_TRY(dalloc_t(m, &L.ln2_g, d, s)); WM_TRY(dalloc_t(m, &L.ln2_b, d, s));
        WM_TRY(dalloc_t(m, &L.w1, (size_t)4 * d * d, s)); WM_TRY(dalloc_t(m, &L.b1, 4 * d, s));
        WM_TRY(dalloc_t(m, &L.w2, (size_t)4 * d * d, s)); WM_TRY(dalloc_t(m, &L.b2, d, s));
        const std::string p = "encoder.blocks." + std::to_string(i);
        attn_regs(p, L.wqkv, L.bqkv, L.wqkv + (size_t)d * d, L.wqkv + (size_t)2 * d * d, L.bqkv + 2 * d, L.wo,
                  L.bo, L.ln1_g, L.ln1_b, "attn", WL_PLAIN, WL_PLAIN, WL_PLAIN);
        mlp_regs(p, L.w1, L.b1, L.w2, L.b2, L.ln2_g, L.ln2_b, WL_PLAIN);
    }
    WM_TRY(dalloc_t(m, &m->ln_post_g, d, s)); WM_TRY(dalloc_t(m, &m->ln_post_b, d, s));
    reg(m, "encoder.ln_post.weight", m->ln_post_g, false, d, 2);
    reg(m, "encoder.ln_post.bias", m->ln_post_b, false, d, 3);

    WM_TRY(dalloc_t(m, &m->tok_emb, (size_t)m->vpad * d, s));
    WM_TRY(dalloc_t(m, &m->dec_pos, (size_t)D.n_text_ctx * d, s));
    reg(m, "decoder.token_embedding.weight", m->tok_emb, true, (size_t)D.n_vocab * d, 0, WL_TILED, 0, d);
    reg(m, "decoder.positional_embedding", m->dec_pos, false, (size_t)D.n_text_ctx * d, 0);
    m->dec.resize(D.n_text_layer);
    for (int i = 0; i < D.n_text_layer; ++i) {
        DecLayerW &L = m->dec[i];
        WM_TRY(dalloc_t(m, &L.ln1_g, d, s)); WM_TRY(dalloc_t(m, &L.ln1_b, d, s));
        WM_TRY(dalloc_t(m, &L.wqkv, (size_t)3 * d * d, s)); WM_TRY(dalloc_t(m, &L.bqkv, 3 * d, s));
        WM_TRY(dalloc_t(m, &L.wo, (size_t)d * d, s)); WM_TRY(dalloc_t(m, &L.bo, d, s));
        WM_TRY(dalloc_t(m, &L.lnx_g, d, s)); WM_TRY(dalloc_t(m, &L.lnx_b, d, s));
        WM_TRY(dalloc_t(m, &L.wxq, (size_t)d * d, s)); WM_TRY(dalloc_t(m, &L.bxq, d, s));
        WM_TRY(dalloc_t(m, &L.wxkv, (size_t)2 * d * d, s)); WM_TRY(dalloc_t(m, &L.bxkv, 2 * d, s));
        WM_TRY(dalloc_t(m, &L.wxo, (size_t)d * d, s)); WM_TRY(dalloc_t(m, &L.bxo, d, s));
        WM_TRY(dalloc_t(m, &L.ln2_g, d, s)); WM_TRY(dalloc_t(m, &L.ln2_b, d, s));
        WM_TRY(dalloc_t(m, &L.w1, (size_t)4 * d * d, s)); WM_TRY(dalloc_t(m, &L.b1, 4 * d, s));
        WM_TRY(dalloc_t(m, &L.w2, (size_t)4 * d * d, s)); WM_TRY(dalloc_t(m, &L.b2, d, s));
        // LayerNorm-folded decode operands (filled by wm_finalize)
        WM_TRY(dalloc_t(m, &L.wqkv_f, (size_t)3 * d * d, s)); WM_TRY(dalloc_t(m, &L.wxq_f, (size_t)d * d, s));
        WM_TRY(dalloc_t(m, &L.w1_f, (size_t)4 * d * d, s));
        WM_TRY(dalloc_t(m, &L.qkv_c1, 3 * d, s)); WM_TRY(dalloc_t(m, &L.qkv_c2, 3 * d, s));
        WM_TRY(dalloc_t(m, &L.xq_c1, d, s)); WM_TRY(dalloc_t(m, &L.xq_c2, d, s));
        WM_TRY(dalloc_t(m, &L.fc1_c1, 4 * d, s)); WM_TRY(dalloc_t(m, &L.fc1_c2, 4 * d, s));
        const std::string p = "decoder.blocks." + std::to_string(i);
        attn_regs(p, L.wqkv, L.bqkv, L.wqkv + (size_t)d * d, L.wqkv + (size_t)2 * d * d, L.bqkv + 2 * d, L.wo,
                  L.bo, L.ln1_g, L.ln1_b, "attn", WL_TILED, WL_TILED, WL_TILED);
        // cross-attention key/value matrices feed the big GEMM (row-major); query/out feed the GEMV
        attn_regs(p, L.wxq, L.bxq, L.wxkv, L.wxkv + (size_t)d * d, L.bxkv + d, L.wxo, L.bxo, L.lnx_g, L.lnx_b,
                  "cross_attn", WL_TILED, WL_PLAIN, WL_TILED);
        mlp_regs(p, L.w1, L.b1, L.w2, L.b2, L.ln2_g, L.ln2_b, WL_TILED);
    }
    WM_TRY(dalloc_t(m, &m->ln_g, d, s)); WM_TRY(dalloc_t(m, &m->ln_b, d, s));
    reg(m, "decoder.ln.weight", m->ln_g, false, d, 2);
    reg(m, "decoder.ln.bias", m->ln_b, false, d, 3);
    WM_TRY(dalloc_t(m, &m->emb_f, (size_t)m->vpad * d, s));
    WM_TRY(dalloc_t(m, &m->logit_c1, m->vpad, s)); WM_TRY(dalloc_t(m, &m->logit_c2, m->vpad, s));

    WM_TRY(alloc_decode_buffers(m, s));
    WM_HIP(hipStreamSynchronize(s));
    return WM_OK;
}

// A second context on the same device that SHARES the parent's (read-only) weights and owns its
// own stream, activations, KV caches and decode graph: lets independent batches overlap on one GPU
// (the decode chain is latency-bound, so concurrent batches fill the idle HBM bandwidth).
int wm_model_clone(wm_ctx *child, const wm_ctx *parent) {
    const WmModel *pm = parent->model;
    WM_REQUIRE(pm && pm->finalized, WM_ERR_STATE, "clone: the parent's weights must be finalised");
    WmModel *m = new WmModel();
    child->model = m;
    m->dims = pm->dims; m->finalized = true; m->k1pad = pm->k1pad; m->vpad = pm->vpad;
    m->conv1_w = pm->conv1_w; m->conv2_w = pm->conv2_w; m->conv1_b = pm->conv1_b; m->conv2_b = pm->conv2_b;
    m->enc_pos = pm->enc_pos; m->enc = pm->enc; m->ln_post_g = pm->ln_post_g; m->ln_post_b = pm->ln_post_b;
    m->tok_emb = pm->tok_emb; m->dec_pos = pm->dec_pos; m->dec = pm->dec; m->ln_g = pm->ln_g; m->ln_b = pm->ln_b;
    m->emb_f = pm->emb_f; m->logit_c1 = pm->logit_c1; m->logit_c2 = pm->logit_c2;
    m->tensors = pm->tensors; m->index = pm->index;   // registry for wm_get_tensor (pointers alias the parent)
    m->shares_weights = true;
    WM_TRY(alloc_decode_buffers(m, child->stream));
    WM_HIP(hipMemcpyAsync(m->dmask, pm->dmask, (size_t)2 * (m->vpad / 32) * 4, hipMemcpyDeviceToDevice, child->stream));
    m->mask_on = pm->mask_on; m->mask_host = pm->mask_host;
    m->ts_on = pm->ts_on; m->ts_begin = pm->ts_begin; m->ts_eot = pm->ts_eot; m->ts_max_initial = pm->ts_max_initial;
    WM_HIP(hipStreamSynchronize(child->stream));
    return WM_OK;
}

void wm_model_destroy(wm_ctx *ctx) {
    WmModel *m = ctx->model;
    if (!m) return;
    wm_model_drop_graphs(m);
    if (m->h_nlive) (void)hipHostFree(m->h_nlive);
    for (void *p : m->allocs) (void)hipFree(p);
    if (m->pcm_stage) (void)hipFree(m->pcm_stage);
    if (m->io_stage) (void)hipFree(m->io_stage);
    delete m;
    ctx->model = nullptr;
}

// ------------------------------------------------------------------ weights ------------
int wm_model_set_tensor(wm_ctx *ctx, const char *name, const float *data, size_t n) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m, WM_ERR_STATE, "context has no model");
    WM_REQUIRE(!m->shares_weights, WM_ERR_STATE, "a cloned context shares its parent's weights: set them on the parent");
    WM_REQUIRE(name && data, WM_ERR_INVALID, "null name / data");
    auto it = m->index.find(name);
    WM_REQUIRE(it != m->index.end(), WM_ERR_INVALID, "unknown tensor '%s'", name);
    WmTensor &t = m->tensors[it->second];
    WM_REQUIRE(n == t.n_elems, WM_ERR_INVALID, "tensor '%s': %zu elements given, %zu expected", name, n, t.n_elems);
    if (!t.is_bf16) {
        WM_HIP(hipMemcpy(t.ptr, data, n * sizeof(float), hipMemcpyHostToDevice));
    } else if (t.layout == WL_PLAIN) {
        std::vector<bf16_t> tmp(n);
        for (size_t i = 0; i < n; ++i) tmp[i] = host_f2bf(data[i]);
        WM_HIP(hipMemcpy(t.ptr, tmp.data(), n * sizeof(bf16_t), hipMemcpyHostToDevice));
    } else if (t.layout == WL_TILED) {
        const size_t K = (size_t)t.conv_kpad, N = n / K;
        const size_t Np = (N + 15) / 16 * 16;
        std::vector<bf16_t> tmp(Np * K, 0);
        for (size_t r = 0; r < N; ++r)
            for (size_t k = 0; k < K; ++k) tmp[wm_tiled_offset(r, k, K)] = host_f2bf(data[r * K + k]);
        WM_HIP(hipMemcpy(t.ptr, tmp.data(), tmp.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    } else {  // WL_CONV: [O][C][3] -> [O][kpad], k = tap*C + c
        const size_t per = (size_t)t.conv_c * 3, O = n / per;
        std::vector<bf16_t> tmp(O * t.conv_kpad, 0);
        for (size_t o = 0; o < O; ++o)
            for (int c = 0; c < t.conv_c; ++c)
                for (int tap = 0; tap < 3; ++tap)
                    tmp[o * t.conv_kpad + (size_t)tap * t.conv_c + c] = host_f2bf(data[o * per + (size_t)c * 3 + tap]);
        WM_HIP(hipMemcpy(t.ptr, tmp.data(), tmp.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    }
    t.set = true;
    m->finalized = false;
    return WM_OK;
}

int wm_model_get_tensor(wm_ctx *ctx, const char *name, float *data, size_t n) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m, WM_ERR_STATE, "context has no model");
    WM_REQUIRE(name && data, WM_ERR_INVALID, "null name / data");
    auto it = m->index.find(name);
    WM_REQUIRE(it != m->index.end(), WM_ERR_INVALID, "unknown tensor '%s'", name);
    const WmTensor &t = m->tensors[it->second];
    WM_REQUIRE(n == t.n_elems, WM_ERR_INVALID, "tensor '%s': %zu elements asked, %zu stored", name, n, t.n_elems);
    WM_HIP(hipStreamSynchronize(ctx->stream));
    if (!t.is_bf16) {
        WM_HIP(hipMemcpy(data, t.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
    } else if (t.layout == WL_PLAIN) {
        std::vector<bf16_t> tmp(n);
        WM_HIP(hipMemcpy(tmp.data(), t.ptr, n * sizeof(bf16_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) data[i] = host_bf2f(tmp[i]);
    } else if (t.layout == WL_TILED) {
        const size_t K = (size_t)t.conv_kpad, N = n / K;
        const size_t Np = (N + 15) / 16 * 16;
        std::vector<bf16_t> tmp(Np * K);
        WM_HIP(hipMemcpy(tmp.data(), t.ptr, tmp.size() * sizeof(bf16_t), hipMemcpyDeviceToHost));
        for (size_t r = 0; r < N; ++r)
            for (size_t k = 0; k < K; ++k) data[r * K + k] = host_bf2f(tmp[wm_tiled_offset(r, k, K)]);
    } else {
        const size_t per = (size_t)t.conv_c * 3, O = n / per;
        std::vector<bf16_t> tmp(O * t.conv_kpad);
        WM_HIP(hipMemcpy(tmp.data(), t.ptr, tmp.size() * sizeof(bf16_t), hipMemcpyDeviceToHost));
        for (size_t o = 0; o < O; ++o)
            for (int c = 0; c < t.conv_c; ++c)
                for (int tap = 0; tap < 3; ++tap)
                    data[o * per + (size_t)c * 3 + tap] = host_bf2f(tmp[o * t.conv_kpad + (size_t)tap * t.conv_c + c]);
    }
    return WM_OK;
}

int wm_model_init_synthetic(wm_ctx *ctx, uint64_t seed, float matrix_gain) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m, WM_ERR_STATE, "context has no model");
    WM_REQUIRE(!m->shares_weights, WM_ERR_STATE, "a cloned context shares its parent's weights");
    const int d = m->dims.n_audio_state;
    for (size_t i = 0; i < m->tensors.size(); ++i) {
        WmTensor &t = m->tensors[i];
        if (t.kind == 4) {  // openai-whisper sinusoids(): fixed encoder positions
            const int L = m->dims.n_audio_ctx, half = d / 2;
            std::vector<float> pe((size_t)L * d);
            const double inc = log(10000.0) / (half - 1);
            for (int p = 0; p < L; ++p)
                for (int j = 0; j < half; ++j) {
                    const double a = (double)p * exp(-inc * j);
                    pe[(size_t)p * d + j] = (float)sin(a);
                    pe[(size_t)p * d + half + j] = (float)cos(a);
                }
            WM_HIP(hipMemcpy(t.ptr, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
        } else {
            // matrix_gain scales the MATRICES only (weights.synthetic_state_dict(..., matrix_gain)): not the learned
            // positional table, biases or LayerNorm parameters -- the "lively" recipe of the token-level tests / bench.py
            const bool scaled = t.kind == 0 && t.name.find("positional") == std::string::npos;
            WM_TRY(wm_fill_synthetic(ctx, t, (uint32_t)(seed & 0xffffffffu), (int)i, scaled ? matrix_gain : 1.0f));
        }
        t.set = true;
    }
    WM_HIP(hipStreamSynchronize(ctx->stream));
    m->finalized = false;
    return WM_OK;
}

int wm_model_finalize(wm_ctx *ctx) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m, WM_ERR_STATE, "context has no model");
    for (const WmTensor &t : m->tensors)
        WM_REQUIRE(t.set, WM_ERR_STATE, "tensor '%s' was never set", t.name.c_str());
    // QKV fusion and conv-tap permutation happen at set time (tensors are written straight into their fused / permuted
    // HBM locations).  What is left: the LayerNorm-folded copies the decode GEMVs multiply (model.h, DecLayerW).
    const int d = m->dims.n_text_state;
    for (DecLayerW &L : m->dec) {
        WM_TRY(wm_ln_fold(ctx, L.wqkv, L.ln1_g, L.ln1_b, L.bqkv, 3 * d, d, L.wqkv_f, L.qkv_c1, L.qkv_c2));
        WM_TRY(wm_ln_fold(ctx, L.wxq, L.lnx_g, L.lnx_b, L.bxq, d, d, L.wxq_f, L.xq_c1, L.xq_c2));
        WM_TRY(wm_ln_fold(ctx, L.w1, L.ln2_g, L.ln2_b, L.b1, 4 * d, d, L.w1_f, L.fc1_c1, L.fc1_c2));
    }
    WM_TRY(wm_ln_fold(ctx, m->tok_emb, m->ln_g, m->ln_b, nullptr, m->dims.n_vocab, d, m->emb_f, m->logit_c1, m->logit_c2));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    m->finalized = true;
    return WM_OK;
}

// ------------------------------------------------------------------ activations --------
int wm_model_reserve(wm_ctx *ctx, int B) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m, WM_ERR_STATE, "context has no model");
    if (B <= m->cap_b) return WM_OK;
    WM_HIP(hipStreamSynchronize(ctx->stream));
    void *olds[] = {m->mel_t, m->h1p, m->x, m->xn, m->qk, m->vt, m->att, m->hid, m->xa_f32, m->mel_f32, m->xkv, m->skv};
    for (void *p : olds)
        if (p) {
            (void)hipFree(p);
            for (auto &a : m->allocs)
                if (a == p) a = nullptr;
        }
    const wm_dims &D = m->dims;
    const int d = D.n_audio_state, H = D.n_audio_head;
    const size_t M = (size_t)B * 1500;
    hipStream_t s = ctx->stream;
    WM_TRY(dalloc_t(m, &m->mel_t, (size_t)B * 3002 * D.n_mels, s));
    WM_TRY(dalloc_t(m, &m->h1p, (size_t)B * 3001 * d, s));
    WM_TRY(dalloc_t(m, &m->x, M * d, s));
    WM_TRY(dalloc_t(m, &m->xn, M * d, s));
    WM_TRY(dalloc_t(m, &m->qk, (M + 64) * 2 * d, s));
    WM_TRY(dalloc_t(m, &m->vt, (size_t)B * H * 64 * 1536, s));
    WM_TRY(dalloc_t(m, &m->att, M * d, s));
    WM_TRY(dalloc_t(m, &m->hid, M * 4 * d, s));
    WM_TRY(dalloc_t(m, &m->xa_f32, M * d, s));
    WM_TRY(dalloc_t(m, &m->mel_f32, (size_t)B * D.n_mels * WM_N_FRAMES, s));
    WM_TRY(dalloc_t(m, &m->xkv, (size_t)D.n_text_layer * 2 * B * H * 1500 * 64, s));
    const int Bd = B < WM_DEC_MAXB ? B : WM_DEC_MAXB;
    WM_TRY(dalloc_t(m, &m->skv, (size_t)D.n_text_layer * 2 * Bd * H * D.n_text_ctx * 64, s));
    m->cap_b = B;
    WM_HIP(hipStreamSynchronize(s));
    return WM_OK;
}

// ------------------------------------------------------------------ encoder ------------
static GemmArgs plain_gemm(const bf16_t *A, int lda, const bf16_t *W, const float *bias, void *C, int ldc,
                           int M, int N, int K, int epi) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.a_rpb = (long)M + 1; g.a_bstride = 0; g.a_rstride = lda;
    g.W = W; g.bias = bias; g.C = C;
    g.c_rpb = (long)M + 1; g.c_bstride = 0; g.c_rstride = ldc;
    g.M = M; g.N = N; g.K = K; g.epi = epi;
    return g;
}

int wm_model_encode_dev(wm_ctx *ctx, const float *d_mel, int B, float *d_xa_out) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m && m->finalized, WM_ERR_STATE, "model weights not finalised (wm_finalize)");
    WM_REQUIRE(B >= 1, WM_ERR_INVALID, "B must be >= 1");
    WM_TRY(wm_model_reserve(ctx, B));
    const wm_dims &D = m->dims;
    const int d = D.n_audio_state, H = D.n_audio_head, S = 1500, C = D.n_mels;
    const int M = B * S;
    // mel [B][C][3000] f32 -> time-major bf16 with zero edge rows (conv padding = 1)
    WM_TRY(wm_mel_to_time_major(ctx, d_mel, B, C, m->mel_t));
    {   // conv1 (k3, p1) + GELU as an implicit GEMM: window of frame t = mel_t[b][t..t+2][:]
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = m->mel_t; g.a_rpb = 3000; g.a_bstride = 3002L * C; g.a_rstride = C;
        g.W = m->conv1_w; g.bias = m->conv1_b;
        g.C = m->h1p + d; g.c_rpb = 3000; g.c_bstride = 3001L * d; g.c_rstride = d;
        g.M = B * 3000; g.N = d; g.K = m->k1pad; g.epi = EPI_GELU_BF16;
        WM_TRY(wm_gemm(ctx, g));
    }
    {   // conv2 (k3, s2, p1) + GELU + positional embedding: window of frame s = h1p[b][2s..2s+2][:]
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = m->h1p; g.a_rpb = S; g.a_bstride = 3001L * d; g.a_rstride = 2L * d;
        g.W = m->conv2_w; g.bias = m->conv2_b;
        g.C = m->x; g.c_rpb = S; g.c_bstride = (long)S * d; g.c_rstride = d;
        g.M = M; g.N = d; g.K = 3 * d; g.epi = EPI_CONV2_F32; g.pos = m->enc_pos;
        WM_TRY(wm_gemm(ctx, g));
    }
    for (int i = 0; i < D.n_audio_layer; ++i) {
        const EncLayerW &L = m->enc[i];
        WM_TRY(wm_layernorm(ctx, m->x, L.ln1_g, L.ln1_b, M, d, m->xn, nullptr));
        GemmArgs g = plain_gemm(m->xn, d, L.wqkv, L.bqkv, m->qk, 2 * d, M, 3 * d, d, EPI_QKV_ENC);
        g.vt = m->vt; g.d_model = d; g.n_head = H; g.seq = S; g.seq_pad = 1536; g.batch = B;
        WM_TRY(wm_gemm(ctx, g));
        WM_TRY(wm_enc_attention(ctx, m->qk, m->vt, m->att, B, H, S, 1536, d));
        WM_TRY(wm_gemm(ctx, plain_gemm(m->att, d, L.wo, L.bo, m->x, d, M, d, d, EPI_RESID_F32)));
        WM_TRY(wm_layernorm(ctx, m->x, L.ln2_g, L.ln2_b, M, d, m->xn, nullptr));
        WM_TRY(wm_gemm(ctx, plain_gemm(m->xn, d, L.w1, L.b1, m->hid, 4 * d, M, 4 * d, d, EPI_GELU_BF16)));
        WM_TRY(wm_gemm(ctx, plain_gemm(m->hid, 4 * d, L.w2, L.b2, m->x, d, M, d, 4 * d, EPI_RESID_F32)));
    }
    WM_TRY(wm_layernorm(ctx, m->x, m->ln_post_g, m->ln_post_b, M, d, m->xn, d_xa_out ? d_xa_out : m->xa_f32));
    return WM_OK;
}

int wm_model_set_xa(wm_ctx *ctx, const float *d_xa, int B) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m && m->finalized, WM_ERR_STATE, "model weights not finalised (wm_finalize)");
    WM_TRY(wm_model_reserve(ctx, B));
    return wm_f32_to_bf16(ctx, d_xa, m->xn, (size_t)B * 1500 * m->dims.n_audio_state);
}

// Cross-attention K/V of the encoder output, once per chunk (the reference's exported decoder
// recomputes these inside every call -- SURVEY.md 8a row a15; caching them is the point of K9).
int wm_model_cross_kv(wm_ctx *ctx, int B) {
    WmModel *m = ctx->model;
    const wm_dims &D = m->dims;
    const int d = D.n_text_state, H = D.n_text_head, S = 1500, M = B * S;
    for (int l = 0; l < D.n_text_layer; ++l) {
        const DecLayerW &L = m->dec[l];
        GemmArgs g = plain_gemm(m->xn, d, L.wxkv, L.bxkv, m->xkv + (size_t)l * 2 * B * H * S * 64, 0, M, 2 * d, d,
                                EPI_XKV);
        g.d_model = d; g.n_head = H; g.seq = S; g.batch = B;
        WM_TRY(wm_gemm(ctx, g));
    }
    return WM_OK;
}

// ------------------------------------------------------------------ decoder ------------
int wm_model_decode_begin(wm_ctx *ctx, int B) {
    WmModel *m = ctx->model;
    WM_REQUIRE(m && m->finalized, WM_ERR_STATE, "model weights not finalised (wm_finalize)");
    WM_REQUIRE(B >= 1 && B <= WM_DEC_MAXB, WM_ERR_INVALID, "decode batch must be 1..%d", WM_DEC_MAXB);
    // the arrival counter of the arg-max workgroups is zero between launches; a decode that was abandoned half way
    // (an error in the middle of a step) must not leave the next one with a stale count
    WM_HIP(hipMemsetAsync(m->darrive, 0, sizeof(int), ctx->stream));
    m->stop_on = false;   // wm_transcribe_greedy switches it on for its own decode (lane_prefill)
    m->xattn_shared = false;   // ... and decides whether the group shares the chip
    return WM_OK;
}

int wm_model_set_pos(wm_ctx *ctx, int pos) {
    WmModel *m = ctx->model;
    if (pos == 0) {
        WM_HIP(hipMemsetAsync(m->dpos, 0, sizeof(int), ctx->stream));  // no host operand: stays asynchronous
        return WM_OK;
    }
    WM_HIP(hipMemcpyAsync(m->dpos, &pos, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));  // `pos` is a stack variable
    return WM_OK;
}

int wm_model_decode_step(wm_ctx *ctx, int B, bool want_logits, int arg_first, int arg_last, int mask_first_pos, bool use_ts) {
    WmModel *m = ctx->model;
    const wm_dims &D = m->dims;
    const int d = D.n_text_state, H = D.n_text_head, T = D.n_text_ctx, S = 1500;
    const int ns = wm_dec_attn_splits(B, H);
    const int xns = g_wm_tuning.xattn_splits > 0 ? g_wm_tuning.xattn_splits : ns;   // 0 in the product
    const int *live = m->stop_on ? m->dlive : nullptr, *nlive = m->stop_on ? m->dnlive : nullptr;
    // mean-centring offsets of the bf16 residual copy: the embedding wrote buffer 0; every LayerNorm-folded GEMV reads
    // the current buffer and leaves the new means in the other one
    int cur = 0;
    auto mean_buf = [&](int i) { return m->dmean + (size_t)i * WM_DEC_MAXB; };
    for (int l = 0; l < D.n_text_layer; ++l) {
        const DecLayerW &L = m->dec[l];
        bf16_t *kc = m->skv + (size_t)(l * 2 + 0) * B * H * T * 64;
        bf16_t *vc = m->skv + (size_t)(l * 2 + 1) * B * H * T * 64;
        const bf16_t *xk = m->xkv + (size_t)(l * 2 + 0) * B * H * S * 64;
        const bf16_t *xv = m->xkv + (size_t)(l * 2 + 1) * B * H * S * 64;
        DecGemvArgs a;
        // 1. attn_ln (folded) + fused q|k|v projection; k, v appended to the self-attention cache
        memset(&a, 0, sizeof(a));
        a.epi = DE_QKV; a.B = B; a.N = 3 * d; a.K = d; a.W = L.wqkv_f; a.c1 = L.qkv_c1; a.c2 = L.qkv_c2;
        a.a = m->dxb; a.out_f32 = m->dq; a.kcache = kc; a.vcache = vc;
        a.stats_in = m->dstats;
        a.mean_in = mean_buf(cur); a.mean_out = mean_buf(cur ^ 1); cur ^= 1;
        a.pos_ptr = m->dpos; a.n_ctx = T; a.n_head = H;
        WM_TRY(wm_dec_gemv(ctx, a));
        // 2. causal self-attention over positions 0..pos
        WM_TRY(wm_dec_self_attention(ctx, m->dq, kc, vc, B, H, T, 0, m->dpos, m->datt, L.wo, d, d, live, nlive));
        // 3. out-projection + residual (f32 stream, its bf16 copy, partial statistics)
        memset(&a, 0, sizeof(a));
        a.epi = DE_RESID; a.B = B; a.N = d; a.K = d; a.W = L.wo; a.c2 = L.bo;
        a.a = m->datt; a.out_f32 = m->dx; a.out_bf16 = m->dxb; a.ldo = d; a.stats_out = m->dstats;
        a.mean_in = mean_buf(cur);
        a.pf_ptr = L.wxq_f; a.pf_rows = d; a.pf_k = d;
        const bool xshort = m->xattn_shared && !g_wm_tuning.xattn_never_short;
        const bool fuse_q = xns == 1 && wm_dec_xattn_fq_applies(B, H, d, xshort);
        a.pf_head_major = fuse_q ? (H * B + 7) / 8 : 0;   // pairs per XCD of the fused consumer
        WM_TRY(wm_dec_gemv(ctx, a));
        // 4. cross_attn_ln (folded) + query projection
        memset(&a, 0, sizeof(a));
        a.epi = DE_Q; a.B = B; a.N = d; a.K = d; a.W = L.wxq_f; a.c1 = L.xq_c1; a.c2 = L.xq_c2;
        a.a = m->dxb; a.out_f32 = m->dq; a.ldo = d;
        a.stats_in = m->dstats;
        a.mean_in = mean_buf(cur); a.mean_out = mean_buf(cur ^ 1); cur ^= 1;
        if (fuse_q) {
            // 4 + 5 as ONE launch (the latency shape: every pair's workgroup forms its own query, dec_kernels.hip).  Differs from
            // the two launches for FINISHED rows only (early stop): their block's means stay stale, m->dq is not written
            WM_TRY(wm_dec_xattn_fq(ctx, a, xk, xv, B, H, S, S, m->datt, live, nlive, L.wxo, d, d));
        } else {
            WM_TRY(wm_dec_gemv(ctx, a));
            // 5. cross-attention over the 1500 cached encoder frames
            WM_TRY(wm_dec_attention(ctx, m->dq, xk, xv, B, H, S, S, nullptr, xns, m->dpart, m->datt, true, L.wxo, d, d, live, nlive, xshort));
        }
        // 6. out-projection + residual
        memset(&a, 0, sizeof(a));
        a.epi = DE_RESID; a.B = B; a.N = d; a.K = d; a.W = L.wxo; a.c2 = L.bxo;
        a.a = m->datt; a.out_f32 = m->dx; a.out_bf16 = m->dxb; a.ldo = d; a.stats_out = m->dstats;
        a.mean_in = mean_buf(cur);
        a.pf_ptr = L.w1_f; a.pf_rows = 4 * d; a.pf_k = d;
        WM_TRY(wm_dec_gemv(ctx, a));
        // 7. mlp_ln (folded) + fc1 + GELU
        memset(&a, 0, sizeof(a));
        a.epi = DE_GELU; a.B = B; a.N = 4 * d; a.K = d; a.W = L.w1_f; a.c1 = L.fc1_c1; a.c2 = L.fc1_c2;
        a.a = m->dxb; a.out_bf16 = m->dhid; a.ldo = 4 * d;
        a.stats_in = m->dstats;
        a.mean_in = mean_buf(cur); a.mean_out = mean_buf(cur ^ 1); cur ^= 1;
        a.pf_ptr = L.w2; a.pf_rows = d; a.pf_k = 4 * d;
        WM_TRY(wm_dec_gemv(ctx, a));
        // 8. fc2 + residual
        memset(&a, 0, sizeof(a));
        a.epi = DE_RESID; a.B = B; a.N = d; a.K = 4 * d; a.W = L.w2; a.c2 = L.b2;
        a.a = m->dhid; a.out_f32 = m->dx; a.out_bf16 = m->dxb; a.ldo = d; a.stats_out = m->dstats;
        a.mean_in = mean_buf(cur);
        if (l + 1 < D.n_text_layer) { a.pf_ptr = m->dec[l + 1].wqkv_f; a.pf_rows = 3 * d; a.pf_k = d; }
        WM_TRY(wm_dec_gemv(ctx, a));
    }
    {   // final LayerNorm (folded) + tied-embedding logits + fused per-tile arg-max
        DecGemvArgs a;
        memset(&a, 0, sizeof(a));
        a.epi = DE_LOGITS; a.B = B; a.N = D.n_vocab; a.K = d; a.W = m->emb_f; a.c1 = m->logit_c1; a.c2 = m->logit_c2;
        a.a = m->dxb; a.stats_in = m->dstats;
        a.mean_in = mean_buf(cur); a.mean_out = mean_buf(cur ^ 1);
        a.out_f32 = want_logits ? m->dlogits : nullptr; a.ldo = m->vpad;
        a.argmax = m->dargmax; a.arg_first = arg_first; a.arg_last = arg_last;
        if (mask_first_pos >= 0) {
            a.mask = m->dmask; a.mask_words = m->vpad / 32; a.mask_first_pos = mask_first_pos; a.pos_ptr = m->dpos;
        }
        if (use_ts) a.ts = wm_model_ts_dev(m);
        WM_TRY(wm_dec_gemv(ctx, a));
    }
    return WM_OK;
}

int wm_model_embed_first(wm_ctx *ctx, int B) {
    WmModel *m = ctx->model;
    return wm_dec_embed(ctx, m->dseq, m->dpos, B, m->tok_emb, m->dec_pos, m->dims.n_text_state, m->dx, m->dxb, m->dstats, m->dmean);
}

int wm_model_close_step(wm_ctx *ctx, int B, int n_prompt, bool write_seq, int *result, int arg_first, bool use_ts) {
    WmModel *m = ctx->model;
    const WmTsDev t = wm_model_ts_dev(m);
    const WmStopDev sp = wm_model_stop_dev(m);
    return wm_argmax_embed(ctx, m->dargmax, m->vpad / 16, B, write_seq ? m->dseq : nullptr, m->dpos, n_prompt, result,
                           arg_first, m->tok_emb, m->dec_pos, m->dims.n_text_state, m->dims.n_text_ctx, m->dx, m->dxb,
                           m->dstats, use_ts ? &t : nullptr, m->darrive, use_ts ? m->ts_eot : arg_first, m->dmean,
                           m->stop_on ? &sp : nullptr);
}
