// model_api.cpp -- boundary #2 of the C ABI (include/whisper_mi355x.h): weight loading, the
// encoder / decoder entry points that replace the CoreML `encoder` / `decoder` classes
// (Whisper/Whisper/Whisper.swift:17-40), the KV-cached greedy transcription asked for by
// BASELINE.json.  (The per-kernel test hooks live in debug_hooks.cpp, which is NOT part of the product library.)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "model.h"

#define WM_MODEL(ctx)                                                               \
    WM_TRY(wm_ctx_make_current(ctx));                                               \
    WmModel *m = (ctx)->model;                                                      \
    WM_REQUIRE(m != nullptr, WM_ERR_STATE, "context was created without a model (use wm_create)")

static int io_stage(wm_ctx *ctx, size_t bytes, char **out) {
    WmModel *m = ctx->model;
    if (m->io_stage_bytes < bytes) {
        WM_HIP(hipStreamSynchronize(ctx->stream));
        if (m->io_stage) WM_HIP(hipFree(m->io_stage));
        m->io_stage = nullptr;
        m->io_stage_bytes = 0;
        WM_HIP(hipMalloc((void **)&m->io_stage, bytes));
        m->io_stage_bytes = bytes;
    }
    *out = (char *)m->io_stage;
    return WM_OK;
}

extern "C" int wm_set_tensor(wm_ctx *ctx, const char *name, const float *data, size_t n) try {
    WM_MODEL(ctx);
    (void)m;
    return wm_model_set_tensor(ctx, name, data, n);
} WM_API_CATCH
extern "C" int wm_get_tensor(wm_ctx *ctx, const char *name, float *data, size_t n) try {
    WM_MODEL(ctx);
    (void)m;
    return wm_model_get_tensor(ctx, name, data, n);
} WM_API_CATCH
extern "C" int wm_init_synthetic(wm_ctx *ctx, uint64_t seed) try {
    WM_MODEL(ctx);
    (void)m;
    return wm_model_init_synthetic(ctx, seed, 1.0f);
} WM_API_CATCH
extern "C" int wm_init_synthetic_gain(wm_ctx *ctx, uint64_t seed, float matrix_gain) try {
    WM_MODEL(ctx);
    (void)m;
    WM_REQUIRE(matrix_gain > 0.f && matrix_gain <= 64.f, WM_ERR_INVALID, "init_synthetic_gain: gain must be in (0, 64]");
    return wm_model_init_synthetic(ctx, seed, matrix_gain);
} WM_API_CATCH
extern "C" int wm_finalize(wm_ctx *ctx) try {
    WM_MODEL(ctx);
    (void)m;
    return wm_model_finalize(ctx);
} WM_API_CATCH
extern "C" int wm_set_suppress(wm_ctx *ctx, const int32_t *suppress, int n, const int32_t *suppress_first, int n_first) try {
    WM_MODEL(ctx);
    (void)m;
    WM_TRY(wm_model_set_suppress(ctx, suppress, n, suppress_first, n_first));
    for (wm_ctx *lane : ctx->lanes) WM_TRY(wm_model_set_suppress(lane, suppress, n, suppress_first, n_first));
    for (auto &v : ctx->part_lanes)
        for (wm_ctx *lane : v) WM_TRY(wm_model_set_suppress(lane, suppress, n, suppress_first, n_first));
    for (auto &kv : ctx->solo_lanes) WM_TRY(wm_model_set_suppress(kv.second, suppress, n, suppress_first, n_first));
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_set_timestamp_rules(wm_ctx *ctx, int enable, int32_t timestamp_begin, int32_t eot,
                                      int32_t max_initial_timestamp_index) try {
    WM_MODEL(ctx);
    (void)m;
    WM_TRY(wm_model_set_timestamp_rules(ctx, enable, timestamp_begin, eot, max_initial_timestamp_index));
    for (wm_ctx *lane : ctx->lanes)
        WM_TRY(wm_model_set_timestamp_rules(lane, enable, timestamp_begin, eot, max_initial_timestamp_index));
    for (auto &v : ctx->part_lanes)
        for (wm_ctx *lane : v)
            WM_TRY(wm_model_set_timestamp_rules(lane, enable, timestamp_begin, eot, max_initial_timestamp_index));
    for (auto &kv : ctx->solo_lanes)
        WM_TRY(wm_model_set_timestamp_rules(kv.second, enable, timestamp_begin, eot, max_initial_timestamp_index));
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_set_lanes(wm_ctx *ctx, int n_lanes) try {
    WM_REQUIRE(ctx, WM_ERR_INVALID, "null context");
    WM_REQUIRE(n_lanes >= 0 && n_lanes <= 8, WM_ERR_INVALID, "set_lanes: 0 (default) .. 8");
    ctx->max_lanes = n_lanes;
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_get_dims(const wm_ctx *ctx, wm_dims *out) try {
    WM_REQUIRE(ctx && out, WM_ERR_INVALID, "null pointer");
    WM_REQUIRE(ctx->model, WM_ERR_STATE, "context has no model");
    *out = ctx->model->dims;
    return WM_OK;
} WM_API_CATCH

// Flat weight file (format: the docstring of openai-whisper-coreml_amd/weights.py).
extern "C" int wm_load_weights(wm_ctx *ctx, const char *path) try {
    WM_MODEL(ctx);
    WM_REQUIRE(path, WM_ERR_INVALID, "null path");
    FILE *f = fopen(path, "rb");
    WM_REQUIRE(f, WM_ERR_IO, "cannot open '%s'", path);
    int st = WM_OK;
    char magic[8];
    int32_t dims[10], count = 0;
    std::vector<float> buf;
    std::vector<char> name;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "WMI355X1", 8) != 0 || fread(dims, 4, 10, f) != 10 ||
        fread(&count, 4, 1, f) != 1) {
        wm_set_error("'%s' is not a WMI355X1 weight file", path);
        st = WM_ERR_IO;
    }
    if (st == WM_OK && memcmp(dims, &m->dims, sizeof(wm_dims)) != 0) {
        wm_set_error("'%s': model dimensions differ from the context's", path);
        st = WM_ERR_IO;
    }
    if (st == WM_OK && count <= 0) {
        wm_set_error("'%s': tensor count %d", path, count);
        st = WM_ERR_IO;
    }
    for (int i = 0; st == WM_OK && i < count; ++i) {
        int32_t nl = 0;
        int64_t ne = 0;
        if (fread(&nl, 4, 1, f) != 1 || nl <= 0 || nl > 4096) { wm_set_error("'%s': truncated", path); st = WM_ERR_IO; break; }
        name.assign(nl + 1, 0);
        if (fread(name.data(), 1, nl, f) != (size_t)nl || fread(&ne, 8, 1, f) != 1 || ne <= 0) {
            wm_set_error("'%s': truncated", path); st = WM_ERR_IO; break;
        }
        // the element count comes from the file: check it against the registered tensor BEFORE allocating
        auto it = m->index.find(name.data());
        if (it == m->index.end()) { wm_set_error("'%s': unknown tensor '%s'", path, name.data()); st = WM_ERR_IO; break; }
        if ((uint64_t)ne != (uint64_t)m->tensors[it->second].n_elems) {
            wm_set_error("'%s': tensor '%s' has %lld elements, %zu expected", path, name.data(), (long long)ne,
                         m->tensors[it->second].n_elems);
            st = WM_ERR_IO;
            break;
        }
        buf.resize((size_t)ne);
        if (fread(buf.data(), 4, (size_t)ne, f) != (size_t)ne) { wm_set_error("'%s': truncated", path); st = WM_ERR_IO; break; }
        st = wm_model_set_tensor(ctx, name.data(), buf.data(), (size_t)ne);
    }
    fclose(f);
    return st;
} WM_API_CATCH

// ------------------------------------------------------------------ encoder ------------
extern "C" int wm_encode(wm_ctx *ctx, const float *mel, int B, float *xa, wm_mem mem) try {
    WM_MODEL(ctx);
    WM_REQUIRE(mel && xa && B >= 1, WM_ERR_INVALID, "null pointer / B < 1");
    const auto encode_dev = ctx->dbg_hooks ? ctx->dbg_hooks->encode_dev : wm_model_encode_dev;   // null hooks in the product
    if (mem == WM_MEM_DEVICE) return encode_dev(ctx, mel, B, xa);
    const size_t in_b = (size_t)B * m->dims.n_mels * WM_N_FRAMES * 4;
    const size_t out_b = (size_t)B * 1500 * m->dims.n_audio_state * 4;
    char *st;
    WM_TRY(io_stage(ctx, in_b + out_b, &st));
    WM_HIP(hipMemcpyAsync(st, mel, in_b, hipMemcpyHostToDevice, ctx->stream));
    WM_TRY(encode_dev(ctx, (const float *)st, B, (float *)(st + in_b)));
    WM_HIP(hipMemcpyAsync(xa, st + in_b, out_b, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    return WM_OK;
} WM_API_CATCH

// ------------------------------------------------------------------ decoder ------------
// Shared: bring xa (f32 [B][1500][d], host or device) into the bf16 encoder-output buffer
// and build the cross-attention K/V cache.
static int load_xa(wm_ctx *ctx, const float *xa, int B, wm_mem mem) {
    WmModel *m = ctx->model;
    const size_t bytes = (size_t)B * 1500 * m->dims.n_audio_state * 4;
    const float *d_xa = xa;
    if (mem == WM_MEM_HOST) {
        char *st;
        WM_TRY(io_stage(ctx, bytes, &st));
        WM_HIP(hipMemcpyAsync(st, xa, bytes, hipMemcpyHostToDevice, ctx->stream));
        d_xa = (const float *)st;
    }
    WM_TRY(wm_model_set_xa(ctx, d_xa, B));
    return wm_model_cross_kv(ctx, B);
}

extern "C" int wm_decode_logits(wm_ctx *ctx, const int32_t *tokens, int B, int T, const float *xa,
                                float *logits, wm_mem mem) try {
    WM_MODEL(ctx);
    WM_REQUIRE(tokens && xa && logits, WM_ERR_INVALID, "null pointer");
    WM_REQUIRE(B >= 1 && B <= WM_DEC_MAXB, WM_ERR_INVALID, "B must be 1..%d", WM_DEC_MAXB);
    WM_REQUIRE(T >= 1 && T <= m->dims.n_text_ctx, WM_ERR_INVALID, "T must be 1..%d", m->dims.n_text_ctx);
    const int V = m->dims.n_vocab;
    // tokens arrive [B][T]; the step kernel wants the B tokens of one position contiguous
    std::vector<int32_t> host_tok((size_t)B * T);
    if (mem == WM_MEM_DEVICE) {
        WM_HIP(hipMemcpy(host_tok.data(), tokens, host_tok.size() * 4, hipMemcpyDeviceToHost));
    } else {
        memcpy(host_tok.data(), tokens, host_tok.size() * 4);
    }
    std::vector<int32_t> tb((size_t)T * B);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            const int32_t tok = host_tok[(size_t)b * T + t];
            WM_REQUIRE(tok >= 0 && tok < V, WM_ERR_INVALID, "token id %d outside [0, %d)", tok, V);
            tb[(size_t)t * B + b] = tok;
        }
    if (ctx->dbg_hooks) {   // debug library only: the all-fp32 path (xa stays f32, no K/V cache, no bf16 anywhere)
        const float *d_xa = xa;
        float *d_out = logits;
        char *st = nullptr;
        const size_t xa_b = (size_t)B * 1500 * m->dims.n_audio_state * 4, out_b = (size_t)B * T * V * 4;
        if (mem == WM_MEM_HOST) {
            WM_TRY(io_stage(ctx, ((xa_b + 255) & ~(size_t)255) + out_b, &st));
            WM_HIP(hipMemcpyAsync(st, xa, xa_b, hipMemcpyHostToDevice, ctx->stream));
            d_xa = (const float *)st;
            d_out = (float *)(st + ((xa_b + 255) & ~(size_t)255));
        }
        WM_TRY(ctx->dbg_hooks->decode_logits_dev(ctx, host_tok.data(), B, T, d_xa, d_out));
        if (mem == WM_MEM_HOST) {
            WM_HIP(hipMemcpyAsync(logits, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
            WM_HIP(hipStreamSynchronize(ctx->stream));
        }
        return WM_OK;
    }
    WM_TRY(wm_model_decode_begin(ctx, B));
    WM_TRY(load_xa(ctx, xa, B, mem));
    WM_HIP(hipMemcpyAsync(m->dseq, tb.data(), tb.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));  // fences `tb`
    WM_TRY(wm_model_set_pos(ctx, 0));
    WM_TRY(wm_model_embed_first(ctx, B));
    float *d_out = logits;
    char *st = nullptr;
    if (mem == WM_MEM_HOST) {
        // io_stage currently holds xa (already consumed into bf16 by load_xa on this stream)
        WM_HIP(hipMalloc((void **)&st, (size_t)B * T * V * 4));
        d_out = (float *)st;
    }
    int rc = WM_OK;
    for (int t = 0; t < T && rc == WM_OK; ++t) {
        rc = wm_model_decode_step(ctx, B, true, 0, V - 1);
        if (rc != WM_OK) break;
        // dlogits [B][vpad] -> out [B][T][V], row t
        if (hipMemcpy2DAsync(d_out + (size_t)t * V, (size_t)T * V * 4, m->dlogits, (size_t)m->vpad * 4,
                             (size_t)V * 4, B, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
            wm_set_error("hipMemcpy2DAsync failed");
            rc = WM_ERR_HIP;
            break;
        }
        // teacher forcing: every position < T is "prompt", so the given tokens are kept; this
        // embeds position t+1 and advances the device-side position
        rc = wm_model_close_step(ctx, B, T, true, nullptr, 0);
    }
    if (rc == WM_OK && mem == WM_MEM_HOST) {
        if (hipMemcpyAsync(logits, d_out, (size_t)B * T * V * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
            wm_set_error("hipMemcpyAsync D2H failed");
            rc = WM_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == WM_OK) {
        wm_set_error("stream sync failed: %s", hipGetErrorString(hipGetLastError()));
        rc = WM_ERR_HIP;
    }
    if (st) (void)hipFree(st);
    return rc;
} WM_API_CATCH

static int detect_language_impl(wm_ctx *ctx, const float *xa, int B, int32_t sot, int32_t lang_first, int32_t lang_last,
                                int32_t *lang_idx, float *probs, wm_mem mem);

extern "C" int wm_detect_language(wm_ctx *ctx, const float *xa, int B, int32_t sot, int32_t lang_first,
                                  int32_t lang_last, int32_t *lang_idx, wm_mem mem) try {
    return detect_language_impl(ctx, xa, B, sot, lang_first, lang_last, lang_idx, nullptr, mem);
} WM_API_CATCH

extern "C" int wm_detect_language_probs(wm_ctx *ctx, const float *xa, int B, int32_t sot, int32_t lang_first,
                                        int32_t lang_last, int32_t *lang_idx, float *probs, wm_mem mem) try {
    WM_REQUIRE(probs != nullptr, WM_ERR_INVALID, "null pointer");
    return detect_language_impl(ctx, xa, B, sot, lang_first, lang_last, lang_idx, probs, mem);
} WM_API_CATCH

static int detect_language_impl(wm_ctx *ctx, const float *xa, int B, int32_t sot, int32_t lang_first, int32_t lang_last,
                                int32_t *lang_idx, float *probs, wm_mem mem) {
    WM_MODEL(ctx);
    WM_REQUIRE(xa && lang_idx, WM_ERR_INVALID, "null pointer");
    WM_REQUIRE(B >= 1 && B <= WM_DEC_MAXB, WM_ERR_INVALID, "B must be 1..%d", WM_DEC_MAXB);
    const int V = m->dims.n_vocab;
    WM_REQUIRE(sot >= 0 && sot < V && lang_first >= 0 && lang_first <= lang_last && lang_last < V, WM_ERR_INVALID,
               "token ids outside the vocabulary (n_vocab = %d)", V);
    WM_TRY(wm_model_decode_begin(ctx, B));
    // (round 6: one stream, no host synchronisation until the result is wanted -- the <|startoftranscript|> row is staged in
    // a buffer that outlives the call, uploaded FIRST, and the features' K/V GEMMs + the decoder step are enqueued behind it)
    m->lid_host.assign(B, sot);  // Whisper.swift:34-35
    WM_HIP(hipMemcpyAsync(m->dseq, m->lid_host.data(), (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_TRY(load_xa(ctx, xa, B, mem));
    WM_TRY(wm_model_set_pos(ctx, 0));
    WM_TRY(wm_model_embed_first(ctx, B));
    static const bool no_graph = getenv("WM_NO_GRAPH") != nullptr;
    auto step = [&]() -> int {
        WM_TRY(wm_model_decode_step(ctx, B, probs != nullptr, lang_first, lang_last));     // :36-37
        return wm_argmax_embed(ctx, m->dargmax, m->vpad / 16, B, nullptr, nullptr, 0, m->dresult, lang_first, nullptr,
                               nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, m->darrive, lang_first);  // :38
    };
    if (no_graph || ctx->prof.on) {
        WM_TRY(step());
    } else {
        WmModel::LidGraph &lg = m->lid_graph;
        const int want = probs != nullptr ? 1 : 0;
        if (!lg.e || lg.B != B || lg.cap_b != m->cap_b || lg.first != lang_first || lg.last != lang_last || lg.logits != want) {
            lg.destroy();
            WM_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            const int crc = step();
            const hipError_t ce = hipStreamEndCapture(ctx->stream, &lg.g);
            if (crc != WM_OK || ce != hipSuccess) {
                if (ce == hipSuccess && lg.g) (void)hipGraphDestroy(lg.g);
                lg.g = nullptr;
                if (crc != WM_OK) return crc;
                WM_HIP(ce);
            }
            if (hipGraphInstantiate(&lg.e, lg.g, nullptr, nullptr, 0) != hipSuccess) {
                lg.destroy();
                wm_set_error("hipGraphInstantiate failed for the language-identification step");
                return WM_ERR_HIP;
            }
            lg.B = B; lg.cap_b = m->cap_b; lg.first = lang_first; lg.last = lang_last; lg.logits = want;
        }
        WM_HIP(hipGraphLaunch(lg.e, ctx->stream));
    }
    if (probs) {  // openai-whisper detect_language(): softmax over the language-token logits only
        const int n_lang = lang_last - lang_first + 1;
        float *d_probs = probs;
        char *st = nullptr;
        if (mem == WM_MEM_HOST) {
            WM_TRY(io_stage(ctx, (size_t)B * n_lang * 4, &st));
            d_probs = (float *)st;
        }
        WM_TRY(wm_range_softmax(ctx, m->dlogits, m->vpad, B, lang_first, n_lang, d_probs));
        if (mem == WM_MEM_HOST)
            WM_HIP(hipMemcpyAsync(probs, d_probs, (size_t)B * n_lang * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (mem == WM_MEM_DEVICE) {   // the caller's buffer is device memory: one device-to-device copy behind the step
        WM_HIP(hipMemcpyAsync(lang_idx, m->dresult, (size_t)B * 4, hipMemcpyDeviceToDevice, ctx->stream));
        WM_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        WM_HIP(hipMemcpyAsync(lang_idx, m->dresult, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
        WM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return WM_OK;
}

// ------------------------------------------------------------------ greedy transcription
static size_t pcm_elem(wm_dtype t) { return t == WM_I16 ? 2 : t == WM_F32 ? 4 : 8; }

extern "C" int wm_set_token_budgets(wm_ctx *ctx, const int32_t *budgets, int n) try {
    WM_MODEL(ctx);
    WM_REQUIRE(n >= 0 && (n == 0 || budgets), WM_ERR_INVALID, "set_token_budgets: bad list");
    for (int i = 0; i < n; ++i)
        WM_REQUIRE(budgets[i] >= 1, WM_ERR_INVALID, "set_token_budgets: budget %d of chunk %d is < 1", budgets[i], i);
    m->budget_host.assign(budgets, budgets + n);
    return WM_OK;
} WM_API_CATCH

// ---------------------------------------------------------------- greedy transcription
// One batch's decode is a chain of ~260 dependent launches per position and is bound by launch latency, not by
// HBM (NOTEBOOK.md section 4), so a call with more chunks than one decode group is spread over LANES: weight-sharing
// clones of the context (wm_clone), each with its own stream, activations, KV caches and decode graphs.  The
// single host thread drives the lanes as a small non-blocking scheduler: a lane takes the next decode group as soon as
// it has finished its previous one, positions are enqueued in BURSTS (one hipGraph of WM_BURST consecutive positions --
// the arg-max kernel advances the device-side position, so consecutive positions do not depend on the host), and with
// early stop on (eot >= 0 or per-chunk token budgets) a lane stays at most two bursts ahead of the GPU and stops
// enqueuing once the device reports that no sequence of its group is live any more.
namespace {
constexpr int kGroupChunks = 8;   // smallest decode group worth a lane (BASELINE.json configs[3]); up to WM_DEC_MAXB

int lane_limit() {
    static const int n = [] {
        const char *e = getenv("WM_LANES");
        const int v = e ? atoi(e) : 3;
        return v < 1 ? 1 : (v > 8 ? 8 : v);
    }();
    return n;
}

int burst_len() {
    static const int n = [] {
        const char *e = getenv("WM_BURST");
        const int v = e ? atoi(e) : 8;
        return v < 1 ? 1 : (v > 32 ? 32 : v);
    }();
    return n;
}

struct LaneJob {
    enum State { IDLE, DECODING, DRAINING };
    wm_ctx *c = nullptr;
    int b0 = 0, Bg = 0;
    State state = IDLE;
    int t = 0;          // decoder positions enqueued so far
    int bursts = 0;     // bursts enqueued so far
    bool stopped = false;   // the device reported zero live rows
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t burst_ev[WM_NLIVE_RING] = {};
    std::vector<int32_t> pr, gen, bud;
    float stage_sum[3] = {0.f, 0.f, 0.f};
    ~LaneJob() {
        if (c) (void)hipStreamSynchronize(c->stream);  // error paths: nothing may outlive pr / gen / bud
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : burst_ev)
            if (e) (void)hipEventDestroy(e);
    }
};

struct StopCfg {
    bool on = false;
    int32_t eot = -1;
    const int32_t *budgets = nullptr;  // [B] of the call, already clamped to max_new (null: none)
};

// front end -> encoder -> cross K/V -> prompt upload -> first embedding, all enqueued on the lane's stream
int lane_prefill(LaneJob &j, const void *pcm, wm_dtype pcm_dtype, const int32_t *prompt, int n_prompt, wm_mem mem,
                 const StopCfg &stop) {
    wm_ctx *c = j.c;
    WmModel *m = c->model;
    const wm_dims &D = m->dims;
    const int Bg = j.Bg;
    const size_t bytes = (size_t)Bg * WM_N_SAMPLES * pcm_elem(pcm_dtype);
    const void *d_pcm = (const char *)pcm + (size_t)j.b0 * WM_N_SAMPLES * pcm_elem(pcm_dtype);
    if (mem == WM_MEM_HOST) {
        if (m->pcm_stage_bytes < bytes) {
            WM_HIP(hipStreamSynchronize(c->stream));
            if (m->pcm_stage) WM_HIP(hipFree(m->pcm_stage));
            m->pcm_stage = nullptr;
            m->pcm_stage_bytes = 0;
            WM_HIP(hipMalloc(&m->pcm_stage, bytes));
            m->pcm_stage_bytes = bytes;
        }
        WM_HIP(hipMemcpyAsync(m->pcm_stage, d_pcm, bytes, hipMemcpyHostToDevice, c->stream));
        d_pcm = m->pcm_stage;
    }
    // decode state first (prompt tokens [n_prompt][Bg], position 0): a pageable H2D copy may wait for the
    // stream to drain, so it is issued while the lane is still idle
    WM_TRY(wm_model_decode_begin(c, Bg));
    j.pr.resize((size_t)n_prompt * Bg);
    for (int t = 0; t < n_prompt; ++t)
        for (int b = 0; b < Bg; ++b) j.pr[(size_t)t * Bg + b] = prompt[t];
    WM_HIP(hipMemcpyAsync(m->dseq, j.pr.data(), j.pr.size() * 4, hipMemcpyHostToDevice, c->stream));
    WM_TRY(wm_model_set_pos(c, 0));
    // early-stop state of this group: done flags, live list, per-row budgets (kernel arguments of the decode graphs)
    m->stop_on = stop.on;
    m->stop_eot = stop.eot;
    m->budget_on = stop.on && stop.budgets != nullptr;
    if (stop.on) {
        if (m->budget_on) {
            j.bud.assign(stop.budgets + j.b0, stop.budgets + j.b0 + Bg);
            WM_HIP(hipMemcpyAsync(m->dbudget, j.bud.data(), (size_t)Bg * 4, hipMemcpyHostToDevice, c->stream));
        }
        WM_TRY(wm_stop_init(c, wm_model_stop_dev(m), Bg));
    }
    WM_TRY(wm_model_reserve(c, Bg));
    WM_HIP(hipEventRecord(j.ev[0], c->stream));
    // 1. log-mel front end (f32 fast path), output stays in HBM
    WM_TRY(wm_frontend_run(&c->fe, &c->prof, c->stream, d_pcm, pcm_dtype, Bg, D.n_mels, m->mel_f32, WM_F32));
    WM_HIP(hipEventRecord(j.ev[1], c->stream));
    // 2. encoder + cross-attention K/V
    WM_TRY(wm_model_encode_dev(c, m->mel_f32, Bg, nullptr));
    WM_TRY(wm_model_cross_kv(c, Bg));
    WM_HIP(hipEventRecord(j.ev[2], c->stream));
    // 3. embedding of the first prompt token (+ the initial timestamp-rule state)
    WM_TRY(wm_model_embed_first(c, Bg));
    if (m->ts_on) WM_TRY(wm_ts_init(c, wm_model_ts_dev(m), Bg));
    return WM_OK;
}

// One decoder position = 8 launches per layer + logits + arg-max/embed (which writes the next token, embeds the
// next position and advances *dpos).  Nothing in it depends on host state, so it is captured ONCE into a
// hipGraph per lane and replayed for every position -- and `burst` consecutive positions are captured as one more graph.
int capture_positions(LaneJob &j, int n_prompt, int n_pos, hipGraph_t *g, hipGraphExec_t *ge) {
    wm_ctx *c = j.c;
    WmModel *m = c->model;
    WM_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int crc = WM_OK;
    for (int i = 0; i < n_pos && crc == WM_OK; ++i) {
        crc = wm_model_decode_step(c, j.Bg, false, 0, m->dims.n_vocab - 1, m->mask_on ? n_prompt - 1 : -1, m->ts_on);
        if (crc == WM_OK) crc = wm_model_close_step(c, j.Bg, n_prompt, true, nullptr, 0, m->ts_on);
    }
    hipError_t ce = hipStreamEndCapture(c->stream, g);
    if (crc != WM_OK || ce != hipSuccess) {   // a half-captured graph is of no use: do not leave it behind
        if (ce == hipSuccess && *g) (void)hipGraphDestroy(*g);
        *g = nullptr;
        if (crc != WM_OK) return crc;
        WM_HIP(ce);
    }
    if (hipGraphInstantiate(ge, *g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(*g);
        *g = nullptr;
        *ge = nullptr;
        wm_set_error("hipGraphInstantiate failed for the %d-position decode graph", n_pos);
        return WM_ERR_HIP;
    }
    return WM_OK;
}

// Select (creating it if needed) the graph set of the lane's current decode shape.  The graphs themselves are captured
// on first use, per sharing mode, by lane_burst.
int lane_graph(LaneJob &j, int n_prompt) {
    wm_ctx *c = j.c;
    WmModel *m = c->model;
    const int mk = (m->mask_on ? 1 : 0) | (m->ts_on ? 2 : 0);
    const int sk = m->stop_on ? (1 | (m->budget_on ? 2 : 0) | ((m->stop_eot + 2) << 2)) : 0;
    int cur = -1;
    for (size_t i = 0; i < m->graph_sets.size(); ++i) {
        const WmModel::GraphSet &g = m->graph_sets[i];
        if (g.B == j.Bg && g.n_prompt == n_prompt && g.cap_b == m->cap_b && g.mask == mk && g.stop_key == sk) cur = (int)i;
    }
    if (cur < 0) {
        if ((int)m->graph_sets.size() >= WmModel::kMaxGraphSets) {   // evict the least recently used shape
            size_t old = 0;
            for (size_t i = 1; i < m->graph_sets.size(); ++i)
                if (m->graph_sets[i].stamp < m->graph_sets[old].stamp) old = i;
            m->graph_sets[old].destroy();
            m->graph_sets.erase(m->graph_sets.begin() + (long)old);
        }
        WmModel::GraphSet g;
        g.B = j.Bg; g.n_prompt = n_prompt; g.cap_b = m->cap_b; g.mask = mk; g.stop_key = sk;
        m->graph_sets.push_back(g);
        cur = (int)m->graph_sets.size() - 1;
    }
    m->graph_cur = cur;
    m->graph_sets[cur].stamp = ++m->graph_clock;
    return WM_OK;
}

// enqueue the next burst of positions of a lane (<= burst_len(), up to the end of the sequence).  `shared`: other decode
// groups are in flight on the device right now -- this burst's cross-attention launches are the short-lived shape.
int lane_burst(LaneJob &j, int n_prompt, int n_steps, bool use_graph, bool stop_on, bool shared) {
    wm_ctx *c = j.c;
    WmModel *m = c->model;
    const int K = burst_len();
    const int k = n_steps - j.t < K ? n_steps - j.t : K;
    const int mode = shared ? 1 : 0;
    m->xattn_shared = shared;   // read by wm_model_decode_step (eager launches and captures alike)
    WmModel::GraphSet *g = use_graph ? &m->graph_sets[m->graph_cur] : nullptr;
    if (use_graph && k == K && K > 1) {
        if (!g->ek[mode] || g->burst[mode] != K) {
            if (g->ek[mode]) { (void)hipGraphExecDestroy(g->ek[mode]); g->ek[mode] = nullptr; }
            if (g->gk[mode]) { (void)hipGraphDestroy(g->gk[mode]); g->gk[mode] = nullptr; }
            const int rc = capture_positions(j, n_prompt, K, &g->gk[mode], &g->ek[mode]);
            if (rc != WM_OK) { g->gk[mode] = nullptr; g->ek[mode] = nullptr; return rc; }
            g->burst[mode] = K;
        }
        WM_HIP(hipGraphLaunch(g->ek[mode], c->stream));
    } else {
        if (use_graph && !g->e1[mode]) {
            const int rc = capture_positions(j, n_prompt, 1, &g->g1[mode], &g->e1[mode]);
            if (rc != WM_OK) { g->g1[mode] = nullptr; g->e1[mode] = nullptr; return rc; }
        }
        for (int i = 0; i < k; ++i) {
            if (use_graph) {
                WM_HIP(hipGraphLaunch(g->e1[mode], c->stream));
            } else {
                WM_TRY(wm_model_decode_step(c, j.Bg, false, 0, m->dims.n_vocab - 1, m->mask_on ? n_prompt - 1 : -1, m->ts_on));
                WM_TRY(wm_model_close_step(c, j.Bg, n_prompt, true, nullptr, 0, m->ts_on));
            }
        }
    }
    j.t += k;
    if (stop_on) {  // the live-row count after this burst, where the host can read it without touching the stream
        const int slot = j.bursts % WM_NLIVE_RING;
        WM_HIP(hipMemcpyAsync(m->h_nlive + slot, m->dnlive, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        WM_HIP(hipEventRecord(j.burst_ev[slot], c->stream));
    }
    ++j.bursts;
    return WM_OK;
}
}  // namespace

// Decode groups of a call of B chunks on at most L lanes (pure: tests/test_abi.py pins the table through the debug library).
// explicit_lanes: the host set a lane count (wm_set_lanes n > 0); gc_probe: the debug knob group_chunks (0 in the product).
// More groups than lanes run in rounds of the lanes; a group never exceeds WM_DEC_MAXB rows.
int wm_group_count(int B, int L, bool explicit_lanes, int gc_probe) {
    if (L < 1) L = 1;
    int G;
    const int gc = gc_probe > 0 ? gc_probe : kGroupChunks;
    if (!explicit_lanes && gc_probe == 0 && B < 144) {
        G = B < 32 ? 1 : 2;       // the measured policy (comment in wm_transcribe_greedy)
        if (G > L) G = L;
    } else if (B <= gc * L) {
        G = (B + gc - 1) / gc;    // the rounds-1-4 rule: groups of ~gc while there is a lane for each
    } else {
        G = (B + WM_DEC_MAXB - 1) / WM_DEC_MAXB;
        if (G < L) G = L;
        G = (G + L - 1) / L * L;
    }
    const int g_min = (B + WM_DEC_MAXB - 1) / WM_DEC_MAXB;   // (one lane, 129 .. 143 chunks: still two groups)
    return G < g_min ? g_min : (G < 1 ? 1 : G);
}

// SUB-CHIP LANES (round 6): P = 2 (or 3) decode groups of a call, each on its OWN part of the chip -- weight-sharing clones
// whose streams carry complementary CU masks (wm_clone_cus: a slice of the CUs of every XCD) -- instead of one
// latency-bound chain, or unmasked chains whose every launch floods all 256 CUs.  Returns 0 when the call is served better
// by the unmasked lanes of wm_group_count.  Launch shapes only: same kernels, same bits (tests).
// MEASURED (profiles/r06_group_policy.txt, r06_solo_lane_latency.txt): whether it pays is decided by how much ONE chain
// needs the CUs.  A NARROW model's chain does not (base, 16 rows: 0.2375 ms per position on 256 CUs, 0.2452 on 128 --
// launch latency, the matrices are 0.5-2 MB), so two half-chip chains run truly side by side: base x 32 +7 % over the best
// unmasked split (14 350 vs 13 400 audio-s/s), x 64 +9 %, tiny.en x 32 +6 %.  A WIDE model's chain does: large-v2, 8 rows,
// 1.55 ms per position on 256 CUs, 2.21 on 128 -- the streaming phases of its big launches are bound by bytes in flight PER
// CU (cross-attention 61 MB: 16.4 -> 27.1 us, fc1 13 MB: 6.4 -> 10.7 us on half the CUs), so two half-chip chains lose to
// one group at every size (15 chunks: 765 vs 937; 24: 1098 vs 1170; 48: 1522 vs 1582) and three thirds lose more.
int wm_lane_parts(int B, int L, bool explicit_lanes, int n_text_state, int n_text_layer) {
    (void)n_text_layer;
    const int knob = g_wm_tuning.lane_parts;     // 0 in the product
    if (knob == 1) return 0;
    if (knob == 2 || knob == 3) return (B >= 2 * knob && B <= knob * WM_DEC_MAXB) ? knob : 0;
    if (explicit_lanes || L < 2) return 0;       // a host that sets a lane count gets the lanes it asked for
    if (n_text_state <= 384) return (B >= 32 && B < 48) ? 2 : 0;      // tiny: 32 .. 47 chunks (+6 %; -2 % from 48)
    if (n_text_state <= 512) return (B >= 24 && B <= 128) ? 2 : 0;    // base: 24 .. 128 chunks (+2 .. +9 %; 40 - 48: -1 %)
    return 0;                                    // d >= 768: the chain needs the whole chip
}

extern "C" int wm_transcribe_greedy(wm_ctx *ctx, const void *pcm, wm_dtype pcm_dtype, int B,
                                    const int32_t *prompt, int n_prompt, int max_new, int32_t eot,
                                    int32_t *tokens_out, int32_t *lens_out, wm_mem mem) try {
    WM_MODEL(ctx);
    // per-chunk token budgets set for THIS call (wm_set_token_budgets) are consumed by it whatever happens next: a call
    // that fails validation must not leave them armed for a later, unrelated call with the same B
    std::vector<int32_t> budgets;
    budgets.swap(m->budget_host);
    WM_REQUIRE(m->finalized, WM_ERR_STATE, "model weights not finalised (wm_finalize)");
    WM_REQUIRE(pcm && prompt && tokens_out && lens_out, WM_ERR_INVALID, "null pointer");
    WM_REQUIRE(pcm_dtype == WM_I16 || pcm_dtype == WM_F32 || pcm_dtype == WM_F64, WM_ERR_INVALID, "bad pcm dtype");
    WM_REQUIRE(B >= 1, WM_ERR_INVALID, "B < 1");
    const wm_dims &D = m->dims;
    WM_REQUIRE(n_prompt >= 1 && max_new >= 1 && n_prompt + max_new <= D.n_text_ctx, WM_ERR_INVALID,
               "prompt (%d) + new tokens (%d) must fit the %d-token context", n_prompt, max_new, D.n_text_ctx);
    for (int i = 0; i < n_prompt; ++i)
        WM_REQUIRE(prompt[i] >= 0 && prompt[i] < D.n_vocab, WM_ERR_INVALID, "prompt token %d out of range", prompt[i]);
    WM_REQUIRE(eot < D.n_vocab, WM_ERR_INVALID, "eot %d outside the vocabulary", eot);
    WM_REQUIRE(budgets.empty() || (int)budgets.size() == B, WM_ERR_INVALID,
               "token budgets were set for %d chunks, the call has %d", (int)budgets.size(), B);
    for (auto &b : budgets) b = b > max_new ? max_new : b;
    static const bool no_graph = getenv("WM_NO_GRAPH") != nullptr;
    const bool no_stop = g_wm_tuning.no_early_stop != 0;   // probes only: decode every position, truncate on the host
    const bool use_graph = !no_graph && !ctx->prof.on;
    StopCfg stop;
    stop.on = !no_stop && (eot >= 0 || !budgets.empty());
    stop.eot = eot;
    stop.budgets = budgets.empty() ? nullptr : budgets.data();
    // Split the B chunks into G balanced decode groups (<= WM_DEC_MAXB each, kGroupChunks preferred) and run
    // them on L lanes.  Per-kernel profiling keeps everything on the caller's context (one lane).
    // GROUP POLICY (round 5, measured: profiles/r05_group_policy.txt).  Rounds 1-4 cut every call into groups of ~8 chunks
    // on up to three lanes.  At large-v2 that is the WORST choice below ~50 chunks (15 chunks: 8 + 7 on two lanes 826
    // audio-s/s, one group of 15 860; 48 chunks: 3 x 16 1449, one group 1538): a group's weight stream is shared by all its
    // rows, and two latency-bound chains on two hardware queues do not overlap for free.  Two groups start to pay once each
    // is big enough to be bandwidth-bound (64: 2 x 32 1697 vs 1661; 96: 2 x 48 1946 vs 1860; 128: 2 x 64 2019 vs 1896), three
    // from ~150 (160: 3 x 53 = 2 x 80).  A small model wants the second group earlier (base, 32 chunks: 2 x 16 +10 %), hence
    // the threshold of 32.  A host that SETS a lane count (wm_set_lanes n > 1) asks for n groups in flight whenever there are
    // 8 chunks for each -- the rounds-1-4 rule, and what keeps the lanes under test at small sizes.
    const bool explicit_lanes = ctx->max_lanes > 0;
    const int L = ctx->prof.on ? 1 : (explicit_lanes ? ctx->max_lanes : lane_limit());
    const int solo = g_wm_tuning.lane_solo_cus;   // probes only (0 in the product)
    int parts = (ctx->prof.on || solo || ctx->no_cu_masks) ? 0 : wm_lane_parts(B, L, explicit_lanes, D.n_text_state, D.n_text_layer);
    if (parts) {
        // the sub-chip lanes of this partition, created on first use.  A device / driver that refuses CU-masked streams (a
        // partitioned GPU, an older KFD) is not an error: the call falls back to the unmasked policy, once and for all
        std::vector<wm_ctx *> &pl = ctx->part_lanes[parts - 2];
        while ((int)pl.size() < parts) {
            wm_ctx *c = nullptr;
            const int k = (int)pl.size();
            if (wm_clone_cus(ctx, k * 32 / parts, (k + 1) * 32 / parts, &c) != WM_OK) {   // 16 + 16, or 10 + 11 + 11 CUs of every XCD
                ctx->no_cu_masks = true;
                parts = 0;
                break;
            }
            pl.push_back(c);
        }
        WM_TRY(wm_ctx_make_current(ctx));
    }
    const int G = parts ? parts : wm_group_count(B, L, explicit_lanes, g_wm_tuning.group_chunks);
    const int n_lanes = solo ? 1 : (parts ? parts : (G < L ? G : L));
    wm_ctx *solo_ctx = nullptr;
    if (solo) {
        WM_REQUIRE(solo >= 1 && solo <= 31, WM_ERR_INVALID, "lane_solo_cus: 1 .. 31 CUs per XCD");
        wm_ctx *&c = ctx->solo_lanes[solo];
        if (!c) WM_TRY(wm_clone_cus(ctx, 0, solo, &c));
        solo_ctx = c;
    } else if (!parts) {
        while ((int)ctx->lanes.size() < n_lanes - 1) {
            wm_ctx *c = nullptr;
            WM_TRY(wm_clone(ctx, &c));
            ctx->lanes.push_back(c);
        }
    }
    std::vector<LaneJob> jobs(n_lanes);
    for (int l = 0; l < n_lanes; ++l) {
        jobs[l].c = solo_ctx ? solo_ctx : parts ? ctx->part_lanes[parts - 2][l] : (l == 0 ? ctx : ctx->lanes[l - 1]);
        for (auto &e : jobs[l].ev) WM_HIP(hipEventCreate(&e));
        if (stop.on)
            for (auto &e : jobs[l].burst_ev) WM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    ctx->stage_ms[0] = ctx->stage_ms[1] = ctx->stage_ms[2] = 0.f;
    const int n_steps = n_prompt + max_new - 1;
    // decodes in flight on this device (this call included): other lanes of this call, other contexts' calls
    struct ActiveGuard {
        std::atomic<int> &n;
        explicit ActiveGuard(std::atomic<int> &a) : n(a) { n.fetch_add(1, std::memory_order_relaxed); }
        ~ActiveGuard() { n.fetch_sub(1, std::memory_order_relaxed); }
    } active_guard(g_wm_active_decodes[ctx->device & 63]);
    const int base = B / G, rem = B % G;
    int next_group = 0, groups_done = 0;
    while (groups_done < G) {
        bool progress = false;
        for (int l = 0; l < n_lanes; ++l) {
            LaneJob &j = jobs[l];
            WM_TRY(wm_ctx_make_current(j.c));
            if (j.state == LaneJob::IDLE) {
                if (next_group >= G) continue;
                const int g = next_group++;
                j.Bg = base + (g < rem ? 1 : 0);
                j.b0 = g * base + (g < rem ? g : rem);
                j.t = 0; j.bursts = 0; j.stopped = false;
                WM_TRY(lane_prefill(j, pcm, pcm_dtype, prompt, n_prompt, mem, stop));
                if (use_graph) WM_TRY(lane_graph(j, n_prompt));
                j.state = LaneJob::DECODING;
                progress = true;
                continue;   // the other lanes get their prefill before anyone's first burst
            }
            if (j.state == LaneJob::DECODING) {
                if (stop.on && j.bursts >= 2 && !j.stopped) {
                    // stay at most two bursts ahead of the GPU: burst (bursts - 2) must have finished, and its live count
                    // says whether there is anything left to decode
                    const int slot = (j.bursts - 2) % WM_NLIVE_RING;
                    const hipError_t q = hipEventQuery(j.burst_ev[slot]);
                    if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }   // "not ready" is not an error to keep
                    WM_HIP(q);
                    if (j.c->model->h_nlive[slot] == 0) j.stopped = true;
                }
                if (j.t < n_steps && !j.stopped) {
                    // does this burst share the chip?  other lanes of this call still decoding, or other calls in flight
                    int busy = 0;
                    for (int o = 0; o < n_lanes; ++o) busy += jobs[o].state == LaneJob::DECODING && jobs[o].t < n_steps && !jobs[o].stopped;
                    // (sub-chip lanes own their CUs: the other lanes of THIS call do not make the chip "shared")
                    const bool shared = (busy > 1 && !parts) || g_wm_active_decodes[ctx->device & 63].load(std::memory_order_relaxed) > 1;
                    WM_TRY(lane_burst(j, n_prompt, n_steps, use_graph, stop.on, shared));
                    progress = true;
                    continue;
                }
                // everything enqueued (or nothing left to decode): fetch the token streams
                WM_HIP(hipEventRecord(j.ev[3], j.c->stream));
                j.gen.resize((size_t)max_new * j.Bg);  // dseq[n_prompt + i][b]
                WM_HIP(hipMemcpyAsync(j.gen.data(), j.c->model->dseq + (size_t)n_prompt * j.Bg, j.gen.size() * 4,
                                      hipMemcpyDeviceToHost, j.c->stream));
                j.state = LaneJob::DRAINING;
                progress = true;
                continue;
            }
            // DRAINING: a decode group runs for seconds -- poll instead of spinning in hipStreamSynchronize, so that the
            // host threads of the other lanes / ranks (one process per GPU, several contexts each) keep their cores
            const hipError_t q = hipStreamQuery(j.c->stream);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
            WM_HIP(q);
            WM_HIP(hipStreamSynchronize(j.c->stream));
            for (int b = 0; b < j.Bg; ++b) {
                int len = max_new;
                if (stop.budgets && stop.budgets[j.b0 + b] < len) len = stop.budgets[j.b0 + b];
                for (int i = 0; i < len; ++i)
                    if (eot >= 0 && j.gen[(size_t)i * j.Bg + b] == eot) { len = i + 1; break; }
                for (int i = 0; i < max_new; ++i)
                    tokens_out[(size_t)(j.b0 + b) * max_new + i] = i < len ? j.gen[(size_t)i * j.Bg + b] : eot;
                lens_out[j.b0 + b] = len;
            }
            float ms;
            for (int i = 0; i < 3; ++i)
                if (hipEventElapsedTime(&ms, j.ev[i], j.ev[i + 1]) == hipSuccess) j.stage_sum[i] += ms;
            j.state = LaneJob::IDLE;
            ++groups_done;
            progress = true;
        }
        if (!progress) usleep(100);
    }
    for (int l = 0; l < n_lanes; ++l)   // lanes overlap: the busiest lane per stage
        for (int i = 0; i < 3; ++i)
            if (jobs[l].stage_sum[i] > ctx->stage_ms[i]) ctx->stage_ms[i] = jobs[l].stage_sum[i];
    return WM_OK;
} WM_API_CATCH
