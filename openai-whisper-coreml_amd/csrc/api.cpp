// api.cpp -- the C ABI of libwhisper_mi355x.so (declared in include/whisper_mi355x.h):
// context life cycle, error reporting, the front-end entry points (boundary #1), device
// memory helpers and the HIP-event profiler.  The model entry points (boundary #2) are in
// model_api.cpp.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_ext.h>

#include "wm_internal.h"
#include "model.h"

WmTuning g_wm_tuning;   // wm_internal.h: defaults = the product; only the debug library has a setter
std::atomic<int> g_wm_active_decodes[64];

// ---------------------------------------------------------------- errors -------------
static thread_local std::string g_last_error;

void wm_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char *wm_last_error(void) { return g_last_error.c_str(); }

// ---------------------------------------------------------------- profiler -----------
hipEvent_t WmProfiler::get() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void WmProfiler::begin(const char *, hipStream_t s, hipEvent_t *e0) {
    *e0 = get();
    (void)hipEventRecord(*e0, s);
}
void WmProfiler::end(const char *name, hipStream_t s, hipEvent_t e0) {
    hipEvent_t e1 = get();
    (void)hipEventRecord(e1, s);
    WmProfFamily &f = fam[name];
    f.pending.emplace_back(e0, e1);
    if (f.pending.size() >= 4096) drain();
}
void WmProfiler::drain() {
    for (auto &kv : fam) {
        for (auto &pr : kv.second.pending) {
            (void)hipEventSynchronize(pr.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.ms += ms;
                kv.second.n += 1;
            }
            pool.push_back(pr.first);
            pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}
void WmProfiler::reset() {
    drain();
    fam.clear();
}
WmProfiler::~WmProfiler() {
    for (auto &kv : fam)
        for (auto &pr : kv.second.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
}

// ---------------------------------------------------------------- context ------------
int wm_ctx_make_current(const wm_ctx *ctx) {
    WM_REQUIRE(ctx != nullptr, WM_ERR_INVALID, "null context");
    WM_HIP(hipSetDevice(ctx->device));
    return WM_OK;
}

// Bit i of a CU mask enables CU i / 8 of XCD i % 8 (the KFD deals the bits round-robin over the 8 XCCs; measured with the
// census of tools/cu_mask_lab.hip, profiles/r06_cu_mask_lab.txt: bits 0..7 = one CU in each XCC, bits 0..31 = four): 32 CUs
// per XCD.  An XCD cannot be switched OFF by the mask: one whose 32 bits are all zero runs on a fallback set (the 0x0f-bytes
// mask "XCDs 0-3 only" measured 256 CUs), so a lane is a slice [lo, hi) of the CUs of EVERY XCD -- which also keeps the
// workgroup id % 8 -> XCD placement the L2 warm-up workgroups and the tile maps assume.
int wm_cu_mask(int cu_lo, int cu_hi, uint32_t mask[8]) {
    for (int w = 0; w < 8; ++w) mask[w] = 0;
    int n = 0;
    for (int i = 0; i < 256; ++i) {
        const int cu = i >> 3;
        if (cu >= cu_lo && cu < cu_hi) { mask[i >> 5] |= 1u << (i & 31); ++n; }
    }
    return n;
}

static int ctx_new(int device, wm_ctx **out, int cu_lo = 0, int cu_hi = 32) {
    WM_REQUIRE(out != nullptr, WM_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    WM_HIP(hipGetDeviceCount(&n));
    WM_REQUIRE(device >= 0 && device < n, WM_ERR_HIP, "device %d not present (%d visible)", device, n);
    WM_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    WM_HIP(hipGetDeviceProperties(&prop, device));
    WM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, WM_ERR_HIP,
               "device %d is %s; this library contains gfx950 code only", device, prop.gcnArchName);
    wm_ctx *c = new wm_ctx();
    c->device = device;
    hipError_t se;
    if (cu_lo > 0 || cu_hi < 32) {   // a lane confined to the CUs [cu_lo, cu_hi) of every XCD
        uint32_t mask[8];
        c->n_cus = wm_cu_mask(cu_lo, cu_hi, mask);
        c->cu_lo = cu_lo;
        c->cu_hi = cu_hi;
        se = hipExtStreamCreateWithCUMask(&c->stream, 8, mask);
    } else {
        se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (se != hipSuccess) {
        delete c;
        wm_set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
        return WM_ERR_HIP;
    }
    int st = wm_frontend_init(&c->fe, c->stream);
    if (st != WM_OK) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return st;
    }
    *out = c;
    return WM_OK;
}

extern "C" int wm_create_frontend(int device, wm_ctx **out) { return ctx_new(device, out); }

extern "C" int wm_create(const wm_dims *dims, int device, wm_ctx **out) try {
    WM_REQUIRE(dims != nullptr, WM_ERR_INVALID, "null dims");
    WM_TRY(ctx_new(device, out));
    int st = wm_model_create(*out, dims);
    if (st != WM_OK) {
        wm_destroy(*out);
        *out = nullptr;
    }
    return st;
} WM_API_CATCH

extern "C" int wm_clone(wm_ctx *parent, wm_ctx **out) try {
    WM_REQUIRE(parent && out, WM_ERR_INVALID, "null pointer");
    WM_REQUIRE(parent->model, WM_ERR_STATE, "clone: the parent context has no model");
    WM_TRY(ctx_new(parent->device, out));
    int st = wm_model_clone(*out, parent);
    if (st != WM_OK) {
        wm_destroy(*out);
        *out = nullptr;
    }
    return st;
} WM_API_CATCH

int wm_clone_cus(wm_ctx *parent, int cu_lo, int cu_hi, wm_ctx **out) {
    WM_REQUIRE(parent && out && parent->model, WM_ERR_INVALID, "clone_cus: bad arguments");
    WM_REQUIRE(cu_lo >= 0 && cu_lo < cu_hi && cu_hi <= 32, WM_ERR_INVALID, "clone_cus: CUs [%d, %d) of 32 per XCD", cu_lo, cu_hi);
    WM_TRY(ctx_new(parent->device, out, cu_lo, cu_hi));
    int st = wm_model_clone(*out, parent);
    if (st != WM_OK) {
        wm_destroy(*out);
        *out = nullptr;
    }
    return st;
}

extern "C" void wm_destroy(wm_ctx *ctx) {
    if (!ctx) return;
    for (wm_ctx *lane : ctx->lanes) wm_destroy(lane);  // clones go before the weights they alias
    ctx->lanes.clear();
    for (auto &v : ctx->part_lanes) {
        for (wm_ctx *lane : v) wm_destroy(lane);
        v.clear();
    }
    for (auto &kv : ctx->solo_lanes) wm_destroy(kv.second);
    ctx->solo_lanes.clear();
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    wm_model_destroy(ctx);
    wm_frontend_destroy(&ctx->fe);
    ctx->prof.reset();
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ---------------------------------------------------------------- device helpers -----
extern "C" int wm_dev_malloc(wm_ctx *ctx, size_t bytes, void **dptr) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(dptr != nullptr, WM_ERR_INVALID, "null dptr");
    WM_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_dev_free(wm_ctx *ctx, void *dptr) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    if (dptr) WM_HIP(hipFree(dptr));
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_dev_upload(wm_ctx *ctx, void *dptr, const void *host, size_t bytes) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(dptr && host, WM_ERR_INVALID, "null pointer");
    WM_HIP(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_dev_download(wm_ctx *ctx, void *host, const void *dptr, size_t bytes) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(dptr && host, WM_ERR_INVALID, "null pointer");
    WM_HIP(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_sync(wm_ctx *ctx) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    return WM_OK;
} WM_API_CATCH

// ---------------------------------------------------------------- profiling ABI ------
extern "C" int wm_profile_enable(wm_ctx *ctx, int on) try {
    WM_TRY(wm_ctx_make_current(ctx));
    ctx->prof.drain();
    ctx->prof.on = (on != 0);
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_profile_reset(wm_ctx *ctx) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->prof.reset();
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_profile_json(wm_ctx *ctx, char *buf, size_t n) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(buf && n > 2, WM_ERR_INVALID, "bad buffer");
    WM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->prof.drain();
    std::string s = "{";
    bool first = true;
    for (auto &kv : ctx->prof.fam) {
        char tmp[256];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"ms\": %.6f, \"n\": %ld}", first ? "" : ", ",
                 kv.first.c_str(), kv.second.ms, kv.second.n);
        s += tmp;
        first = false;
    }
    s += "}";
    WM_REQUIRE(s.size() + 1 <= n, WM_ERR_INVALID, "buffer too small (%zu needed)", s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return WM_OK;
} WM_API_CATCH
// Cost of the measurement itself.  A kernel that spins for exactly T microseconds of the device
// clock is bracketed by the same two hipEventRecord calls the profiler uses, queued behind a long
// kernel so that the host runs ahead of the GPU (as it does in the real pipeline); the bias is the
// mean (elapsed - T): launch latency of a dependent kernel plus the completion of the closing event.
int wm_launch_spin(hipStream_t s, int *p, int grid, int cycles);
extern "C" int wm_profile_overhead_us(wm_ctx *ctx, float *us) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(us, WM_ERR_INVALID, "null pointer");
    WM_HIP(hipStreamSynchronize(ctx->stream));
    const int n = 64, spin_us = 15;
    std::vector<hipEvent_t> ev(2 * n);
    for (auto &e : ev) WM_HIP(hipEventCreate(&e));
    WM_TRY(wm_launch_spin(ctx->stream, nullptr, 160, 200000));  // 2 ms head start for the host
    for (int i = 0; i < n; ++i) {
        WM_HIP(hipEventRecord(ev[2 * i], ctx->stream));
        WM_TRY(wm_launch_spin(ctx->stream, nullptr, 160, spin_us * 100));  // wall_clock64: 100 MHz
        WM_HIP(hipEventRecord(ev[2 * i + 1], ctx->stream));
    }
    WM_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        WM_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    *us = (float)(tot * 1e3 / n) - (float)spin_us;
    return WM_OK;
} WM_API_CATCH
extern "C" int wm_last_stage_ms(wm_ctx *ctx, float out3[3]) try {
    WM_REQUIRE(ctx && out3, WM_ERR_INVALID, "null pointer");
    for (int i = 0; i < 3; ++i) out3[i] = ctx->stage_ms[i];
    return WM_OK;
} WM_API_CATCH

// ---------------------------------------------------------------- boundary #1 --------
static int ensure_scratch(wm_ctx *ctx, size_t bytes) {
    if (ctx->fe.scratch_bytes >= bytes) return WM_OK;
    if (ctx->fe.scratch) WM_HIP(hipFree(ctx->fe.scratch));
    ctx->fe.scratch = nullptr;
    ctx->fe.scratch_bytes = 0;
    WM_HIP(hipMalloc(&ctx->fe.scratch, bytes));
    ctx->fe.scratch_bytes = bytes;
    return WM_OK;
}

static size_t dtype_size(wm_dtype t) {
    switch (t) {
        case WM_I16: return 2;
        case WM_F32: return 4;
        case WM_F64: return 8;
        case WM_BF16: return 2;
    }
    return 0;
}

extern "C" int wm_logmel(wm_ctx *ctx, const void *pcm, wm_dtype pcm_dtype, int n_chunks,
                         int n_mels, void *out, wm_dtype out_dtype, wm_mem mem) try {
    WM_TRY(wm_ctx_make_current(ctx));
    WM_REQUIRE(n_chunks >= 0, WM_ERR_INVALID, "n_chunks < 0");
    WM_REQUIRE(n_mels == 80 || n_mels == 128, WM_ERR_INVALID, "n_mels must be 80 or 128, got %d", n_mels);
    WM_REQUIRE(pcm_dtype == WM_I16 || pcm_dtype == WM_F32 || pcm_dtype == WM_F64, WM_ERR_INVALID,
               "pcm dtype must be WM_I16 / WM_F32 / WM_F64");
    WM_REQUIRE(out_dtype == WM_F32 || out_dtype == WM_F64, WM_ERR_INVALID,
               "out dtype must be WM_F32 / WM_F64");
    if (n_chunks == 0) return WM_OK;
    WM_REQUIRE(pcm && out, WM_ERR_INVALID, "null pcm / out");
    if (mem == WM_MEM_DEVICE)
        return wm_frontend_run(&ctx->fe, &ctx->prof, ctx->stream, pcm, pcm_dtype, n_chunks, n_mels,
                               out, out_dtype);
    const size_t in_b = (size_t)n_chunks * WM_N_SAMPLES * dtype_size(pcm_dtype);
    const size_t out_b = (size_t)n_chunks * n_mels * WM_N_FRAMES * dtype_size(out_dtype);
    const size_t in_al = (in_b + 255) & ~(size_t)255;
    WM_TRY(ensure_scratch(ctx, in_al + out_b));
    char *d_in = (char *)ctx->fe.scratch, *d_out = d_in + in_al;
    WM_HIP(hipMemcpyAsync(d_in, pcm, in_b, hipMemcpyHostToDevice, ctx->stream));
    WM_TRY(wm_frontend_run(&ctx->fe, &ctx->prof, ctx->stream, d_in, pcm_dtype, n_chunks, n_mels,
                           d_out, out_dtype));
    WM_HIP(hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    return WM_OK;
} WM_API_CATCH

// The reference's symbol (bridge.h:11 / lib.rs:110-122).  The Rust crate's entry is re-entrant (immutable lazily
// initialised state, lib.rs:11-14), so concurrent callers must not queue behind one another here either: a small pool of
// lazily created front-end contexts (own stream, tables and staging each), handed out under a mutex and used outside it.
// At most kGsMaxContexts (8) callers run concurrently -- a ninth waits for a context to come back --; the contexts live
// for the rest of the process (the symbol has no teardown counterpart in the reference's ABI either: bridge.h declares
// one function), a few MB of HBM each; $WM_DEVICE is read once, when the first context is made.
namespace {
constexpr int kGsMaxContexts = 8;
std::mutex g_gs_mutex;
std::condition_variable g_gs_cv;
std::vector<wm_ctx *> g_gs_free;
int g_gs_created = 0;

wm_ctx *gs_acquire() {
    {
        std::unique_lock<std::mutex> lock(g_gs_mutex);
        for (;;) {
            if (!g_gs_free.empty()) {
                wm_ctx *c = g_gs_free.back();
                g_gs_free.pop_back();
                return c;
            }
            if (g_gs_created < kGsMaxContexts) {
                ++g_gs_created;
                break;  // create one outside the lock
            }
            g_gs_cv.wait(lock);
        }
    }
    static const int device = [] {
        const char *dev = getenv("WM_DEVICE");
        return dev ? atoi(dev) : 0;
    }();
    wm_ctx *c = nullptr;
    if (wm_create_frontend(device, &c) != WM_OK) {
        fprintf(stderr, "generate_spectrogram: no usable MI355X context (%s); there is no CPU fallback\n", wm_last_error());
        abort();  // the reference panics (=abort across FFI) on its internal failures too
    }
    return c;
}

void gs_release(wm_ctx *c) {
    {
        std::lock_guard<std::mutex> lock(g_gs_mutex);
        g_gs_free.push_back(c);
    }
    g_gs_cv.notify_one();
}
}  // namespace

extern "C" void generate_spectrogram(double *audio, double *output) {
    // lib.rs:34-40,113: the reflect pad is written into the CALLER's buffer (visible side effect).
    for (int i = 0; i < 200; ++i) {
        audio[i] = audio[400 - i];
        const int j = 16000 * 30 + i + 200;
        audio[j] = audio[200 + (16000 * 30 - 2) - i];
    }
    wm_ctx *ctx = gs_acquire();
    // The device kernel re-derives the same pad by index arithmetic from samples [200, 480200).
    int st = wm_logmel(ctx, audio + 200, WM_F64, 1, 80, output, WM_F64, WM_MEM_HOST);
    if (st != WM_OK) {
        fprintf(stderr, "generate_spectrogram: %s\n", wm_last_error());
        abort();
    }
    gs_release(ctx);
}
