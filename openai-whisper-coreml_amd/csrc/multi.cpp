// multi.cpp -- the 8-GPU split behind the C ABI (SURVEY.md 8b `device_ids[], n`, 8e): ONE host process that dlopens the
// library drives every GPU of the node.  Independent 30 s chunks are the natural shard of the path (the front end's only
// reduction is per chunk, stft/src/lib.rs:82-88; the reference never conditions one window on another,
// Whisper.swift:33-40), so
//   * weights are replicated: one wm_ctx per device (fill them through wm_multi_device_ctx + the usual weight calls);
//   * a call's chunks are cut into contiguous blocks, rank r owns [r ceil(N/R), min(N, (r+1) ceil(N/R)));
//   * one host thread per GPU runs the whole path of its block (front end -> encoder -> cross-K/V -> greedy decode);
//   * the ONLY exchange is one fixed-stride all-gather of int32 [ceil(N/R)][1 + max_new] (length, tokens) per rank over
//     RCCL (ncclCommInitAll: single process, xGMI between the GPUs) -- ~13.5 KB per rank for an hour of audio, latency-
//     bound; every rank ends with every stream, rank 0's copy is handed to the caller.
// bench.py --gpus N uses the one-process-per-GPU launch the driver prescribes (torch.distributed over the same RCCL);
// this is the same partition and the same collective for a dlopen-only host (host/multi_main.cpp).
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is NOT a link-time dependency (see rccl_api below)
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "model.h"

// RCCL is bound at run time, on the first wm_multi_create, never at link time:
//   * a front-end-only or single-GPU host (the reference's app calls generate_spectrogram and one model) must be able to
//     dlopen libwhisper_mi355x.so on a machine without librccl;
//   * a process that already carries an RCCL (bench.py --gpus N: torch's bundled librccl.so, same soname) must keep ONE
//     copy -- a hard DT_NEEDED on /opt/rocm/lib/librccl beside torch's put two builds behind one soname.
// Resolution order: an RCCL already mapped into the process (RTLD_NOLOAD), $WM_RCCL_PATH, then the usual search
// (this library's RUNPATH is /opt/rocm/lib).
namespace {
struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;   // why binding failed (empty when bound)
    bool ok = false;
};

const RcclApi &rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // the copy the process already has (torch's, the host's)
        if (!h)
            if (const char *e = getenv("WM_RCCL_PATH")) h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
        for (const char *n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char *e = dlerror();
            api.err = std::string("librccl.so.1 not found (set WM_RCCL_PATH): ") + (e ? e : "dlopen failed");
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(h, name);
            if (!p && api.err.empty()) api.err = std::string("librccl: missing symbol ") + name;
            return p;
        };
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.ok = api.err.empty();
    });
    return api;
}
}  // namespace

struct wm_multi {
    std::vector<wm_ctx *> ctx;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;
    std::vector<int32_t *> d_send, d_recv;
    size_t cap_rows = 0, cap_stride = 0;
};

extern "C" int wm_multi_partition(int n_chunks, int world_size, int rank, int *lo, int *hi) try {
    WM_REQUIRE(lo && hi && n_chunks >= 0 && world_size >= 1 && rank >= 0 && rank < world_size, WM_ERR_INVALID,
               "partition: bad arguments");
    const int per = (n_chunks + world_size - 1) / world_size;
    *lo = rank * per < n_chunks ? rank * per : n_chunks;
    *hi = (rank + 1) * per < n_chunks ? (rank + 1) * per : n_chunks;
    return WM_OK;
} WM_API_CATCH

// Host-side packing of the fixed-stride gather payload: row i of a rank's block = [len, tok_0 .. tok_{max_new-1}];
// rows past the block are zero.  wm_multi_unpack_tokens inverts it over the concatenation of all ranks' payloads.
extern "C" int wm_multi_pack_tokens(const int32_t *tokens, const int32_t *lens, int n_local, int per, int max_new,
                                    int32_t *payload) try {
    WM_REQUIRE(payload && n_local >= 0 && n_local <= per && max_new >= 0 && (n_local == 0 || (tokens && lens)),
               WM_ERR_INVALID, "pack_tokens: bad arguments");
    WM_REQUIRE((uint64_t)per * (1 + (uint64_t)max_new) <= (uint64_t)1 << 40, WM_ERR_INVALID, "pack_tokens: payload too large");
    const size_t stride = 1 + (size_t)max_new;
    memset(payload, 0, (size_t)per * stride * sizeof(int32_t));
    for (int i = 0; i < n_local; ++i)
        WM_REQUIRE(lens[i] >= 0 && lens[i] <= max_new, WM_ERR_INVALID, "pack_tokens: length %d of chunk %d outside [0, %d]", lens[i], i, max_new);
    for (int i = 0; i < n_local; ++i) {
        payload[i * stride] = lens[i];
        memcpy(payload + i * stride + 1, tokens + (size_t)i * max_new, (size_t)max_new * sizeof(int32_t));
    }
    return WM_OK;
} WM_API_CATCH

extern "C" int wm_multi_unpack_tokens(const int32_t *gathered, int world_size, int per, int max_new, int n_chunks,
                                      int32_t *tokens_out, int32_t *lens_out) try {
    WM_REQUIRE(gathered && tokens_out && lens_out && world_size >= 1 && per >= 0 && n_chunks >= 0 && max_new >= 0 &&
                   (long)n_chunks <= (long)world_size * per, WM_ERR_INVALID, "unpack_tokens: bad arguments");
    WM_REQUIRE((uint64_t)world_size * (uint64_t)per * (1 + (uint64_t)max_new) <= (uint64_t)1 << 40, WM_ERR_INVALID,
               "unpack_tokens: payload too large");
    const size_t stride = 1 + (size_t)max_new;
    // the payload was written by OTHER ranks: a length outside [0, max_new] is a corrupt exchange, not something to hand to
    // a caller who will index tokens_out with it (fuzz test: tests/test_fuzz_cpu.py)
    for (size_t c = 0; c < (size_t)n_chunks; ++c)
        WM_REQUIRE(gathered[c * stride] >= 0 && gathered[c * stride] <= max_new, WM_ERR_INVALID,
                   "unpack_tokens: chunk %zu carries length %d outside [0, %d]: corrupt token payload", c, gathered[c * stride], max_new);
    for (size_t c = 0; c < (size_t)n_chunks; ++c) {  // rank r's block starts at row r * per of the gathered buffer == chunk r * per
        lens_out[c] = gathered[c * stride];
        memcpy(tokens_out + c * (size_t)max_new, gathered + c * stride + 1, (size_t)max_new * sizeof(int32_t));
    }
    return WM_OK;
} WM_API_CATCH

extern "C" int wm_multi_create(const wm_dims *dims, const int *devices, int n, wm_multi **out) try {
    WM_REQUIRE(dims && devices && out && n >= 1 && n <= 64, WM_ERR_INVALID, "multi_create: bad arguments");
    *out = nullptr;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) WM_REQUIRE(devices[i] != devices[j], WM_ERR_INVALID, "multi_create: device %d listed twice", devices[i]);
    const RcclApi &nc = rccl_api();
    WM_REQUIRE(nc.ok, WM_ERR_STATE, "multi_create: RCCL is not available: %s", nc.err.c_str());
    wm_multi *m = new wm_multi();
    m->dev.assign(devices, devices + n);
    int st = WM_OK;
    for (int i = 0; i < n && st == WM_OK; ++i) {
        wm_ctx *c = nullptr;
        st = wm_create(dims, devices[i], &c);
        if (st == WM_OK) m->ctx.push_back(c);
    }
    if (st == WM_OK) {
        m->comm.resize(n);
        const ncclResult_t r = nc.CommInitAll(m->comm.data(), n, m->dev.data());
        if (r != ncclSuccess) {
            wm_set_error("ncclCommInitAll over %d device(s) failed: %s", n, nc.GetErrorString(r));
            m->comm.clear();
            st = WM_ERR_HIP;
        }
    }
    if (st != WM_OK) {
        const std::string keep = wm_last_error();
        wm_multi_destroy(m);
        wm_set_error("%s", keep.c_str());
        return st;
    }
    m->d_send.assign(n, nullptr);
    m->d_recv.assign(n, nullptr);
    *out = m;
    return WM_OK;
} WM_API_CATCH

extern "C" void wm_multi_destroy(wm_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        if (i < m->d_send.size() && m->d_send[i]) (void)hipFree(m->d_send[i]);
        if (i < m->d_recv.size() && m->d_recv[i]) (void)hipFree(m->d_recv[i]);
    }
    if (!m->comm.empty() && rccl_api().ok)
        for (ncclComm_t c : m->comm) (void)rccl_api().CommDestroy(c);
    for (wm_ctx *c : m->ctx) wm_destroy(c);
    delete m;
}

extern "C" int wm_multi_size(const wm_multi *m) { return m ? (int)m->ctx.size() : 0; }

extern "C" int wm_multi_device_ctx(wm_multi *m, int rank, wm_ctx **out) try {
    WM_REQUIRE(m && out && rank >= 0 && rank < (int)m->ctx.size(), WM_ERR_INVALID, "multi_device_ctx: bad rank");
    *out = m->ctx[rank];
    return WM_OK;
} WM_API_CATCH

static int ensure_gather_buffers(wm_multi *m, size_t per, size_t stride) {
    const size_t R = m->ctx.size();
    if (per * stride <= m->cap_rows * m->cap_stride && m->d_send[0]) return WM_OK;
    for (size_t r = 0; r < R; ++r) {
        WM_HIP(hipSetDevice(m->dev[r]));
        if (m->d_send[r]) WM_HIP(hipFree(m->d_send[r]));
        if (m->d_recv[r]) WM_HIP(hipFree(m->d_recv[r]));
        m->d_send[r] = m->d_recv[r] = nullptr;
        WM_HIP(hipMalloc((void **)&m->d_send[r], per * stride * sizeof(int32_t) + 256));
        WM_HIP(hipMalloc((void **)&m->d_recv[r], R * per * stride * sizeof(int32_t) + 256));
    }
    m->cap_rows = per;
    m->cap_stride = stride;
    return WM_OK;
}

extern "C" int wm_multi_transcribe_greedy(wm_multi *m, const void *pcm, wm_dtype pcm_dtype, int B, const int32_t *prompt,
                                          int n_prompt, int max_new, int32_t eot, int32_t *tokens_out, int32_t *lens_out) try {
    WM_REQUIRE(m && pcm && prompt && tokens_out && lens_out && B >= 1 && max_new >= 1, WM_ERR_INVALID,
               "multi_transcribe_greedy: bad arguments");
    WM_REQUIRE(pcm_dtype == WM_I16 || pcm_dtype == WM_F32 || pcm_dtype == WM_F64, WM_ERR_INVALID, "bad pcm dtype");
    const int R = (int)m->ctx.size();
    WM_REQUIRE(R >= 1 && m->ctx[0]->model, WM_ERR_STATE, "multi_transcribe_greedy: no model");
    {   // every size that feeds an allocation below is checked against the model BEFORE any thread exists
        const wm_dims &D = m->ctx[0]->model->dims;
        WM_REQUIRE(n_prompt >= 1 && n_prompt + max_new <= D.n_text_ctx, WM_ERR_INVALID,
                   "prompt (%d) + new tokens (%d) must fit the %d-token context", n_prompt, max_new, D.n_text_ctx);
    }
    const RcclApi &nc = rccl_api();
    WM_REQUIRE(nc.ok, WM_ERR_STATE, "RCCL is not available: %s", nc.err.c_str());
    const int per = (B + R - 1) / R;
    const size_t stride = 1 + (size_t)max_new;
    const size_t elem = pcm_dtype == WM_I16 ? 2 : pcm_dtype == WM_F32 ? 4 : 8;
    WM_TRY(ensure_gather_buffers(m, (size_t)per, stride));
    // ---- one host thread per GPU: the whole path of its block of chunks
    std::vector<int> status(R, WM_OK);
    std::vector<std::string> errs(R);
    std::vector<std::vector<int32_t>> payload(R, std::vector<int32_t>((size_t)per * stride, 0));
    std::vector<std::thread> th;
    th.reserve(R);
    // A worker's body never lets an exception escape (it would std::terminate the host process, not return through the
    // FFI); if a std::thread cannot be started, the ones already running are joined before the error is reported.
    struct Joiner {
        std::vector<std::thread> &t;
        ~Joiner() {
            for (auto &x : t)
                if (x.joinable()) x.join();
        }
    } joiner{th};
    for (int r = 0; r < R; ++r) {
        th.emplace_back([&, r] {
          try {
            int lo = 0, hi = 0;
            (void)wm_multi_partition(B, R, r, &lo, &hi);
            const int n_local = hi - lo;
            std::vector<int32_t> tok((size_t)(n_local > 0 ? n_local : 1) * max_new), len(n_local > 0 ? n_local : 1);
            int st = WM_OK;
            if (n_local > 0)
                st = wm_transcribe_greedy(m->ctx[r], (const char *)pcm + (size_t)lo * WM_N_SAMPLES * elem, pcm_dtype, n_local,
                                          prompt, n_prompt, max_new, eot, tok.data(), len.data(), WM_MEM_HOST);
            if (st == WM_OK) st = wm_multi_pack_tokens(tok.data(), len.data(), n_local, per, max_new, payload[r].data());
            if (st == WM_OK) {
                if (hipSetDevice(m->dev[r]) != hipSuccess ||
                    hipMemcpyAsync(m->d_send[r], payload[r].data(), payload[r].size() * sizeof(int32_t), hipMemcpyHostToDevice,
                                   m->ctx[r]->stream) != hipSuccess ||
                    hipStreamSynchronize(m->ctx[r]->stream) != hipSuccess) {
                    wm_set_error("rank %d: staging the token payload failed", r);
                    st = WM_ERR_HIP;
                }
            }
            status[r] = st;
            if (st != WM_OK) errs[r] = wm_last_error();  // thread-local: hand it to the caller's thread
          } catch (const std::bad_alloc &) {
            status[r] = WM_ERR_NOMEM;
            try { errs[r] = "out of host memory"; } catch (...) {}
          } catch (...) {
            status[r] = WM_ERR_NOMEM;
            try { errs[r] = "internal error in the worker thread"; } catch (...) {}
          }
        });
    }
    for (auto &t : th) t.join();
    for (int r = 0; r < R; ++r)
        if (status[r] != WM_OK) {
            wm_set_error("rank %d (device %d): %s", r, m->dev[r], errs[r].c_str());
            return status[r];
        }
    // ---- the one collective of the job: fixed-stride all-gather of the token streams (RCCL over xGMI)
    ncclResult_t nr = nc.GroupStart();
    for (int r = 0; r < R && nr == ncclSuccess; ++r)
        nr = nc.AllGather(m->d_send[r], m->d_recv[r], (size_t)per * stride, ncclInt32, m->comm[r], m->ctx[r]->stream);
    if (nr == ncclSuccess) nr = nc.GroupEnd();
    else (void)nc.GroupEnd();
    WM_REQUIRE(nr == ncclSuccess, WM_ERR_HIP, "ncclAllGather failed: %s", nc.GetErrorString(nr));
    std::vector<int32_t> all((size_t)R * per * stride);
    WM_HIP(hipSetDevice(m->dev[0]));
    WM_HIP(hipMemcpyAsync(all.data(), m->d_recv[0], all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, m->ctx[0]->stream));
    WM_HIP(hipStreamSynchronize(m->ctx[0]->stream));
    for (int r = 1; r < R; ++r) {  // every rank received every stream: drain the others' streams too
        WM_HIP(hipSetDevice(m->dev[r]));
        WM_HIP(hipStreamSynchronize(m->ctx[r]->stream));
    }
    return wm_multi_unpack_tokens(all.data(), R, per, max_new, B, tokens_out, lens_out);
} WM_API_CATCH
