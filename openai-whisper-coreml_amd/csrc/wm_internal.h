// wm_internal.h -- private declarations shared by the translation units of
// libwhisper_mi355x.so (gfx950 only; nothing here is part of the public C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/whisper_mi355x.h"

// ---------------------------------------------------------------- error plumbing -----
void wm_set_error(const char *fmt, ...);

#define WM_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            wm_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,               \
                         hipGetErrorString(_e));                                          \
            return WM_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define WM_TRY(expr)                                                                      \
    do {                                                                                  \
        int _s = (expr);                                                                  \
        if (_s != WM_OK) return _s;                                                       \
    } while (0)

#define WM_REQUIRE(cond, code, ...)                                                       \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            wm_set_error(__VA_ARGS__);                                                    \
            return (code);                                                                \
        }                                                                                 \
    } while (0)

// Every int-returning entry point of the C ABI is a function-try-block closed by this: no C++ exception (bad_alloc /
// length_error from a std::vector sized by caller- or file-supplied numbers, ...) ever crosses the FFI.
#define WM_API_CATCH                                                                      \
    catch (const std::bad_alloc &) {                                                      \
        wm_set_error("out of host memory");                                               \
        return WM_ERR_NOMEM;                                                              \
    }                                                                                     \
    catch (const std::exception &e) {                                                     \
        wm_set_error("internal error: %s", e.what());                                     \
        return WM_ERR_NOMEM;                                                              \
    }                                                                                     \
    catch (...) {                                                                         \
        wm_set_error("internal error (unknown exception)");                               \
        return WM_ERR_NOMEM;                                                              \
    }

// ---------------------------------------------------------------- fixed geometry -----
// Literals of the reference front end (stft/src/lib.rs:24,26,35-37,50-52,112,116).
constexpr int WM_N_SAMPLES = 480000;  // 16000 * 30
constexpr int WM_N_FFT = 400;
constexpr int WM_HOP = 160;
constexpr int WM_N_BINS = 201;
constexpr int WM_N_FRAMES = 3000;
constexpr int WM_MEL_MAXW = 32;  // widest supported filter band (80 mels: 14)

typedef unsigned short bf16_t;  // raw bf16 bits in HBM

// ---------------------------------------------------------------- profiling ----------
struct WmProfFamily {
    double ms = 0.0;
    long n = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct WmProfiler {
    bool on = false;
    std::map<std::string, WmProfFamily> fam;
    std::vector<hipEvent_t> pool;
    hipEvent_t get();
    void begin(const char *name, hipStream_t s, hipEvent_t *e0);
    void end(const char *name, hipStream_t s, hipEvent_t e0);
    void drain();
    void reset();
    ~WmProfiler();
};

struct WmProfScope {
    WmProfiler *p;
    const char *name;
    hipStream_t s;
    hipEvent_t e0 = nullptr;
    WmProfScope(WmProfiler *p_, const char *n, hipStream_t s_) : p(p_), name(n), s(s_) {
        if (p && p->on) p->begin(name, s, &e0);
    }
    ~WmProfScope() {
        if (p && p->on && e0) p->end(name, s, e0);
    }
};

// ---------------------------------------------------------------- front end ----------
struct WmFrontend {
    // DFT-as-MFMA tables: rows n = 1..200, 208 columns (bins 0..200, rest zero).
    float *cos32 = nullptr, *sin32 = nullptr, *win32 = nullptr;
    double *cos64 = nullptr, *sin64 = nullptr, *win64 = nullptr;
    // banded mel filters per supported n_mels (index 0: 80, index 1: 128)
    int *band_start[2] = {nullptr, nullptr};
    int *band_len[2] = {nullptr, nullptr};
    float *band_w[2] = {nullptr, nullptr};  // [n_mels][WM_MEL_MAXW]
    int band_maxw[2] = {0, 0};              // widest band of the filterbank (the LDS-resident copy of the f32 kernel holds 16)
    void *gmax = nullptr;                   // per-chunk encoded maxima (u64 per chunk)
    int gmax_cap = 0;
    void *scratch = nullptr;                // staging for host-pointer calls
    size_t scratch_bytes = 0;
    bool ready = false;
};

int wm_frontend_init(WmFrontend *fe, hipStream_t stream);
void wm_frontend_destroy(WmFrontend *fe);
// Device-pointer core: pcm [n][480000] (dtype) -> out [n][n_mels][3000] (f32 or f64).
int wm_frontend_run(WmFrontend *fe, WmProfiler *prof, hipStream_t stream, const void *d_pcm,
                    wm_dtype pcm_dtype, int n_chunks, int n_mels, void *d_out,
                    wm_dtype out_dtype);
// Host-side slaney mel filterbank generator (librosa.filters.mel semantics; used for
// n_mels = 128 and validated against the reference's m80.npy at n_mels = 80).
void wm_mel_filterbank(int n_mels, std::vector<float> &out /* [n_mels][201] */);
const float *wm_mel80_table();

// ---------------------------------------------------------------- context ------------
struct WmModel;  // model.h
struct wm_ctx;

// Model-call overrides a DEBUG build can install on a context (libwhisper_mi355x_dbg.so: wmdbg_set_precision -> the
// all-fp32 path of f32_path.hip).  The product library has no code that sets them: always null there.
struct WmDebugHooks {
    int (*encode_dev)(wm_ctx *ctx, const float *d_mel, int B, float *d_xa);
    int (*decode_logits_dev)(wm_ctx *ctx, const int32_t *host_tokens /*[B][T]*/, int B, int T, const float *d_xa,
                             float *d_logits /*[B][T][n_vocab]*/);
};

struct wm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    WmProfiler prof;
    WmFrontend fe;
    WmModel *model = nullptr;
    float stage_ms[3] = {0, 0, 0};
    std::vector<wm_ctx *> lanes;  // weight-sharing clones owned by this context (wm_transcribe_greedy)
    const WmDebugHooks *dbg_hooks = nullptr;
    int max_lanes = 0;            // wm_set_lanes: decode groups in flight per wm_transcribe_greedy call (0: $WM_LANES, default 3)
    // SUB-CHIP LANES (round 6).  A context whose stream was created with a CU mask (hipExtStreamCreateWithCUMask) runs
    // everything it launches on `n_cus` of the 256 CUs: the CUs [cu_lo, cu_hi) of every XCD.  Two or
    // three SMALL decode groups (latency-bound launch chains) then run side by side on disjoint CUs instead of flooding
    // all 256 CUs with every launch and serialising at the CU level (model_api.cpp, the group policy).
    int n_cus = 256, cu_lo = 0, cu_hi = 32;   // the CUs [cu_lo, cu_hi) of every XCD
    std::vector<wm_ctx *> part_lanes[2];   // [0]: the two clones of the 2-way partition, [1]: the three of the 3-way one (created on first use)
    bool no_cu_masks = false;              // a CU-masked stream could not be created on this device: the policy stays unmasked
    std::map<int, wm_ctx *> solo_lanes;    // probes (debug knob lane_solo_cus): a clone confined to the first n CUs of every XCD
};
// weight-sharing clone of `parent` whose stream is confined to the CUs [cu_lo, cu_hi) of every XCD (api.cpp)
int wm_clone_cus(wm_ctx *parent, int cu_lo, int cu_hi, wm_ctx **out);
// CU mask (256 bits) of the CUs [cu_lo, cu_hi) of every XCD; returns the CU count
int wm_cu_mask(int cu_lo, int cu_hi, uint32_t mask[8]);

int wm_ctx_make_current(const wm_ctx *ctx);
int wm_group_count(int B, int L, bool explicit_lanes, int gc_probe);   // model_api.cpp: decode groups of a wm_transcribe_greedy call
// CU-masked groups of a call (0: none -- unmasked lanes as wm_group_count says; 2 / 3: that many groups, one per part of the chip)
int wm_lane_parts(int B, int L, bool explicit_lanes, int n_text_state, int n_text_layer);

// ---------------------------------------------------------------- launch-shape experiment knobs
// Launch shapes are chosen by fixed rules (dec_kernels.hip pick_shape / wm_dec_attn_splits, gemm.hip wm_gemm): the
// product library reads NO environment variable for them, so nothing outside the process's own calls can change a launch
// shape in a library whose contract is bit-level batch invariance.  The A/B probes of tools/ change these fields through
// wmdbg_set_tuning(), which exists in libwhisper_mi355x_dbg.so only (debug_hooks.cpp).  A value of 0 / the default below
// means "the product's rule".
struct WmTuning {
    int gemv_tn = 0;              // force tiles per workgroup of the wide decode GEMVs (1, 2, 4)
    int gemv_nblk = 0;            // 1: one batch block per workgroup also above 16 rows
    int gemv_ppw2_nblk = 0;       // probes: 1 / 2 = force one / two batch blocks per workgroup in the two-parts-per-wave K = 4d residual product (0: the rule)
    int gemv_no_ppw2 = 0;         // 1: K = 4d residual product as 16-wave workgroups (no two-parts-per-wave kernel)
    int prefetch_max_b = 16;      // L2 warm-up workgroups up to this decode-group size (0: never)
    int xattn_split_below = 96;   // (sequence, head) pairs below which the cross-attention streams are dealt flat
    int xattn_wgs = 256;          // workgroup cap of the cross-attention
    int xattn_no_flat = 0;        // 1: split launches as (pair, split) grids instead of the flat deal
    int xattn_lds_pad = 84 * 1024;  // dynamic LDS reserved per cross-attention workgroup (one per CU chip-wide); 0: off
    int xattn_splits = 0;         // force the split count of the cross-attention (1, 2, 4, 8)
    int gemm_tile = 0;            // force the encoder GEMM tile (128 / 256)
    int gemm_gm = 4;              // grouped tile order of the encoder GEMM
    int gemm128_pipe = 0;         // PROBE: the 128 x 128 tile's 3-stage pipeline kernel: 0 = the rule (grids <= 2 rounds of the chip), 1 = never, 2 = always
    int no_early_stop = 0;        // 1: decode every position and truncate on the host (the round-2 behaviour)
    int logits_tn = 0;            // 1 / 2: tiles per workgroup of the logits product at <= 16 rows (product: 4)
    int enc_attn_mfma_sum = 0;    // 1: encoder attention row sums by a ones-operand MFMA instead of f32 VALU adds
    int xattn_never_short = 0;    // 1: persistent cross-attention workgroups also when the chip is shared (rounds 2-3)
    int xattn_no_deep = 0;        // 1: the flat (few-pair) cross-attention walks its blocks one round trip at a time
    int xattn_pair_wg_max_pairs = 0;  // EXPERIMENT (lost, same file): alone, 257 .. this many pairs: one cross-attention workgroup per pair, two per CU
    int xattn_fuse_q = 1;         // 96 .. 256 pairs, alone: query projection fused into the cross-attention launch (0: two launches)
    int argmax_rows_per_wg = 0;   // PROBE: rows per workgroup of the step-closing arg-max (0 = the product's rule: 1, or 16 for <= 16 rows with early stop)
    int group_chunks = 0;         // preferred decode-group size of a wm_transcribe_greedy call (product rule: model_api.cpp)
    int frontend_per_wave_twiddles = 0;   // 1: the f32 front end's round-1-5 stage-1 kernel (every wave fetches its own twiddles from L2)
    int lane_parts = 0;           // sub-chip lanes: 0 = the product's rule, 1 = never, 2 / 3 = that many CU-masked groups whenever the call has >= 2 chunks per part
    int lane_solo_cus = 0;        // PROBE: n in 1 .. 31 = run the call's decode groups one after the other on ONE lane confined to the first n CUs of every XCD
};
extern WmTuning g_wm_tuning;   // api.cpp

// wm_transcribe_greedy calls in flight per device (any context, any host thread): a decode group that shares the chip
// with others launches its cross-attention as short-lived workgroups (model.h WmModel::xattn_shared).
#include <atomic>
extern std::atomic<int> g_wm_active_decodes[64];   // api.cpp
