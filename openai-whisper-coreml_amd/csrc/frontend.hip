// frontend.hip -- log-mel front end for gfx950 (replaces the Rust `stft` crate,
// /root/reference/stft/src/lib.rs:22-122).
//
// Per 30 s chunk:  reflect-index (lib.rs:34-40) -> 3000 frames, hop 160, window 400
// (lib.rs:52) -> periodic Hann (lib.rs:26,43) -> 400-point real DFT (lib.rs:44-45) ->
// power (lib.rs:54) -> mel projection (lib.rs:60-69) -> log10(max(.,1e-10)) (lib.rs:71-79)
// -> per-chunk max (lib.rs:82-88) -> (max(x, gmax-8)+4)/4 (lib.rs:91-99), output
// [n_mels][3000] row-major (lib.rs:116-121).
//
// Two arithmetic types from one template:
//   double : the ABI-exact path (generate_spectrogram; <= 1e-9 abs vs the f64 oracle)
//   float  : the fast path      (<= 1e-4 abs vs the f64 oracle)
//
// Kernel 1 (logmel_stage1): one workgroup owns FPB = 16*WPB consecutive frames of one
// chunk.  The PCM span those frames cover is staged ONCE in LDS (coalesced HBM reads,
// every sample read from HBM once per workgroup instead of 2.5x), with the reflect pad
// done by index arithmetic.  The DFT uses two symmetries.  Window: w[n] == w[400-n], w[0]==0:
//     Re X[k] =  sum_{n=1..200} e[n] cos(2 pi k n/400),  e[n] = w[n] (x[n] + x[400-n])  (n=200 once)
//     Im X[k] = -sum_{n=1..199} o[n] sin(2 pi k n/400),  o[n] = w[n] (x[n] - x[400-n])
// Bin pairs: cos(2 pi (200-k) n/400) = (-1)^n cos(2 pi k n/400), sin(..) = -(-1)^n sin(..), so with the
// sums split over even and odd n (Ce, Co, Se, So) bins k and 200-k come from the SAME four sums:
//     X[k] = (Ce+Co) - i(Se+So),   X[200-k] = (Ce-Co) - i(So-Se),   k = 0..100.
// That is four [16 frames x 100] x [100 x 101] products per wave (a quarter of the naive DFT's
// multiply-adds), issued on the matrix pipe with the exact-f32 / f64 MFMA (v_mfma_f32_16x16x4_f32 /
// v_mfma_f64_16x16x4_f64: an fmaf chain, no reduced precision); A operand built on the fly from the
// LDS span, B operand = the L2-resident twiddle tables.  Power -> LDS -> banded mel (the filterbank is
// 97.6% zeros; summing the non-zero band in ascending k equals the reference's dense
// ascending-k sum exactly) -> log10 -> store + per-chunk atomic max.
// Kernel 2 (logmel_stage2): dynamic-range clamp + affine, in place.
#include "wm_internal.h"

#include <math.h>
#include <string.h>

#include <vector>

namespace {

constexpr int NJT = 7;           // 7 * 16 = 112 >= 101 paired bins (k = 0..100)
constexpr int TW_COLS = NJT * 16;
constexpr int TW_ROWS = 100;     // even n = 2..200 / odd n = 1..199
constexpr int PW_STRIDE = 209;   // odd stride: conflict-free column walks

template <typename T>
struct Acc;
template <>
struct Acc<float> {
    typedef __attribute__((ext_vector_type(4))) float type;
};
template <>
struct Acc<double> {
    typedef __attribute__((ext_vector_type(4))) double type;
};

__device__ __forceinline__ Acc<float>::type mfma4(float a, float b, Acc<float>::type c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ Acc<double>::type mfma4(double a, double b, Acc<double>::type c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// C/D row of accumulator register r for this lane (col is lane & 15 for both types).
__device__ __forceinline__ int acc_row(float, int lane, int r) { return (lane >> 4) * 4 + r; }
__device__ __forceinline__ int acc_row(double, int lane, int r) { return (lane >> 4) + 4 * r; }

// f32 path: log10 as log2 x log10(2) -- v_log_f32 (1 ulp) and one multiply instead of the ~30-instruction libm expansion, 20-32
// times per lane; |error| <= 2e-7 relative, four orders below the path's 1e-4 gate (round 6; both f32 kernels share it)
__device__ __forceinline__ float log10_t(float x) { return __builtin_amdgcn_logf(x) * 0.30102999566398120f; }
__device__ __forceinline__ double log10_t(double x) { return log10(x); }

// Monotone encodings so an unsigned atomicMax orders floating-point values.
__device__ __forceinline__ unsigned long long enc_max(float v) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return (unsigned long long)u;
}
__device__ __forceinline__ unsigned long long enc_max(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ void dec_max(unsigned long long e, float *out) {
    unsigned u = (unsigned)e;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    *out = __uint_as_float(u);
}
__device__ __forceinline__ void dec_max(unsigned long long e, double *out) {
    unsigned long long u = (e & 0x8000000000000000ull) ? (e & 0x7fffffffffffffffull) : ~e;
    *out = __longlong_as_double((long long)u);
}

template <typename T>
__device__ __forceinline__ T load_sample(const void *pcm, int dtype, size_t idx) {
    if (dtype == WM_I16) return (T)((const short *)pcm)[idx] * (T)(1.0 / 32768.0);
    if (dtype == WM_F32) return (T)((const float *)pcm)[idx];
    return (T)((const double *)pcm)[idx];
}

// LDS address of span sample p (p = local_frame * 160 + n): one pad word per 160 samples
// so that the 16 frames a wave reads at the same n fall in different banks.
__device__ __forceinline__ int span_addr(int frame, int n) { return frame * 161 + n + n / 160; }

template <typename T, int WPB>
__global__ __launch_bounds__(WPB * 64) void logmel_stage1(
    const void *__restrict__ pcm, int pcm_dtype, int n_mels, const T *__restrict__ tw,
    const T *__restrict__ /*unused*/, const T *__restrict__ win, const int *__restrict__ band_start,
    const int *__restrict__ band_len, const float *__restrict__ band_w, T *__restrict__ out,
    unsigned long long *__restrict__ gmax, int blocks_per_chunk) {
    constexpr int FPB = 16 * WPB;
    constexpr int SPAN = (FPB - 1) * WM_HOP + WM_N_FFT;
    constexpr int SPAN_LDS = SPAN + SPAN / 160 + 4;
    constexpr int PW_LDS = WPB * 16 * PW_STRIDE;
    // one LDS region: the PCM span during the DFT, the power spectrum afterwards
    __shared__ T smem[SPAN_LDS > PW_LDS ? SPAN_LDS : PW_LDS];
    T *xs = smem;
    T(*pw)[16][PW_STRIDE] = (T(*)[16][PW_STRIDE])smem;
    // non-finite power bins per frame.  The reference's mel product is DENSE (lib.rs:60-69 multiplies every bin by its
    // weight, zero or not), so a non-finite bin that meets a zero weight turns the row into NaN (0 x Inf), which
    // lib.rs:76 then floors to 1e-10; the banded sum below reproduces that from this count.
    __shared__ int bad[WPB][16];
    if (threadIdx.x < WPB * 16) bad[threadIdx.x >> 4][threadIdx.x & 15] = 0;

    const int chunk = blockIdx.x / blocks_per_chunk;
    const int f0 = (blockIdx.x % blocks_per_chunk) * FPB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- stage the PCM span (reflect pad by index; lib.rs:34-40) -----------------------
    const size_t chunk_base = (size_t)chunk * WM_N_SAMPLES;
    for (int i = tid; i < SPAN; i += WPB * 64) {
        int n = f0 * WM_HOP + i - 200;                     // index into the unpadded chunk
        if (n < 0) n = -n;                                 // a[i] = a[400 - i]
        if (n >= WM_N_SAMPLES) n = 2 * (WM_N_SAMPLES - 1) - n;  // a[j] = a[200 + (N-2) - i]
        if (n < 0) n = 0;                                  // only for masked frames >= 3000
        xs[i + i / 160] = load_sample<T>(pcm, pcm_dtype, chunk_base + n);
    }
    __syncthreads();

    // ---- DFT of 16 frames per wave on the matrix pipe ------------------------------------
    typedef typename Acc<T>::type acc_t;
    acc_t acc[4][NJT];  // Ce, Co, Se, So
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NJT; ++j) acc[t][j] = (acc_t){0, 0, 0, 0};
    const int fr = wave * 16 + (lane & 15);  // A row: frame within the block
    const int kq = lane >> 4;                // A col / B row within the 4-deep k step
    const int col = lane & 15;
    for (int kk = 0; kk < TW_ROWS / 4; ++kk) {
        const int ie = 4 * kk + kq;          // table row 0..99
        const int ne = 2 * (ie + 1);         // even n: 2..200
        const int no = 2 * ie + 1;           // odd n: 1..199
        const T x1e = xs[span_addr(fr, ne)], x2e = xs[span_addr(fr, WM_N_FFT - ne)];
        const T x1o = xs[span_addr(fr, no)], x2o = xs[span_addr(fr, WM_N_FFT - no)];
        const T we = win[ne], wo = win[no];
        const T a_ce = we * ((ne == 200) ? x1e : (x1e + x2e));
        const T a_se = we * (x1e - x2e);
        const T a_co = wo * (x1o + x2o);
        const T a_so = wo * (x1o - x2o);
        const T *row = tw + (size_t)ie * TW_COLS + col;
#pragma unroll
        for (int j = 0; j < NJT; ++j) {
            acc[0][j] = mfma4(a_ce, row[0 * TW_ROWS * TW_COLS + j * 16], acc[0][j]);
            acc[1][j] = mfma4(a_co, row[1 * TW_ROWS * TW_COLS + j * 16], acc[1][j]);
            acc[2][j] = mfma4(a_se, row[2 * TW_ROWS * TW_COLS + j * 16], acc[2][j]);
            acc[3][j] = mfma4(a_so, row[3 * TW_ROWS * TW_COLS + j * 16], acc[3][j]);
        }
    }
    __syncthreads();  // every wave is done with the PCM span: its LDS becomes the power spectrum
    // power (lib.rs:54) -> LDS, [frame][bin]; bins k and 200-k from the same four sums
#pragma unroll
    for (int j = 0; j < NJT; ++j) {
        const int k = j * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const T ce = acc[0][j][r], co = acc[1][j][r], se = acc[2][j][r], so = acc[3][j][r];
            const int row = acc_row(T(0), lane, r);
            int nbad = 0;
            if (k <= 100) {
                const T re = ce + co, im = se + so;
                const T pv = re * re + im * im;
                pw[wave][row][k] = pv;
                nbad += !(pv - pv == (T)0);  // Inf - Inf and NaN - NaN are NaN
            }
            if (k < 100) {
                const T re = ce - co, im = so - se;
                const T pv = re * re + im * im;
                pw[wave][row][200 - k] = pv;
                nbad += !(pv - pv == (T)0);
            }
            if (nbad) atomicAdd(&bad[wave][row], nbad);
        }
    }
    __syncthreads();

    // ---- banded mel + log10 + per-chunk max (lib.rs:60-88) -------------------------------
    const int frame = f0 + wave * 16 + (lane & 15);
    const bool live = frame < WM_N_FRAMES;
    T vmax = (T)-1e30;
    for (int m = lane >> 4; m < n_mels; m += 4) {
        const int ks = band_start[m], kl = band_len[m];
        const float *wrow = band_w + m * WM_MEL_MAXW;
        T s = 0;
        int inband = 0;
        for (int t = 0; t < kl; ++t) {
            const T pv = pw[wave][lane & 15][ks + t];
            s += pv * (T)wrow[t];
            inband += !(pv - pv == (T)0);
        }
        // a non-finite bin outside this row's band would have met a zero weight in the reference's dense sum: NaN
        if (bad[wave][lane & 15] > inband) s = (T)0;
        s = (s > (T)1e-10) ? s : (T)1e-10;  // f64::max(1e-10): NaN -> 1e-10
        const T v = log10_t(s);
        if (live) {
            out[((size_t)chunk * n_mels + m) * WM_N_FRAMES + frame] = v;
            vmax = (v > vmax) ? v : vmax;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(vmax, off);
        vmax = (o > vmax) ? o : vmax;
    }
    if (lane == 0 && vmax > (T)-1e29) atomicMax(gmax + chunk, enc_max(vmax));
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the f32 fast path's stage 1 with the twiddle slices SHARED through LDS.  The template above has every wave
// fetch its own B operands from the L2-resident tables -- 28 global loads per lane and k-step, the whole 179 KB table per 16
// frames: 1.9 GB of L2 traffic per 56-chunk launch, which is what bound the kernel (453.9 us: a quarter of the exact-f32
// MFMA rate its 700 MFMAs per 16 frames allow).  Here the 7-KiB slice of a k-step (4 tables x 4 rows x 112 columns) is
// fetched ONCE per workgroup of 4 waves = 64 frames (448 x 16 bytes), double-buffered in LDS one k-step
// ahead, and read by every wave as ds_read_b32 fragments (conflict-free: rows are 112 words apart, 112 mod 32 = 16); the
// window is in LDS too.  Same A operands, same MFMA chains in the same order, same epilogue: bit-identical to the template
// (test).  LDS: max(PCM span, power spectrum) + 2 slices + window + mel bands = 76.5 KiB: TWO workgroups per CU, so one's
// power / mel / log epilogue (no matrix work) runs under the other's DFT, and 2 688 workgroups at 56 chunks are 5.25 rounds of
// the chip instead of the 6 that 1 344 eight-wave ones were (the first version of this kernel: 303 us).
// What the stall counters said the kernel was really waiting for (profiles/r06_pmc_frontend_stalls.txt: the matrix pipe busy
// 22 % of a wave's life, 42 % parked at s_waitcnt) were two LATENCY chains outside the DFT, and both are gone here:
//   * the PCM span was staged one 2-byte load per thread and pass, ~41 dependent HBM round trips per wave: now every thread
//     requests its whole share up front as 16-byte loads (8 int16 / 4 f32 samples; the blocks that touch the reflect pad or
//     the end of the chunk, 2 of 24, gather by index -- still all loads before the first store);
//   * the mel loop fetched band start / length / weights from global memory per row (three dependent L2 round trips x 20-32
//     rows per lane): the band tables are copied to LDS once per workgroup.
constexpr int W8 = 4;                                   // waves per workgroup: 64 frames, TWO workgroups per CU (one's epilogue under the other's DFT)
constexpr int FPB8 = 16 * W8;
constexpr int SPAN8 = (FPB8 - 1) * WM_HOP + WM_N_FFT;   // 10 480 samples
constexpr int SPAN8_LDS = SPAN8 + SPAN8 / 160 + 4;
constexpr int PW8_STRIDE = 201;                         // odd (conflict-free column walks), >= 201 bins
constexpr int PW8_LDS = W8 * 16 * PW8_STRIDE;           // 12 864 words
constexpr int MAIN8 = SPAN8_LDS > PW8_LDS ? SPAN8_LDS : PW8_LDS;
constexpr int SLICE8 = 4 * 4 * TW_COLS;                 // 1 792 words per k-step
constexpr int BW8 = 16;                                 // band weights kept per mel row in LDS (80 mels: <= 14, 128 mels: <= 10; checked by the host)
constexpr int BAND8 = 128 * (BW8 + 2);                  // band start / length / weights of up to 128 mel rows
constexpr int LDS8_WORDS = MAIN8 + 2 * SLICE8 + 404 + W8 * 16 + BAND8;
constexpr int LDS8_BYTES = LDS8_WORDS * 4;              // 76.5 KiB: two workgroups per CU

__global__ __launch_bounds__(W8 * 64, 2) void logmel_stage1_f32_lds(
    const void *__restrict__ pcm, int pcm_dtype, int n_mels, const float *__restrict__ tw, const float *__restrict__ win,
    const int *__restrict__ band_start, const int *__restrict__ band_len, const float *__restrict__ band_w,
    float *__restrict__ out, unsigned long long *__restrict__ gmax, int blocks_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float smem8[];
    float *xs = smem8;
    float(*pw)[16][PW8_STRIDE] = (float(*)[16][PW8_STRIDE])smem8;
    float *slice = smem8 + MAIN8;               // [2][4 tables][4 rows][112]
    float *wl = slice + 2 * SLICE8;             // window, 401 values
    int(*bad)[16] = (int(*)[16])(wl + 404);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < W8 * 16) bad[tid >> 4][tid & 15] = 0;
    const int chunk = blockIdx.x / blocks_per_chunk;
    const int f0 = (blockIdx.x % blocks_per_chunk) * FPB8;
    // slice 0 requested first: 448 x 16 bytes (element e: table t = e / 112, float4 e % 112 of rows 4 kk .. + 3), thread tid
    // fetches elements tid and tid + 256
    const bool loader2 = tid + 256 < 448;
    const float *tsrc0 = tw + (size_t)(tid / 112) * TW_ROWS * TW_COLS + (tid % 112) * 4;
    const float *tsrc1 = tw + (size_t)((tid + 256) / 112) * TW_ROWS * TW_COLS + ((tid + 256) % 112) * 4;
    float4 tnext0 = *(const float4 *)tsrc0;
    float4 tnext1 = loader2 ? *(const float4 *)tsrc1 : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i <= 400; i += W8 * 64) wl[i] = win[i];
    float *bnd = (float *)(bad + W8);           // [n_mels] start | [n_mels] length (as ints) | [n_mels][BW8] weights
    int *bnd_i = (int *)bnd;
    for (int i = tid; i < n_mels; i += W8 * 64) { bnd_i[i] = band_start[i]; bnd_i[128 + i] = band_len[i]; }
    for (int i = tid; i < n_mels * BW8; i += W8 * 64) bnd[256 + i] = band_w[(i / BW8) * WM_MEL_MAXW + (i % BW8)];
    // ---- stage the PCM span (reflect pad by index; lib.rs:34-40): all loads of a thread first, then its stores ----------
    const size_t chunk_base = (size_t)chunk * WM_N_SAMPLES;
    constexpr int PER = 8, NPASS = (SPAN8 / PER + W8 * 64 - 1) / (W8 * 64);   // 1310 groups of 8 samples, 6 passes of 256 threads
    static_assert(SPAN8 % PER == 0, "the span is a whole number of 8-sample groups");
    const int first = f0 * WM_HOP - 200;        // unpadded-chunk index of span sample 0
    // (workgroup-uniform) no reflect, no masked frames -- and a caller's buffer that is 16-byte aligned (chunk offsets and
    // `first` are multiples of 16 bytes for int16 and f32; an odd base pointer takes the gather path)
    const bool interior = first >= 0 && first + SPAN8 <= WM_N_SAMPLES && ((size_t)pcm & 15) == 0;
    float v[NPASS][PER];
    if (interior && pcm_dtype == WM_I16) {      // 16 bytes = 8 samples per load; first * 2 and chunk_base * 2 are multiples of 16
        const short *src = (const short *)pcm + chunk_base + first;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int g = ps * W8 * 64 + tid;
            const uint4 raw = g * PER < SPAN8 ? *(const uint4 *)(src + g * PER) : make_uint4(0, 0, 0, 0);
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[ps][2 * u] = (float)(short)(w[u] & 0xffffu) * (float)(1.0 / 32768.0);
                v[ps][2 * u + 1] = (float)(short)(w[u] >> 16) * (float)(1.0 / 32768.0);
            }
        }
    } else if (interior && pcm_dtype == WM_F32) {
        const float *src = (const float *)pcm + chunk_base + first;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int g = ps * W8 * 64 + tid;
            const bool ok = g * PER < SPAN8;
            const float4 a = ok ? *(const float4 *)(src + g * PER) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b = ok ? *(const float4 *)(src + g * PER + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[ps][0] = a.x; v[ps][1] = a.y; v[ps][2] = a.z; v[ps][3] = a.w;
            v[ps][4] = b.x; v[ps][5] = b.y; v[ps][6] = b.z; v[ps][7] = b.w;
        }
    } else {                                    // edge blocks / f64 input: gather by (reflected) index
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int g = ps * W8 * 64 + tid;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                int n = first + g * PER + u;
                if (n < 0) n = -n;                                 // a[i] = a[400 - i]
                if (n >= WM_N_SAMPLES) n = 2 * (WM_N_SAMPLES - 1) - n;  // a[j] = a[200 + (N-2) - i]
                if (n < 0) n = 0;                                  // only for masked frames >= 3000
                if (n >= WM_N_SAMPLES) n = WM_N_SAMPLES - 1;       // (groups past the span: never stored)
                v[ps][u] = load_sample<float>(pcm, pcm_dtype, chunk_base + n);
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int i0 = (ps * W8 * 64 + tid) * PER;
        if (i0 < SPAN8) {
            const int q = i0 / 160, r = i0 - q * 160;   // pad word per 160 samples: i + i / 160
#pragma unroll
            for (int u = 0; u < PER; ++u) xs[i0 + u + q + (r + u >= 160 ? 1 : 0)] = v[ps][u];
        }
    }
    *(float4 *)(slice + tid * 4) = tnext0;
    if (loader2) *(float4 *)(slice + (tid + 256) * 4) = tnext1;
    __syncthreads();

    typedef Acc<float>::type acc_t;
    acc_t acc[4][NJT];  // Ce, Co, Se, So
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NJT; ++j) acc[t][j] = (acc_t){0, 0, 0, 0};
    const int fr = wave * 16 + (lane & 15);
    const int kq = lane >> 4, col = lane & 15;
    for (int kk = 0; kk < TW_ROWS / 4; ++kk) {
        if (kk + 1 < TW_ROWS / 4) {
            tnext0 = *(const float4 *)(tsrc0 + (size_t)(kk + 1) * 4 * TW_COLS);
            if (loader2) tnext1 = *(const float4 *)(tsrc1 + (size_t)(kk + 1) * 4 * TW_COLS);
        }
        const int ie = 4 * kk + kq;
        const int ne = 2 * (ie + 1), no = 2 * ie + 1;
        const float x1e = xs[span_addr(fr, ne)], x2e = xs[span_addr(fr, WM_N_FFT - ne)];
        const float x1o = xs[span_addr(fr, no)], x2o = xs[span_addr(fr, WM_N_FFT - no)];
        const float we = wl[ne], wo = wl[no];
        const float a_ce = we * ((ne == 200) ? x1e : (x1e + x2e));
        const float a_se = we * (x1e - x2e);
        const float a_co = wo * (x1o + x2o);
        const float a_so = wo * (x1o - x2o);
        const float *row = slice + (kk & 1) * SLICE8 + kq * TW_COLS + col;
#pragma unroll
        for (int j = 0; j < NJT; ++j) {
            acc[0][j] = mfma4(a_ce, row[0 * 4 * TW_COLS + j * 16], acc[0][j]);
            acc[1][j] = mfma4(a_co, row[1 * 4 * TW_COLS + j * 16], acc[1][j]);
            acc[2][j] = mfma4(a_se, row[2 * 4 * TW_COLS + j * 16], acc[2][j]);
            acc[3][j] = mfma4(a_so, row[3 * 4 * TW_COLS + j * 16], acc[3][j]);
        }
        // the next slice goes into the OTHER buffer (read in k-step kk - 1: everybody passed the barrier since)
        if (kk + 1 < TW_ROWS / 4) {
            *(float4 *)(slice + ((kk + 1) & 1) * SLICE8 + tid * 4) = tnext0;
            if (loader2) *(float4 *)(slice + ((kk + 1) & 1) * SLICE8 + (tid + 256) * 4) = tnext1;
        }
        __syncthreads();
    }
    // (the loop's last barrier: every wave is done with the PCM span -- its LDS becomes the power spectrum)
#pragma unroll
    for (int j = 0; j < NJT; ++j) {
        const int k = j * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ce = acc[0][j][r], co = acc[1][j][r], se = acc[2][j][r], so = acc[3][j][r];
            const int row = acc_row(0.f, lane, r);
            int nbad = 0;
            if (k <= 100) {
                const float re = ce + co, im = se + so;
                const float pv = re * re + im * im;
                pw[wave][row][k] = pv;
                nbad += !(pv - pv == 0.f);
            }
            if (k < 100) {
                const float re = ce - co, im = so - se;
                const float pv = re * re + im * im;
                pw[wave][row][200 - k] = pv;
                nbad += !(pv - pv == 0.f);
            }
            if (nbad) atomicAdd(&bad[wave][row], nbad);
        }
    }
    __syncthreads();
    const int frame = f0 + wave * 16 + (lane & 15);
    const bool live = frame < WM_N_FRAMES;
    float vmax = -1e30f;
    for (int m = lane >> 4; m < n_mels; m += 4) {
        const int ks = bnd_i[m], kl = bnd_i[128 + m];
        const float *wrow = bnd + 256 + m * BW8;
        float s = 0;
        int inband = 0;
        for (int t = 0; t < kl; ++t) {
            const float pv = pw[wave][lane & 15][ks + t];
            s += pv * wrow[t];
            inband += !(pv - pv == 0.f);
        }
        if (bad[wave][lane & 15] > inband) s = 0.f;
        s = (s > 1e-10f) ? s : 1e-10f;
        const float v = log10_t(s);
        if (live) {
            out[((size_t)chunk * n_mels + m) * WM_N_FRAMES + frame] = v;
            vmax = (v > vmax) ? v : vmax;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(vmax, off);
        vmax = (o > vmax) ? o : vmax;
    }
    if (lane == 0 && vmax > -1e29f) atomicMax(gmax + chunk, enc_max(vmax));
}

template <typename T>
__global__ __launch_bounds__(256) void logmel_stage2(T *__restrict__ io,
                                                     const unsigned long long *__restrict__ gmax,
                                                     int per_chunk, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        T g;
        dec_max(gmax[i / per_chunk], &g);
        const T fl = g - (T)8.0;
        T x = io[i];
        x = (x > fl) ? x : fl;         // x.max(gmax - 8.0)   lib.rs:96
        io[i] = (x + (T)4.0) / (T)4.0; // (.. + 4.0) / 4.0
    }
}

template <typename T>
int upload(T **dptr, const std::vector<T> &h, hipStream_t s) {
    WM_HIP(hipMalloc((void **)dptr, h.size() * sizeof(T)));
    WM_HIP(hipMemcpyAsync(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    WM_HIP(hipStreamSynchronize(s));
    return WM_OK;
}

}  // namespace

// ------------------------------------------------------------------ mel filterbanks -----
static const float kMel80[80 * 201] = {
#include "mel80.inc"
};
const float *wm_mel80_table() { return kMel80; }

// librosa.filters.mel(sr=16000, n_fft=400, n_mels, htk=False, norm="slaney") -- the recipe
// openai-whisper's assets/mel_filters.npz was made with (export_m80.py:4 reads "mel_80").
void wm_mel_filterbank(int n_mels, std::vector<float> &out) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = 1000.0 / f_sp;
    const double logstep = log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) {
        return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
    };
    auto mel_to_hz = [&](double m) {
        return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
    };
    const int nb = WM_N_BINS;
    std::vector<double> fft_f(nb), mel_f(n_mels + 2);
    for (int k = 0; k < nb; ++k) fft_f[k] = 8000.0 * k / (nb - 1);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
    for (int i = 0; i < n_mels + 2; ++i)
        mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    out.assign((size_t)n_mels * nb, 0.0f);
    for (int i = 0; i < n_mels; ++i) {
        const double d0 = mel_f[i + 1] - mel_f[i], d1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < nb; ++k) {
            const double lower = (fft_f[k] - mel_f[i]) / d0;
            const double upper = (mel_f[i + 2] - fft_f[k]) / d1;
            double w = lower < upper ? lower : upper;
            if (w < 0) w = 0;
            // librosa stores the triangle in an f32 array, then scales it in place
            const float w32 = (float)w;
            out[(size_t)i * nb + k] = (float)((double)w32 * enorm);
        }
    }
}

static int build_bands(const float *filt, int n_mels, int **d_start, int **d_len, float **d_w,
                       hipStream_t s, int *max_w) {
    *max_w = 0;
    std::vector<int> st(n_mels), ln(n_mels);
    std::vector<float> w((size_t)n_mels * WM_MEL_MAXW, 0.0f);
    for (int m = 0; m < n_mels; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < WM_N_BINS; ++k)
            if (filt[m * WM_N_BINS + k] != 0.0f) {
                if (lo < 0) lo = k;
                hi = k;
            }
        if (lo < 0) {
            lo = 0;
            hi = -1;
        }
        WM_REQUIRE(hi - lo + 1 <= WM_MEL_MAXW, WM_ERR_INVALID, "mel band %d too wide", m);
        st[m] = lo;
        ln[m] = hi - lo + 1;
        if (ln[m] > *max_w) *max_w = ln[m];
        for (int k = lo; k <= hi; ++k) w[(size_t)m * WM_MEL_MAXW + (k - lo)] = filt[m * WM_N_BINS + k];
    }
    WM_TRY(upload(d_start, st, s));
    WM_TRY(upload(d_len, ln, s));
    WM_TRY(upload(d_w, w, s));
    return WM_OK;
}

int wm_frontend_init(WmFrontend *fe, hipStream_t stream) {
    if (fe->ready) return WM_OK;
    const double PI = 3.14159265358979323846264338327950288;
    // four tables [cosE | cosO | sinE | sinO], each [100 rows][TW_COLS]: row i of the even tables is
    // n = 2(i+1), of the odd tables n = 2i+1; column k = 0..100
    std::vector<double> c64((size_t)4 * TW_ROWS * TW_COLS, 0.0), s64(1, 0.0), w64(401);
    // exact argument reduction: angle index (k*n) mod 400, table of 400 f64 values
    std::vector<double> ct(400), stb(400);
    for (int t = 0; t < 400; ++t) {
        ct[t] = cos(2.0 * PI * t / 400.0);
        stb[t] = sin(2.0 * PI * t / 400.0);
    }
    // exact zeros / ones where the angle is a multiple of pi/2
    ct[100] = 0.0; ct[300] = 0.0; stb[0] = 0.0; stb[200] = 0.0;
    for (int i = 0; i < TW_ROWS; ++i)
        for (int k = 0; k <= 100; ++k) {
            const int te = (2 * (i + 1) * k) % 400, to = ((2 * i + 1) * k) % 400;
            c64[((size_t)0 * TW_ROWS + i) * TW_COLS + k] = ct[te];
            c64[((size_t)1 * TW_ROWS + i) * TW_COLS + k] = ct[to];
            c64[((size_t)2 * TW_ROWS + i) * TW_COLS + k] = stb[te];
            c64[((size_t)3 * TW_ROWS + i) * TW_COLS + k] = stb[to];
        }
    for (int i = 0; i <= 400; ++i) w64[i] = (1.0 - cos(((double)i * 2.0 * PI) / 400.0)) / 2.0;  // lib.rs:26
    std::vector<float> c32(c64.begin(), c64.end()), s32(s64.begin(), s64.end()),
        w32(w64.begin(), w64.end());
    WM_TRY(upload(&fe->cos64, c64, stream));
    WM_TRY(upload(&fe->sin64, s64, stream));
    WM_TRY(upload(&fe->win64, w64, stream));
    WM_TRY(upload(&fe->cos32, c32, stream));
    WM_TRY(upload(&fe->sin32, s32, stream));
    WM_TRY(upload(&fe->win32, w32, stream));
    WM_TRY(build_bands(kMel80, 80, &fe->band_start[0], &fe->band_len[0], &fe->band_w[0], stream, &fe->band_maxw[0]));
    std::vector<float> m128;
    wm_mel_filterbank(128, m128);
    WM_TRY(build_bands(m128.data(), 128, &fe->band_start[1], &fe->band_len[1], &fe->band_w[1],
                       stream, &fe->band_maxw[1]));
    fe->ready = true;
    return WM_OK;
}

void wm_frontend_destroy(WmFrontend *fe) {
    void *ptrs[] = {fe->cos32, fe->sin32, fe->win32, fe->cos64, fe->sin64, fe->win64,
                    fe->band_start[0], fe->band_start[1], fe->band_len[0], fe->band_len[1],
                    fe->band_w[0], fe->band_w[1], fe->gmax, fe->scratch};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    *fe = WmFrontend();
}

int wm_frontend_run(WmFrontend *fe, WmProfiler *prof, hipStream_t stream, const void *d_pcm,
                    wm_dtype pcm_dtype, int n_chunks, int n_mels, void *d_out,
                    wm_dtype out_dtype) {
    WM_REQUIRE(fe->ready, WM_ERR_STATE, "front end not initialised");
    WM_REQUIRE(n_mels == 80 || n_mels == 128, WM_ERR_INVALID, "n_mels must be 80 or 128, got %d", n_mels);
    WM_REQUIRE(pcm_dtype == WM_I16 || pcm_dtype == WM_F32 || pcm_dtype == WM_F64, WM_ERR_INVALID,
               "pcm dtype must be i16/f32/f64");
    WM_REQUIRE(out_dtype == WM_F32 || out_dtype == WM_F64, WM_ERR_INVALID, "out dtype must be f32/f64");
    if (n_chunks == 0) return WM_OK;
    WM_REQUIRE(n_chunks > 0 && d_pcm && d_out, WM_ERR_INVALID, "bad n_chunks / null pointer");
    const int fi = (n_mels == 80) ? 0 : 1;
    if (fe->gmax_cap < n_chunks) {
        if (fe->gmax) WM_HIP(hipFree(fe->gmax));
        fe->gmax = nullptr;
        WM_HIP(hipMalloc(&fe->gmax, sizeof(unsigned long long) * n_chunks));
        fe->gmax_cap = n_chunks;
    }
    WM_HIP(hipMemsetAsync(fe->gmax, 0, sizeof(unsigned long long) * n_chunks, stream));
    const size_t total = (size_t)n_chunks * n_mels * WM_N_FRAMES;
    const int g2 = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (out_dtype == WM_F32) {
        if (g_wm_tuning.frontend_per_wave_twiddles || fe->band_maxw[fi] > BW8) {   // probes / the bit-identity test: the round-1-5 kernel (also: a filterbank with bands wider than the LDS copy holds)
            constexpr int WPB = 4;
            const int bpc = (WM_N_FRAMES + 16 * WPB - 1) / (16 * WPB);
            WmProfScope ps(prof, "logmel_stage1_f32", stream);
            logmel_stage1<float, WPB><<<n_chunks * bpc, WPB * 64, 0, stream>>>(
                d_pcm, (int)pcm_dtype, n_mels, fe->cos32, fe->sin32, fe->win32, fe->band_start[fi],
                fe->band_len[fi], fe->band_w[fi], (float *)d_out, (unsigned long long *)fe->gmax, bpc);
        } else {
            static std::atomic<int> attr_set[64];   // per device: the kernel's dynamic-LDS allowance is set once
            int dev = 0;
            WM_HIP(hipGetDevice(&dev));
            if (!attr_set[dev & 63].load(std::memory_order_acquire)) {
                WM_HIP(hipFuncSetAttribute((const void *)logmel_stage1_f32_lds, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8_BYTES));
                attr_set[dev & 63].store(1, std::memory_order_release);
            }
            const int bpc = (WM_N_FRAMES + FPB8 - 1) / FPB8;
            WmProfScope ps(prof, "logmel_stage1_f32", stream);
            logmel_stage1_f32_lds<<<n_chunks * bpc, W8 * 64, LDS8_BYTES, stream>>>(
                d_pcm, (int)pcm_dtype, n_mels, fe->cos32, fe->win32, fe->band_start[fi], fe->band_len[fi],
                fe->band_w[fi], (float *)d_out, (unsigned long long *)fe->gmax, bpc);
        }
        {
            WmProfScope ps(prof, "logmel_stage2_f32", stream);
            logmel_stage2<float><<<g2, 256, 0, stream>>>((float *)d_out,
                                                         (const unsigned long long *)fe->gmax,
                                                         n_mels * WM_N_FRAMES, total);
        }
    } else {
        constexpr int WPB = 2;
        const int bpc = (WM_N_FRAMES + 16 * WPB - 1) / (16 * WPB);
        {
            WmProfScope ps(prof, "logmel_stage1_f64", stream);
            logmel_stage1<double, WPB><<<n_chunks * bpc, WPB * 64, 0, stream>>>(
                d_pcm, (int)pcm_dtype, n_mels, fe->cos64, fe->sin64, fe->win64, fe->band_start[fi],
                fe->band_len[fi], fe->band_w[fi], (double *)d_out, (unsigned long long *)fe->gmax, bpc);
        }
        {
            WmProfScope ps(prof, "logmel_stage2_f64", stream);
            logmel_stage2<double><<<g2, 256, 0, stream>>>((double *)d_out,
                                                          (const unsigned long long *)fe->gmax,
                                                          n_mels * WM_N_FRAMES, total);
        }
    }
    WM_HIP(hipGetLastError());
    return WM_OK;
}
