/*
 * oracle/logmel_ref.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, f64 restatement of the reference's log-mel front end,
 *   /root/reference/stft/src/lib.rs:22-122
 * exporting the SAME C symbol the reference exports (lib.rs:110-111, declared to
 * Swift at Whisper/Whisper/bridge.h:11):
 *
 *     void generate_spectrogram(double *audio [480400, mutated], double *out [240000]);
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline.  The product library
 * (libwhisper_mi355x.so) never links or calls anything in oracle/.
 *
 * PARITY PINNING.  The Rust crate cannot be compiled here (no cargo/rustc, crates not
 * vendored) and the reference has no tests, so there are no reference-held golden
 * outputs.  This restatement is pinned by
 *   (1) the reference's own data artefact stft/src/m80.npy (sha256 3cd88cce...de9580),
 *       embedded below from mel80.inc;
 *   (2) analytic known-answer tests derived from lib.rs (tests/test_oracle_logmel.py:
 *       zeros => -1.5 everywhere; DC => 1.5988866134681166 / 1.3247581029878088 /
 *       -0.40111338653188344; bin-centred cosine; reflect == np.pad(...,"reflect"));
 *   (3) agreement with an independent numpy restatement (oracle/logmel_np.py, numpy's
 *       pocketfft instead of this file's own FFT) to <= 1e-12.
 * The third-party FFT the reference calls (realfft 3.0.1 -> rustfft 6.0.1,
 * stft/Cargo.lock:143-163; call sites lib.rs:23-24,44-45) is an unnormalised forward
 * real DFT; any f64 DFT agrees with it to ~1e-13 relative, which is far inside the
 * 1e-9 abs gate used for the ABI-exact GPU path.
 *
 * Every function cites the reference lines it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N_SAMPLES 480000 /* 16000 * 30                      lib.rs:37,112 */
#define N_PADDED 480400  /* + 200 each side                 lib.rs:112    */
#define N_FFT 400        /*                                 lib.rs:24,26  */
#define HOP 160          /* stride_size                     lib.rs:50     */
#define N_BINS 201       /*                                 lib.rs:51     */
#define N_FRAMES 3000    /* (0..len-400).step_by(160)       lib.rs:52     */

static const float MEL80[80 * 201] = {
#include "mel80.inc"
};

/* ---- one-time state: Hann window + twiddles (lib.rs:22-32) ------------------------- */
static double g_window[N_FFT];
static double g_tw_re[N_FFT], g_tw_im[N_FFT]; /* exp(-2*pi*i*k/400) */
static int g_init = 0;

static void oracle_init(void) {
    if (g_init) return;
    const double PI = 3.14159265358979323846264338327950288;
    for (int i = 0; i < N_FFT; ++i) {
        /* lib.rs:26  ((i as f64 * 2.0 * PI) / 400.0) -> (1.0 - cos) / 2.0  (periodic Hann) */
        double a = ((double)i * 2.0 * PI) / 400.0;
        g_window[i] = (1.0 - cos(a)) / 2.0;
        g_tw_re[i] = cos(a);
        g_tw_im[i] = -sin(a);
    }
    g_init = 1;
}

/* ---- reflect pad, in place on the caller's buffer (lib.rs:34-40) ------------------- */
static void reflect(double *audio) {
    for (int i = 0; i < 200; ++i) {
        audio[i] = audio[400 - i];
        int j = 16000 * 30 + i + 200;
        audio[j] = audio[200 + (16000 * 30 - 2) - i];
    }
}

/* ---- complex mixed-radix FFT whose length n divides 400 ---------------------------- */
/* Decimation in time, radix r in {5,4,2} chosen per level; tw_step = 400 / n so that
 * W_n^j = table[(j * tw_step) mod 400].  Stands in for the rustfft plan the reference
 * builds at lib.rs:23-24. */
static void fft_rec(const double *in_re, const double *in_im, int in_stride, double *out_re,
                    double *out_im, int n, int tw_step) {
    if (n == 1) {
        out_re[0] = in_re[0];
        out_im[0] = in_im[0];
        return;
    }
    const int r = (n % 5 == 0) ? 5 : (n % 4 == 0) ? 4 : 2;
    const int m = n / r;
    for (int q = 0; q < r; ++q) /* r sub-transforms over x[q], x[q+r], ... */
        fft_rec(in_re + q * in_stride, in_im + q * in_stride, in_stride * r, out_re + q * m,
                out_im + q * m, m, tw_step * r);
    for (int k = 0; k < m; ++k) { /* X[k + m p] = sum_q W_n^{q k} W_r^{q p} Y_q[k] */
        double yr[5], yi[5], tr[5], ti[5];
        for (int q = 0; q < r; ++q) {
            const int t = (q * k * tw_step) % N_FFT;
            const double ar = out_re[q * m + k], ai = out_im[q * m + k];
            yr[q] = ar * g_tw_re[t] - ai * g_tw_im[t];
            yi[q] = ar * g_tw_im[t] + ai * g_tw_re[t];
        }
        for (int p = 0; p < r; ++p) {
            double sr = 0.0, si = 0.0;
            for (int q = 0; q < r; ++q) {
                const int t = (((q * p) % r) * m * tw_step) % N_FFT; /* W_r^{q p} */
                sr += yr[q] * g_tw_re[t] - yi[q] * g_tw_im[t];
                si += yr[q] * g_tw_im[t] + yi[q] * g_tw_re[t];
            }
            tr[p] = sr;
            ti[p] = si;
        }
        for (int p = 0; p < r; ++p) {
            out_re[p * m + k] = tr[p];
            out_im[p * m + k] = ti[p];
        }
    }
}

/* ---- windowed 400-point real DFT of one frame -> 201 power bins (lib.rs:42-47,54) -- */
/* realfft's even-length algorithm: pack x[2n] + i x[2n+1] into a 200-point complex FFT,
 * then split.  use_naive != 0 selects a direct O(N^2) DFT used only to validate this. */
static int g_use_naive = 0;
void oracle_set_naive_dft(int on) { g_use_naive = on; }

static void frame_power(const double *frame, double *power) {
    double w[N_FFT];
    for (int n = 0; n < N_FFT; ++n) w[n] = frame[n] * g_window[n]; /* lib.rs:43 */
    if (g_use_naive) {
        for (int k = 0; k < N_BINS; ++k) {
            double sr = 0.0, si = 0.0;
            for (int n = 0; n < N_FFT; ++n) {
                const int t = (k * n) % N_FFT;
                sr += w[n] * g_tw_re[t];
                si += w[n] * g_tw_im[t];
            }
            power[k] = sr * sr + si * si; /* norm_sqr, lib.rs:54 */
        }
        return;
    }
    double zr[200], zi[200], Zr[200], Zi[200];
    for (int n = 0; n < 200; ++n) {
        zr[n] = w[2 * n];
        zi[n] = w[2 * n + 1];
    }
    fft_rec(zr, zi, 1, Zr, Zi, 200, 2);
    for (int k = 0; k <= 200; ++k) {
        /* X[k] = E[k] + W_400^k O[k],  E = (Z[k] + conj Z[200-k]) / 2,
         *                               O = (Z[k] - conj Z[200-k]) / (2i)          */
        const int a = k % 200, b = (200 - k) % 200;
        const double er = 0.5 * (Zr[a] + Zr[b]), ei = 0.5 * (Zi[a] - Zi[b]);
        const double orr = 0.5 * (Zi[a] + Zi[b]), oi = -0.5 * (Zr[a] - Zr[b]);
        const int t = k % N_FFT;
        const double xr = er + (orr * g_tw_re[t] - oi * g_tw_im[t]);
        const double xi = ei + (orr * g_tw_im[t] + oi * g_tw_re[t]);
        power[k] = xr * xr + xi * xi; /* norm_sqr, lib.rs:54 */
    }
}

/* ---- SpectrogramGenerator::spectrogram (lib.rs:49-102), n_mels generalised --------- */
/* `audio` is the reflect-padded 480400-sample buffer; `filt` is [n_mels][201] f32,
 * mel-major (lib.rs:65 indexes MELS[i*201+k]); `out` is [n_mels][3000] row-major
 * (lib.rs:117-121). */
static void spectrogram(const double *audio, const float *filt, int n_mels, double *out) {
    /* lib.rs:51-58: power spectrum stored transposed, [201][3000] */
    double *spec = (double *)malloc(sizeof(double) * N_BINS * N_FRAMES);
    int j = 0;
    for (int i = 0; i < N_PADDED - N_FFT; i += HOP, ++j) { /* lib.rs:52: exclusive bound */
        double p[N_BINS];
        frame_power(audio + i, p);
        for (int k = 0; k < N_BINS; ++k) spec[(size_t)k * N_FRAMES + j] = p[k];
    }
    /* lib.rs:60-69: dense mel projection, k ascending, f64 accumulate, f32 filter widened */
    for (int i = 0; i < n_mels; ++i)
        for (int f = 0; f < N_FRAMES; ++f) {
            double sum = 0.0;
            for (int k = 0; k < N_BINS; ++k)
                sum += spec[(size_t)k * N_FRAMES + f] * (double)filt[i * N_BINS + k];
            out[(size_t)i * N_FRAMES + f] = sum;
        }
    free(spec);
    const size_t total = (size_t)n_mels * N_FRAMES;
    /* lib.rs:71-79: x.max(1e-10).log10()   (Rust f64::max returns the non-NaN operand) */
    for (size_t t = 0; t < total; ++t) {
        double x = out[t];
        x = (x > 1e-10) ? x : 1e-10; /* NaN > 1e-10 is false => 1e-10, as f64::max */
        out[t] = log10(x);
    }
    /* lib.rs:82-88: global max over every value of this chunk */
    double gmax = out[0];
    for (size_t t = 1; t < total; ++t)
        if (out[t] > gmax) gmax = out[t];
    /* lib.rs:91-99: (x.max(gmax - 8.0) + 4.0) / 4.0 */
    const double floor_v = gmax - 8.0;
    for (size_t t = 0; t < total; ++t) {
        double x = out[t];
        x = (x > floor_v) ? x : floor_v;
        out[t] = (x + 4.0) / 4.0;
    }
}

/* ---- the reference's FFI entry (lib.rs:110-122; bridge.h:11) ----------------------- */
static void generate_spectrogram_impl(double *audio, double *output) {
    oracle_init();
    reflect(audio);                          /* lib.rs:113 (mutates the caller's buffer) */
    spectrogram(audio, MEL80, 80, output);   /* lib.rs:114-121 */
}
/* The exported name is the reference's; internal callers use the static _impl so that a
 * process which also has the PRODUCT library's generate_spectrogram loaded can never have
 * one interposed for the other. */
void generate_spectrogram(double *audio, double *output) { generate_spectrogram_impl(audio, output); }

/* Same pipeline with a caller-supplied filterbank (128 mels for large-v3). */
void oracle_generate_spectrogram_filt(double *audio, double *output, const float *filt,
                                      int n_mels) {
    oracle_init();
    reflect(audio);
    spectrogram(audio, filt, n_mels, output);
}

/* ---- batched convenience used by the cpu_baseline leg of bench.py ------------------- */
/* pcm: [n_chunks][480000] f32 (the Swift caller's Float samples, ContentView.swift:57-60
 * widened to f64 and given 200 zeros each side, stft.swift:10-11); out: f64
 * [n_chunks][80][3000].  Single-threaded, as the Rust crate is (no rayon in lib.rs). */
void oracle_logmel_batch_f32(const float *pcm, int n_chunks, double *out) {
    double *buf = (double *)malloc(sizeof(double) * N_PADDED);
    for (int c = 0; c < n_chunks; ++c) {
        memset(buf, 0, sizeof(double) * N_PADDED);
        for (int i = 0; i < N_SAMPLES; ++i) buf[200 + i] = (double)pcm[(size_t)c * N_SAMPLES + i];
        generate_spectrogram_impl(buf, out + (size_t)c * 80 * N_FRAMES);
    }
    free(buf);
}

/* The same, chunk-parallel over all host cores (OpenMP): NOT how the reference runs (its crate is single-threaded), but
 * the fairest CPU number for a batch of independent chunks (SURVEY.md 8d asks for both; bench.py labels them). */
void oracle_logmel_batch_f32_omp(const float *pcm, int n_chunks, double *out) {
#pragma omp parallel
    {
        double *buf = (double *)malloc(sizeof(double) * N_PADDED);
#pragma omp for schedule(dynamic, 1)
        for (int c = 0; c < n_chunks; ++c) {
            memset(buf, 0, sizeof(double) * N_PADDED);
            for (int i = 0; i < N_SAMPLES; ++i) buf[200 + i] = (double)pcm[(size_t)c * N_SAMPLES + i];
            generate_spectrogram_impl(buf, out + (size_t)c * 80 * N_FRAMES);
        }
        free(buf);
    }
}

const float *oracle_mel80(void) { return MEL80; }
