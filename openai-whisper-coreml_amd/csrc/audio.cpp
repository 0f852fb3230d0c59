// audio.cpp -- the step immediately BEFORE the hot path, behind the C ABI: a 16 kHz mono 16-bit
// RIFF/WAVE file in, independent 30 s windows out.  Host only, no GPU.
//
// The reference records 16 kHz mono LinearPCM to `query.wav` (Whisper/Whisper/AudioRecorder.swift:56-61), reads it back
// through AVFoundation (:74-86) and pads / truncates to ONE 30 s window (Whisper/Whisper/ContentView.swift:57-60).  A
// dlopen-only host has no AVFoundation: these functions give it the same input path, and apply the reference's pad rule
// per window so that a recording of any length becomes the list of chunks wm_transcribe_greedy shards over.
// (openai-whisper-coreml_amd/audio.py is the Python twin; tests pin one against the other.)
#include <stdio.h>
#include <string.h>

#include <vector>

#include "wm_internal.h"

struct wm_wav {
    std::vector<unsigned char> raw;  // the file image (kept as read: ONE copy of the recording, whatever its size)
    size_t data_off = 0;             // first byte of the data chunk
    size_t n_samples = 0;
};

namespace {
uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
constexpr size_t kMaxWavBytes = (size_t)1 << 32;  // 4 GiB: > 37 h of 16 kHz int16; RIFF cannot describe more anyway
}  // namespace

extern "C" int wm_wav_open(const char *path, wm_wav **out) try {
    WM_REQUIRE(path && out, WM_ERR_INVALID, "wav_open: null pointer");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    WM_REQUIRE(f, WM_ERR_IO, "cannot open '%s'", path);
    std::vector<unsigned char> buf;
    {
        unsigned char tmp[1 << 16];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) {
            if (buf.size() + n > kMaxWavBytes) { fclose(f); wm_set_error("'%s': larger than 4 GiB", path); return WM_ERR_IO; }
            buf.insert(buf.end(), tmp, tmp + n);
        }
    }
    fclose(f);
    WM_REQUIRE(buf.size() >= 12 && memcmp(buf.data(), "RIFF", 4) == 0 && memcmp(buf.data() + 8, "WAVE", 4) == 0, WM_ERR_IO,
               "'%s' is not a RIFF/WAVE file", path);
    bool have_fmt = false;
    size_t pos = 12, data_off = 0, data_len = 0;
    while (pos + 8 <= buf.size()) {
        const unsigned char *h = buf.data() + pos;
        size_t len = rd32(h + 4);
        const size_t body = pos + 8;
        if (memcmp(h, "fmt ", 4) == 0) {
            WM_REQUIRE(len >= 16 && body + 16 <= buf.size(), WM_ERR_IO, "'%s': truncated fmt chunk", path);
            const unsigned char *p = buf.data() + body;
            unsigned fmt = rd16(p);
            const unsigned ch = rd16(p + 2), rate = rd32(p + 4), bits = rd16(p + 14);
            if (fmt == 0xFFFE && len >= 26 && body + 26 <= buf.size()) fmt = rd16(p + 24);  // WAVE_FORMAT_EXTENSIBLE: sub-format
            WM_REQUIRE(fmt == 1 && ch == 1 && rate == 16000 && bits == 16, WM_ERR_IO,
                       "'%s': expected 16 kHz mono 16-bit PCM, got format %u, %u Hz, %u ch, %u-bit", path, fmt, rate, ch, bits);
            have_fmt = true;
        } else if (memcmp(h, "data", 4) == 0) {
            if (len == 0xFFFFFFFFu || body + len > buf.size()) len = buf.size() - body;  // streamed / truncated: what is there
            data_off = body;
            data_len = len;
            break;
        }
        pos = body + len + (len & 1);  // chunks are word-aligned
    }
    WM_REQUIRE(have_fmt, WM_ERR_IO, "'%s': no fmt chunk before the data", path);
    WM_REQUIRE(data_off != 0, WM_ERR_IO, "'%s': no data chunk", path);
    wm_wav *w = new wm_wav();
    w->raw.swap(buf);   // no second copy: samples are decoded (little endian -> host int16) window by window on read
    w->data_off = data_off;
    w->n_samples = data_len / 2;
    *out = w;
    return WM_OK;
} WM_API_CATCH

extern "C" void wm_wav_close(wm_wav *w) { delete w; }

extern "C" long wm_wav_num_samples(const wm_wav *w) { return w ? (long)w->n_samples : 0; }

// ceil(n / 480000), at least one window (an empty recording is one silent chunk -- sharding.chunk_pcm's rule)
extern "C" int wm_wav_num_chunks(const wm_wav *w) {
    if (!w) return 0;
    const size_t n = w->n_samples;
    return n == 0 ? 1 : (int)((n + WM_N_SAMPLES - 1) / WM_N_SAMPLES);
}

extern "C" int wm_wav_read_chunks(const wm_wav *w, int first_chunk, int n_chunks, int16_t *out) try {
    WM_REQUIRE(w && out && first_chunk >= 0 && n_chunks >= 0 && first_chunk + (long)n_chunks <= wm_wav_num_chunks(w),
               WM_ERR_INVALID, "wav_read_chunks: windows [%d, %d) outside the recording's %d", first_chunk,
               first_chunk + n_chunks, wm_wav_num_chunks(w));
    const size_t n = w->n_samples;
    for (int c = 0; c < n_chunks; ++c) {
        const size_t lo = (size_t)(first_chunk + c) * WM_N_SAMPLES;
        const size_t have = lo < n ? (n - lo < (size_t)WM_N_SAMPLES ? n - lo : (size_t)WM_N_SAMPLES) : 0;
        int16_t *dst = out + (size_t)c * WM_N_SAMPLES;
        const unsigned char *src = w->raw.data() + w->data_off + 2 * lo;
        for (size_t i = 0; i < have; ++i) dst[i] = (int16_t)rd16(src + 2 * i);   // little endian, whatever the host
        memset(dst + have, 0, ((size_t)WM_N_SAMPLES - have) * sizeof(int16_t));  // ContentView.swift:57-60: zero-pad
    }
    return WM_OK;
} WM_API_CATCH
