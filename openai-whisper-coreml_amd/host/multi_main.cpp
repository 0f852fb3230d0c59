// multi_main.cpp -- dlopen-only C++ host driving ALL GPUs of the node through ONE library in ONE process: the loading
// model BASELINE.json's north star prescribes for the Swift host (`.so` via dlopen), extended to the 8-GPU chunk split of
// SURVEY.md 8e.  No torch, no launcher: wm_multi_create(devices[], n) -> weights on every device -> one
// wm_multi_transcribe_greedy over a recording cut into 30 s chunks (ContentView.swift:57-60's pad rule per window).
//
// usage: multi_main <libwhisper_mi355x.so> <model> <n_gpus> <n_chunks | recording.wav> [max_new]
// prints: one line per chunk "chunk i: len tok tok ...", then "identical_to_single_gpu 1|0" (the same chunks through a
// plain single-device wm_transcribe_greedy on device 0) and the wall time.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "whisper_mi355x.h"

#define LOAD(name) auto name = (decltype(&::name))dlsym(lib, #name); if (!name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <libwhisper_mi355x.so> <tiny.en|base|small|large-v2|large-v3> <n_gpus> <n_chunks> [max_new]\n", argv[0]);
        return 2;
    }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(wm_multi_create) LOAD(wm_multi_destroy) LOAD(wm_multi_device_ctx) LOAD(wm_multi_transcribe_greedy) LOAD(wm_multi_size)
    LOAD(wm_init_synthetic) LOAD(wm_finalize) LOAD(wm_transcribe_greedy) LOAD(wm_last_error)
    wm_dims d;
    const std::string model = argv[2];
    if (model == "tiny.en") d = {80, 1500, 384, 6, 4, 51864, 448, 384, 6, 4};
    else if (model == "base") d = {80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6};
    else if (model == "small") d = {80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12};
    else if (model == "large-v2") d = {80, 1500, 1280, 20, 32, 51865, 448, 1280, 20, 32};
    else if (model == "large-v3") d = {128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32};
    else { fprintf(stderr, "unknown model %s\n", argv[2]); return 2; }
    // argv[4]: a chunk count (synthetic recording) or a 16 kHz mono 16-bit .wav of any length (cut into 30 s windows by the
    // library's own reader, wm_wav_*: the long-audio front door of SURVEY.md 8f rank 1 for a host without AVFoundation)
    const std::string rec = argv[4];
    const bool from_wav = rec.size() > 4 && rec.compare(rec.size() - 4, 4, ".wav") == 0;
    std::vector<int16_t> pcm;
    int n_chunks = from_wav ? 0 : atoi(argv[4]);
    if (from_wav) {
        LOAD(wm_wav_open) LOAD(wm_wav_close) LOAD(wm_wav_num_chunks) LOAD(wm_wav_read_chunks)
        wm_wav *wav = nullptr;
        if (wm_wav_open(rec.c_str(), &wav) != WM_OK) { fprintf(stderr, "wm_wav_open: %s\n", wm_last_error()); return 1; }
        n_chunks = wm_wav_num_chunks(wav);
        pcm.resize((size_t)n_chunks * 480000);
        if (wm_wav_read_chunks(wav, 0, n_chunks, pcm.data()) != WM_OK) {
            fprintf(stderr, "wm_wav_read_chunks: %s\n", wm_last_error());
            wm_wav_close(wav);
            return 1;
        }
        wm_wav_close(wav);
    }
    const int n_gpus = atoi(argv[3]), max_new = argc > 5 ? atoi(argv[5]) : 8;
    std::vector<int> devs(n_gpus);
    for (int i = 0; i < n_gpus; ++i) devs[i] = i;
    wm_multi *m = nullptr;
    if (wm_multi_create(&d, devs.data(), n_gpus, &m) != WM_OK) { fprintf(stderr, "wm_multi_create: %s\n", wm_last_error()); return 1; }
    for (int r = 0; r < wm_multi_size(m); ++r) {  // weights are replicated: the same seed / file on every device
        wm_ctx *c = nullptr;
        if (wm_multi_device_ctx(m, r, &c) != WM_OK || wm_init_synthetic(c, 11) != WM_OK || wm_finalize(c) != WM_OK) {
            fprintf(stderr, "weights on rank %d: %s\n", r, wm_last_error());
            return 1;
        }
    }
    // a "recording" of n_chunks windows: amplitude-modulated tones, int16 (AudioRecorder.swift:56-61 records LinearPCM)
    if (!from_wav) pcm.resize((size_t)n_chunks * 480000);
    for (int c = 0; c < (from_wav ? 0 : n_chunks); ++c)
        for (int i = 0; i < 480000; ++i) {
            const double t = i / 16000.0;
            pcm[(size_t)c * 480000 + i] = (int16_t)lrint(9000.0 * sin(2 * M_PI * (200 + 370 * (c % 7)) * t) * (0.5 + 0.5 * sin(2 * M_PI * 0.3 * t)));
        }
    const int32_t prompt[4] = {d.n_vocab >= 51865 ? 50258 : 50257, d.n_vocab >= 51865 ? 50259 : 50362, 10, 11};
    std::vector<int32_t> tok((size_t)n_chunks * max_new), len(n_chunks), tok1(tok.size()), len1(n_chunks);
    const auto t0 = std::chrono::steady_clock::now();
    if (wm_multi_transcribe_greedy(m, pcm.data(), WM_I16, n_chunks, prompt, 4, max_new, -1, tok.data(), len.data()) != WM_OK) {
        fprintf(stderr, "wm_multi_transcribe_greedy: %s\n", wm_last_error());
        return 1;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int c = 0; c < n_chunks; ++c) {
        printf("chunk %d: %d", c, len[c]);
        for (int i = 0; i < max_new; ++i) printf(" %d", tok[(size_t)c * max_new + i]);
        printf("\n");
    }
    wm_ctx *c0 = nullptr;
    wm_multi_device_ctx(m, 0, &c0);
    if (wm_transcribe_greedy(c0, pcm.data(), WM_I16, n_chunks, prompt, 4, max_new, -1, tok1.data(), len1.data(), WM_MEM_HOST) != WM_OK) {
        fprintf(stderr, "single-device run: %s\n", wm_last_error());
        return 1;
    }
    printf("identical_to_single_gpu %d\n", (int)(tok == tok1 && len == len1));
    printf("%.3f\n", sec);
    wm_multi_destroy(m);
    return 0;
}
