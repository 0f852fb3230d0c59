// lid_main.cpp -- C++ host harness standing in for the reference's Swift caller.
//
// The reference's host is a SwiftUI app (no Swift toolchain exists in this image or on the GPU
// box; INTEGRATION.md shows the Swift binding a maintainer would add).  This program performs the
// SAME sequence through the SAME C ABI the Swift side would dlopen:
//
//   ContentView.getAudioPredict  (Whisper/Whisper/ContentView.swift:56-63)
//     pad / truncate to 480000 samples, Float -> Double                  (:57-60)
//     Whisper.encode(audio:)      (Whisper/Whisper/Whisper.swift:23-31)
//        generateSpectrogram      (Whisper/Whisper/stft.swift:8-19)  -> generate_spectrogram()
//        f64 -> f32 [1,80,3000]   (Whisper.swift:25-28)
//        encoder.prediction       (Whisper.swift:29)                 -> wm_encode()
//     Whisper.decode(audioFeatures:) (Whisper.swift:33-40)
//        SOT 50258 -> decoder -> arg-max over 50259...50357          -> wm_detect_language()
//     print language, print elapsed seconds                           (Whisper.swift:39, ContentView.swift:63)
//
// usage: lid_main <libwhisper_mi355x.so> <model: tiny.en|base|small|large-v2> [weights.wm | synthetic:<seed>] [query.wav | pcm_f32.raw]
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "whisper_mi355x.h"

static const char *LANGUAGES[99] = {  // Whisper.swift:12
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi",
    "vi", "iw", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml",
    "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs",
    "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu", "am",
    "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln",
    "ha", "ba", "jw", "su"};

#define LOAD(name) auto name = (decltype(&::name))dlsym(lib, #name); if (!name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <libwhisper_mi355x.so> <tiny.en|base|small|large-v2> [weights.wm|synthetic:<seed>] [pcm_f32.raw]\n", argv[0]);
        return 2;
    }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);  // the north-star loading model: dlopen from the host
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(generate_spectrogram) LOAD(wm_create) LOAD(wm_load_weights) LOAD(wm_init_synthetic) LOAD(wm_finalize)
    LOAD(wm_encode) LOAD(wm_detect_language) LOAD(wm_destroy) LOAD(wm_last_error)

    wm_dims d;
    const std::string model = argv[2];
    if (model == "tiny.en") d = {80, 1500, 384, 6, 4, 51864, 448, 384, 6, 4};
    else if (model == "base") d = {80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6};
    else if (model == "small") d = {80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12};  // whisper_to_cml.py:7
    else if (model == "large-v2") d = {80, 1500, 1280, 20, 32, 51865, 448, 1280, 20, 32};
    else { fprintf(stderr, "unknown model %s\n", argv[2]); return 2; }

    // Whisper.init (Whisper.swift:17-21)
    wm_ctx *ctx = nullptr;
    if (wm_create(&d, 0, &ctx) != WM_OK) { fprintf(stderr, "wm_create: %s\n", wm_last_error()); return 1; }
    const std::string w = argc > 3 ? argv[3] : "synthetic:1";
    int st = w.rfind("synthetic:", 0) == 0 ? wm_init_synthetic(ctx, strtoull(w.c_str() + 10, nullptr, 10))
                                            : wm_load_weights(ctx, w.c_str());
    if (st == WM_OK) st = wm_finalize(ctx);
    if (st != WM_OK) { fprintf(stderr, "weights: %s\n", wm_last_error()); return 1; }

    // the recording: query.wav as AudioRecorder.swift:56-61 writes it (16 kHz mono 16-bit, read through the library's
    // wm_wav_* since there is no AVFoundation here; samples s -> s / 32768 as AVAudioFile's Float32 view gives them),
    // raw little-endian f32 mono 16 kHz, or 10 s of a 440 Hz tone (the app records 10 s, ContentView.swift:47)
    std::vector<float> audio;
    const std::string rec = argc > 4 ? argv[4] : "";
    if (rec.size() > 4 && rec.compare(rec.size() - 4, 4, ".wav") == 0) {
        LOAD(wm_wav_open) LOAD(wm_wav_close) LOAD(wm_wav_num_samples) LOAD(wm_wav_read_chunks)
        wm_wav *wav = nullptr;
        if (wm_wav_open(rec.c_str(), &wav) != WM_OK) { fprintf(stderr, "wm_wav_open: %s\n", wm_last_error()); return 1; }
        std::vector<int16_t> first(480000);
        if (wm_wav_read_chunks(wav, 0, 1, first.data()) != WM_OK) {
            fprintf(stderr, "wm_wav_read_chunks: %s\n", wm_last_error());
            wm_wav_close(wav);
            return 1;
        }
        const long ns = wm_wav_num_samples(wav);
        audio.resize(ns < 480000 ? (size_t)ns : (size_t)480000);
        for (size_t i = 0; i < audio.size(); ++i) audio[i] = (float)first[i] / 32768.0f;
        wm_wav_close(wav);
    } else if (argc > 4) {
        FILE *f = fopen(argv[4], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[4]); return 1; }
        float buf[4096];
        size_t n;
        while ((n = fread(buf, 4, 4096, f)) > 0) audio.insert(audio.end(), buf, buf + n);
        fclose(f);
    } else {
        audio.resize(160000);
        for (size_t i = 0; i < audio.size(); ++i) audio[i] = 0.2f * (float)__builtin_sin(2.0 * 3.14159265358979 * 440.0 * i / 16000.0);
    }

    const auto t0 = std::chrono::steady_clock::now();                       // ContentView.swift:56
    std::vector<double> input(16000 * 30 + 400, 0.0);                       // :57 (+ stft.swift:10-11 pads)
    const size_t n = audio.size() < 480000 ? audio.size() : 480000;         // :58 min(audio.count, input.count)
    for (size_t i = 0; i < n; ++i) input[200 + i] = (double)audio[i];       // :59
    std::vector<double> spec(80 * 3000);                                    // stft.swift:12
    generate_spectrogram(input.data(), spec.data());                        // stft.swift:13-17
    std::vector<float> mel(80 * 3000);
    for (size_t i = 0; i < mel.size(); ++i) mel[i] = (float)spec[i];        // Whisper.swift:25-28
    std::vector<float> xa((size_t)1500 * d.n_audio_state);
    if (wm_encode(ctx, mel.data(), 1, xa.data(), WM_MEM_HOST) != WM_OK) {   // Whisper.swift:29
        fprintf(stderr, "wm_encode: %s\n", wm_last_error());
        return 1;
    }
    int32_t lang = -1;
    const int32_t sot = d.n_vocab >= 51865 ? 50258 : 50257;                 // Whisper.swift:35
    if (d.n_vocab >= 51865) {
        if (wm_detect_language(ctx, xa.data(), 1, sot, 50259, 50357, &lang, WM_MEM_HOST) != WM_OK) {  // :36-38
            fprintf(stderr, "wm_detect_language: %s\n", wm_last_error());
            return 1;
        }
        printf("%s\n", LANGUAGES[lang]);                                    // Whisper.swift:39
    } else {
        printf("(English-only vocabulary: no language tokens)\n");
    }
    printf("%f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());  // ContentView.swift:63
    wm_destroy(ctx);
    dlclose(lib);
    return 0;
}
