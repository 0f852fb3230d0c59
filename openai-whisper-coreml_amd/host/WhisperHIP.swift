// WhisperHIP.swift -- drop-in replacement for Whisper/Whisper/Whisper.swift of tanmayb123/OpenAI-Whisper-CoreML that
// keeps the reference's surface (`struct Whisper`: init / encode(audio:) / decode(audioFeatures:) / LANGUAGES) and
// swaps the two CoreML models for libwhisper_mi355x.so, loaded with dlopen as BASELINE.json's north star prescribes.
//
// *** NOT COMPILED IN THIS REPOSITORY ***  Neither the build image nor the GPU box has a Swift toolchain (`swift`,
// `swiftc` absent; SURVEY.md section 0), so this file is written against include/whisper_mi355x.h by hand and has never
// been through a compiler.  What IS compiled and tested in its place, call for call:
//   host/lid_main.cpp    the language-ID flow of ContentView.swift:56-63 -> Whisper.swift:23-40 (C++, dlopen)
//   host/multi_main.cpp  all GPUs of the node from one process (wm_multi_*)
//   binding.py           ctypes mirror with these very names, used by every parity test
// On a machine with Swift for Linux: `swiftc -O WhisperHIP.swift stft.swift ... -o whisper` next to the .so.
//
// Unchanged from the reference and still needed: Whisper/Whisper/stft.swift (generateSpectrogram: +200 zeros each side,
// raw-pointer call of `generate_spectrogram`) and bridge.h:11 -- seam #1 keeps its symbol, so link libwhisper_mi355x.so
// where libstft.a was linked (project.pbxproj:18,44) and nothing in stft.swift changes.
import Foundation
#if canImport(Glibc)
import Glibc
#endif

struct WhisperError: Error, CustomStringConvertible {
    let description: String
}

/// Field-for-field `wm_dims` of include/whisper_mi355x.h (openai-whisper's ModelDimensions).
struct wm_dims {
    var n_mels: Int32, n_audio_ctx: Int32, n_audio_state: Int32, n_audio_head: Int32, n_audio_layer: Int32
    var n_vocab: Int32, n_text_ctx: Int32, n_text_state: Int32, n_text_head: Int32, n_text_layer: Int32

    /// `whisper.load_model("small")`, the reference's only model (whisper_to_cml.py:7).
    static let small = wm_dims(n_mels: 80, n_audio_ctx: 1500, n_audio_state: 768, n_audio_head: 12, n_audio_layer: 12,
                               n_vocab: 51865, n_text_ctx: 448, n_text_state: 768, n_text_head: 12, n_text_layer: 12)
    static let largeV2 = wm_dims(n_mels: 80, n_audio_ctx: 1500, n_audio_state: 1280, n_audio_head: 20, n_audio_layer: 32,
                                 n_vocab: 51865, n_text_ctx: 448, n_text_state: 1280, n_text_head: 20, n_text_layer: 32)
}

struct Whisper {
    // Whisper.swift:12, unchanged (99 codes, openai-whisper tokenizer order)
    static let LANGUAGES = ["en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it",
                            "id", "hi", "fi", "vi", "iw", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur",
                            "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn",
                            "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si",
                            "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo",
                            "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha",
                            "ba", "jw", "su"]

    private let lib: UnsafeMutableRawPointer
    private let ctx: OpaquePointer
    private let dims: wm_dims

    // C function types of include/whisper_mi355x.h
    private typealias CreateFn = @convention(c) (UnsafePointer<wm_dims>, Int32, UnsafeMutablePointer<OpaquePointer?>) -> Int32
    private typealias LoadFn = @convention(c) (OpaquePointer, UnsafePointer<CChar>) -> Int32
    private typealias CtxFn = @convention(c) (OpaquePointer) -> Int32
    private typealias DestroyFn = @convention(c) (OpaquePointer) -> Void
    private typealias EncodeFn = @convention(c) (OpaquePointer, UnsafePointer<Float>, Int32, UnsafeMutablePointer<Float>, Int32) -> Int32
    private typealias LangFn = @convention(c) (OpaquePointer, UnsafePointer<Float>, Int32, Int32, Int32, Int32,
                                               UnsafeMutablePointer<Int32>, Int32) -> Int32
    private typealias GreedyFn = @convention(c) (OpaquePointer, UnsafeRawPointer, Int32, Int32, UnsafePointer<Int32>, Int32, Int32,
                                                 Int32, UnsafeMutablePointer<Int32>, UnsafeMutablePointer<Int32>, Int32) -> Int32
    private typealias ErrFn = @convention(c) () -> UnsafePointer<CChar>
    private typealias VocabLoadFn = @convention(c) (UnsafePointer<CChar>, UnsafeMutablePointer<OpaquePointer?>) -> Int32
    private typealias DetokFn = @convention(c) (OpaquePointer, UnsafePointer<Int32>, Int32, Int32, UnsafeMutablePointer<CChar>?,
                                                Int, UnsafeMutablePointer<Int>?) -> Int32

    private func sym<T>(_ name: String) throws -> T {
        guard let p = dlsym(lib, name) else { throw WhisperError(description: "missing symbol \(name)") }
        return unsafeBitCast(p, to: T.self)
    }
    private func message() -> String {
        guard let f: ErrFn = try? sym("wm_last_error") else { return "unknown error" }
        return String(cString: f())
    }
    private func check(_ status: Int32) throws {
        if status != 0 { throw WhisperError(description: "wm status \(status): \(message())") }
    }

    /// was Whisper.swift:17-21 (`decoder(configuration:)`, `encoder(configuration:)`): create the context, load the flat
    /// weight file made by weights.convert_openai_pt (the counterpart of whisper_to_cml.py:6-8,45-52), freeze.
    init(library: String = "libwhisper_mi355x.so", weights: String = "small.wm", dims: wm_dims = .small,
         device: Int32 = 0) throws {
        guard let h = dlopen(library, RTLD_NOW | RTLD_LOCAL) else {
            throw WhisperError(description: "dlopen \(library): \(String(cString: dlerror()))")
        }
        lib = h
        self.dims = dims
        var d = dims
        var c: OpaquePointer?
        let create: CreateFn = unsafeBitCast(dlsym(h, "wm_create")!, to: CreateFn.self)
        let load: LoadFn = unsafeBitCast(dlsym(h, "wm_load_weights")!, to: LoadFn.self)
        let fin: CtxFn = unsafeBitCast(dlsym(h, "wm_finalize")!, to: CtxFn.self)
        let err: ErrFn = unsafeBitCast(dlsym(h, "wm_last_error")!, to: ErrFn.self)
        guard create(&d, device, &c) == 0, let cc = c, load(cc, weights) == 0, fin(cc) == 0 else {
            throw WhisperError(description: String(cString: err()))
        }
        ctx = cc
    }

    /// was Whisper.swift:23-31: spectrogram -> f32 [1, 80, 3000] -> encoder -> [1, 1500, d] (`.var_1385`).
    func encode(audio: [Double]) throws -> [Float] {
        let spec = generateSpectrogram(audio: audio)            // stft.swift:8-19, unchanged (seam #1)
        let mel = spec.map { Float($0) }                        // Whisper.swift:25-28: f64 -> f32 narrowing
        var xa = [Float](repeating: 0, count: 1500 * Int(dims.n_audio_state))
        let f: EncodeFn = try sym("wm_encode")
        try check(f(ctx, mel, 1, &xa, 0 /* WM_MEM_HOST */))
        return xa
    }

    /// was Whisper.swift:33-40: SOT 50258 -> decoder -> first arg-max over ids 50259...50357 -> print the code.
    func decode(audioFeatures: [Float]) throws {
        var idx: Int32 = 0
        let f: LangFn = try sym("wm_detect_language")
        try check(f(ctx, audioFeatures, 1, 50258, 50259, 50357, &idx, 0))
        print(Self.LANGUAGES[Int(idx)])                         // Whisper.swift:39
    }

    /// New surface (BASELINE.json): log-mel -> encoder -> KV-cached greedy decode of `audio` cut into 30 s windows
    /// (the last one zero-padded: ContentView.swift:57-60's rule per window).  Returns the token ids per window.
    func transcribe(audio: [Float], prompt: [Int32] = [50258, 50259, 50359, 50363], maxNew: Int32 = 224,
                    eot: Int32 = 50257) throws -> [[Int32]] {
        let n = 480_000
        let chunks = max(1, (audio.count + n - 1) / n)
        var pcm = [Float](repeating: 0, count: chunks * n)
        pcm.replaceSubrange(0..<audio.count, with: audio)
        var tokens = [Int32](repeating: 0, count: chunks * Int(maxNew))
        var lens = [Int32](repeating: 0, count: chunks)
        let f: GreedyFn = try sym("wm_transcribe_greedy")
        try pcm.withUnsafeBytes { p in
            try check(f(ctx, p.baseAddress!, 1 /* WM_F32 */, Int32(chunks), prompt, Int32(prompt.count), maxNew, eot,
                        &tokens, &lens, 0))
        }
        return (0..<chunks).map { c in Array(tokens[c * Int(maxNew)..<c * Int(maxNew) + Int(lens[c])]) }
    }

    /// The same from a recording on disk -- query.wav as AudioRecorder.swift:56-61 writes it (16 kHz mono 16-bit) -- through the
    /// library's own reader / chunker (wm_wav_*: no AVFoundation on a Linux host), with an optional per-window token budget
    /// (wm_set_token_budgets: a window that has used it up leaves the decode like one that has emitted `eot`).
    func transcribe(wav path: String, prompt: [Int32] = [50258, 50259, 50359, 50363], maxNew: Int32 = 224,
                    eot: Int32 = 50257, budgets: [Int32]? = nil) throws -> [[Int32]] {
        typealias WavOpenFn = @convention(c) (UnsafePointer<CChar>, UnsafeMutablePointer<OpaquePointer?>) -> Int32
        typealias WavCountFn = @convention(c) (OpaquePointer) -> Int32
        typealias WavReadFn = @convention(c) (OpaquePointer, Int32, Int32, UnsafeMutablePointer<Int16>) -> Int32
        typealias BudgetFn = @convention(c) (OpaquePointer, UnsafePointer<Int32>?, Int32) -> Int32
        let open: WavOpenFn = try sym("wm_wav_open")
        let count: WavCountFn = try sym("wm_wav_num_chunks")
        let read: WavReadFn = try sym("wm_wav_read_chunks")
        let close: DestroyFn = try sym("wm_wav_close")
        var w: OpaquePointer?
        try check(open(path, &w))
        defer { close(w!) }
        let chunks = Int(count(w!))
        var pcm = [Int16](repeating: 0, count: chunks * 480_000)
        try check(read(w!, 0, Int32(chunks), &pcm))
        if let b = budgets {
            let setBudgets: BudgetFn = try sym("wm_set_token_budgets")
            try check(setBudgets(ctx, b, Int32(b.count)))          // must be one per window
        }
        var tokens = [Int32](repeating: 0, count: chunks * Int(maxNew))
        var lens = [Int32](repeating: 0, count: chunks)
        let f: GreedyFn = try sym("wm_transcribe_greedy")
        try pcm.withUnsafeBytes { p in
            try check(f(ctx, p.baseAddress!, 0 /* WM_I16 */, Int32(chunks), prompt, Int32(prompt.count), maxNew, eot,
                        &tokens, &lens, 0))
        }
        return (0..<chunks).map { c in Array(tokens[c * Int(maxNew)..<c * Int(maxNew) + Int(lens[c])]) }
    }

    /// ids -> text with the tokenizer's vocab.json (wm_vocab_load / wm_detokenize; no vocabulary ships with the library).
    func text(of ids: [Int32], vocabJSON: String) throws -> String {
        let load: VocabLoadFn = try sym("wm_vocab_load")
        let detok: DetokFn = try sym("wm_detokenize")
        let free: DestroyFn = try sym("wm_vocab_free")
        var v: OpaquePointer?
        try check(load(vocabJSON, &v))
        defer { free(v!) }
        var need = 0
        try check(detok(v!, ids, Int32(ids.count), 1, nil, 0, &need))
        var buf = [CChar](repeating: 0, count: need)
        try check(detok(v!, ids, Int32(ids.count), 1, &buf, need, nil))
        return String(cString: buf)
    }
}

// ContentView.swift:61-62 (`whisper.encode(audio:)`, `whisper.decode(audioFeatures:)`) compiles unchanged against this
// struct, except that the intermediate is `[Float]` instead of `MLMultiArray`.
