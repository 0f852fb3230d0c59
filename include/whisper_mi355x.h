/*
 * whisper_mi355x.h -- C ABI of libwhisper_mi355x.so, the MI355X (gfx950) drop-in for the
 * two native halves of tanmayb123/OpenAI-Whisper-CoreML's hot path:
 *
 *   boundary #1  the Rust `stft` staticlib          (stft/src/lib.rs:110-122, bridge.h:11)
 *   boundary #2  the CoreML encoder/decoder classes (Whisper/Whisper/Whisper.swift:17-40,
 *                contract fixed by whisper_to_cml.py:10-43)
 *
 * Plain C: pointers, sizes, ints.  No torch / C++ types cross this boundary.  Every
 * function except generate_spectrogram returns an int status (WM_OK == 0) and never
 * aborts or throws across the FFI; wm_last_error() returns the message for the calling
 * thread.  A wm_ctx is not thread-safe; distinct contexts are independent.
 *
 * All pointers are HOST pointers unless the argument is documented "mem-space
 * selectable", in which case `mem` says where it lives (WM_MEM_HOST / WM_MEM_DEVICE).
 */
#ifndef WHISPER_MI355X_H
#define WHISPER_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#define WM_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ status codes --- */
enum {
    WM_OK = 0,
    WM_ERR_INVALID = 1,   /* bad argument (null, size, dtype, dims)                     */
    WM_ERR_HIP = 2,       /* a HIP runtime call failed / no gfx950 device               */
    WM_ERR_STATE = 3,     /* call order (weights not finalised, ctx has no model, ...)  */
    WM_ERR_IO = 4,        /* weight file unreadable / malformed                         */
    WM_ERR_NOMEM = 5
};

typedef enum { WM_I16 = 0, WM_F32 = 1, WM_F64 = 2, WM_BF16 = 3 } wm_dtype;
typedef enum { WM_MEM_HOST = 0, WM_MEM_DEVICE = 1 } wm_mem;

/* Model dimensions: the fields of openai-whisper's ModelDimensions, i.e. what
 * whisper.load_model(...) at whisper_to_cml.py:7 fixes for the exported graphs. */
typedef struct wm_dims {
    int32_t n_mels;        /* 80 (128 for large-v3)                                     */
    int32_t n_audio_ctx;   /* 1500                                                      */
    int32_t n_audio_state; /* d                                                         */
    int32_t n_audio_head;
    int32_t n_audio_layer;
    int32_t n_vocab;       /* 51865 (51864 *.en, 51866 large-v3)                        */
    int32_t n_text_ctx;    /* 448                                                       */
    int32_t n_text_state;
    int32_t n_text_head;
    int32_t n_text_layer;
} wm_dims;

typedef struct wm_ctx wm_ctx;

/* ------------------------------------------------------- boundary #1: the front end --- */

/* EXACT replacement for the reference's only native symbol:
 *   Whisper/Whisper/bridge.h:11      void generate_spectrogram(double *, double *);
 *   stft/src/lib.rs:110-122          #[no_mangle] pub extern fn generate_spectrogram(...)
 * arg0: 480400 f64, caller-owned and MUTATED exactly as lib.rs:34-40,113 does (elements
 *       [0,200) and [480200,480400) are overwritten with the reflected samples);
 * arg1: 240000 f64, row-major [80][3000] (lib.rs:116-121).
 * Lengths are implicit, nothing is retained, the symbol is re-entrant.  Runs the f64
 * kernels on device $WM_DEVICE (default 0).  Like the reference (unwrap -> panic ->
 * abort, lib.rs:45,85-87,106) it cannot report an error: on a HIP failure it prints the
 * reason to stderr and abort()s -- there is no CPU fallback. */
WM_API void generate_spectrogram(double *audio, double *output);

/* Batched, typed, error-returning form of the same computation (lib.rs:49-102).
 *   pcm   : [n_chunks][480000] samples, dtype WM_I16 (x = s/32768), WM_F32 or WM_F64;
 *           mem-space selectable.  (No +200 padding: the reflect of lib.rs:34-40 is done
 *           by index arithmetic on device.)
 *   n_mels: 80 (the reference's m80.npy filterbank, bit-exact) or 128 (slaney filters
 *           generated as openai-whisper's mel_128, for large-v3).
 *   out   : [n_chunks][n_mels][3000], dtype WM_F64 (f64 arithmetic end to end: the
 *           ABI-exact path) or WM_F32 (f32 arithmetic: the fast path); mem-space
 *           selectable (same `mem` as pcm). */
WM_API int wm_logmel(wm_ctx *ctx, const void *pcm, wm_dtype pcm_dtype, int n_chunks, int n_mels,
              void *out, wm_dtype out_dtype, wm_mem mem);

/* --------------------------------------------------------- context / weight loading --- */

/* Front-end-only context (no model): enough for wm_logmel. */
WM_API int wm_create_frontend(int device, wm_ctx **out);

/* Model context with uninitialised weights ( == Whisper.init, Whisper.swift:17-21, minus
 * the load).  Fill with wm_set_tensor / wm_load_weights / wm_init_synthetic, then
 * wm_finalize. */
WM_API int wm_create(const wm_dims *dims, int device, wm_ctx **out);

/* Set one parameter from host f32 data.  Names are openai-whisper state-dict keys, e.g.
 * "encoder.conv1.weight", "encoder.blocks.0.attn.query.weight",
 * "decoder.token_embedding.weight" (SURVEY.md 8f row 2).  n_elems must match. */
WM_API int wm_set_tensor(wm_ctx *ctx, const char *name, const float *data, size_t n_elems);
/* Read a parameter back as f32 (the value the kernels use, i.e. after bf16 rounding for
 * matrix weights). */
WM_API int wm_get_tensor(wm_ctx *ctx, const char *name, float *data, size_t n_elems);
/* Flat weight file written by openai-whisper-coreml_amd/weights.py (format: the docstring of weights.py). */
WM_API int wm_load_weights(wm_ctx *ctx, const char *path);
/* ---- TEST / BENCHMARK WEIGHTS: NOT FOR PRODUCTION USE.  The two generators below exist because no checkpoint can be
 * shipped or downloaded where the tests and bench.py run; a deployment loads real weights (wm_load_weights /
 * wm_set_tensor) and never calls them.  They stay in the product library (not the debug one) for one reason: bench.py's
 * headline must be measured on libwhisper_mi355x.so itself, with weights whose token streams can fail a cross-check. ----
 * Deterministic synthetic weights generated ON DEVICE (hash-based, approx N(0, std^2));
 * identical values to weights.synthetic_state_dict(dims, seed) on the host. */
WM_API int wm_init_synthetic(wm_ctx *ctx, uint64_t seed);
/* The same generator with every weight MATRIX (conv / linear / token embedding; not the positional tables, biases or
 * LayerNorm parameters) multiplied by matrix_gain -- weights.synthetic_state_dict(dims, seed, matrix_gain).  Gain 4 makes a
 * random-init model whose token streams depend on the audio and on the decode history (N(0, 0.02^2) weights give a nearly
 * input-independent one): what the token-level parity tests and bench.py's cross-checks decode.  A power of two keeps the
 * values bf16-exact. */
WM_API int wm_init_synthetic_gain(wm_ctx *ctx, uint64_t seed, float matrix_gain);
/* Freeze weights: fuse QKV, permute conv taps, precompute tables.  Required before any
 * model call. */
WM_API int wm_finalize(wm_ctx *ctx);
/* A second context on the same device that SHARES the (finalised, read-only) weights of `parent`
 * and owns its own HIP stream, activations, KV caches and decode graph.  Independent batches
 * submitted to different contexts from different host threads overlap on the GPU.  Destroy clones
 * before their parent. */
WM_API int wm_clone(wm_ctx *parent, wm_ctx **out);
WM_API void wm_destroy(wm_ctx *ctx);

WM_API const char *wm_last_error(void);
WM_API int wm_get_dims(const wm_ctx *ctx, wm_dims *out);

/* ---------------------------------------------------------- boundary #2: the model --- */

/* == encoderModel.prediction(x_1:).var_1385 (Whisper.swift:29; whisper_to_cml.py:10-23).
 *   mel: f32 [B][n_mels][3000]  ->  xa: f32 [B][n_audio_ctx][n_audio_state].
 * Both mem-space selectable (same `mem`). */
WM_API int wm_encode(wm_ctx *ctx, const float *mel, int B, float *xa, wm_mem mem);

/* == decoderModel.prediction(x_1:xa:).var_2217 (Whisper.swift:36; whisper_to_cml.py:25-43),
 * generalised from T == 1 to a T-token prefix (stateless: offset 0, no cache is kept).
 *   tokens: i32 [B][T];  xa: f32 [B][n_audio_ctx][d];  logits: f32 [B][T][n_vocab]. */
WM_API int wm_decode_logits(wm_ctx *ctx, const int32_t *tokens, int B, int T, const float *xa,
                     float *logits, wm_mem mem);

/* == Whisper.decode (Whisper.swift:33-40): one decoder step on <|startoftranscript|>
 * (id `sot`, 50258 in the reference), arg-max over logits[lang_first .. lang_last]
 * (50259...50357 in the reference), FIRST maximal element wins (Swift max(by:)).
 *   lang_idx: i32 [B], index into Whisper.LANGUAGES. */
WM_API int wm_detect_language(wm_ctx *ctx, const float *xa, int B, int32_t sot, int32_t lang_first,
                       int32_t lang_last, int32_t *lang_idx, wm_mem mem);

/* Same step, additionally returning the language probabilities of openai-whisper's detect_language() [3p]: softmax over
 * the language-token logits only.  probs: f32 [B][lang_last - lang_first + 1], same memory space as xa / lang_idx. */
WM_API int wm_detect_language_probs(wm_ctx *ctx, const float *xa, int B, int32_t sot, int32_t lang_first,
                             int32_t lang_last, int32_t *lang_idx, float *probs, wm_mem mem);

/* New surface asked for by BASELINE.json (not in the reference): front end + encoder +
 * KV-cached greedy decode of B independent 30 s chunks.  Any B >= 1: the call is cut into balanced
 * decode groups (8 .. 128 chunks) that run concurrently on up to $WM_LANES (default 3) weight-sharing
 * lanes inside the context; tokens do not depend on the grouping (bit-level batch invariance).
 *   pcm        : [B][480000], dtype WM_I16 / WM_F32 / WM_F64, mem-space selectable;
 *   prompt     : i32 [n_prompt] initial tokens (e.g. {sot, lang, transcribe, notimestamps});
 *   max_new    : tokens to generate per chunk (<= n_text_ctx - n_prompt);
 *   eot        : stop token; pass -1 to suppress stopping (fixed-length benchmark decode).  With eot >= 0 a chunk that
 *                has produced it leaves the decode (its K/V caches are not read again) and a decode group whose chunks
 *                have all stopped is not decoded any further: the work follows the longest live sequence, the results
 *                are what decoding all max_new positions and truncating would give;
 *   tokens_out : i32 [B][max_new] (host), padded with `eot` after a chunk stops;
 *   lens_out   : i32 [B] (host) generated length per chunk. */
WM_API int wm_transcribe_greedy(wm_ctx *ctx, const void *pcm, wm_dtype pcm_dtype, int B,
                         const int32_t *prompt, int n_prompt, int max_new, int32_t eot,
                         int32_t *tokens_out, int32_t *lens_out, wm_mem mem);

/* Decode groups a wm_transcribe_greedy call on this context keeps in flight.
 *   0 (default): the library's own measured policy -- one group below 32 chunks, two groups (two weight-sharing lanes)
 *                up to 143, three from 144 chunks, never more than $WM_LANES (default 3) at once; for the NARROW models
 *                (decoder width <= 512: tiny, base) the two groups of a 24 .. 128-chunk call (tiny: 32 .. 47) run on two
 *                SUB-CHIP lanes -- streams confined by a CU mask to complementary halves of every XCD's CUs (round 6:
 *                +2 .. +9 % there; a wide model's decode needs all the CUs and is never split this way);
 *   1          : the whole call (up to 128 chunks) is ONE decode group on the context's own stream -- what a host that runs
 *                its own concurrency over wm_clone'd contexts wants (bench.py);
 *   n = 2 .. 8 : n groups in flight whenever the call has 8 chunks for each (groups of ~8 up to 8 n chunks, n balanced
 *                groups of up to 128 beyond): a host that knows its latency / throughput trade-off better than the default.
 * Tokens do not depend on the choice (bit-level batch invariance).
 * EARLY STOP TRADE-OFF (eot >= 0 or wm_set_token_budgets): a decode group runs until its LAST row is finished, so the
 * default's single group below 32 chunks -- measured with fixed-length decodes, where it is the fastest cut -- decodes a
 * 24-chunk call with one straggler for ~0.8 of a full decode, where three groups of 8 (wm_set_lanes(3)) would have
 * spent ~0.6-0.7 (each group stops on its own).  A host whose utterance lengths vary widely and whose calls are 9 .. 31
 * chunks may prefer wm_set_lanes(2) or (3); the default is tuned for throughput at fixed length. */
WM_API int wm_set_lanes(wm_ctx *ctx, int n_lanes);

/* Logit filters of openai-whisper's greedy decode() (whisper/decoding.py SuppressTokens and SuppressBlank; SURVEY.md 8f
 * rank 3), applied inside the fused logits / arg-max kernel of wm_transcribe_greedy:
 *   suppress       : n token ids that are never generated (e.g. the tokenizer's non-speech tokens, sot, translate, ...);
 *   suppress_first : n_first more ids excluded only for the FIRST generated token (SuppressBlank: " " and <|endoftext|>).
 * The lists are copied; n = n_first = 0 clears the filter.  wm_decode_logits / wm_detect_language are unaffected (raw
 * logits).  Contexts made later with wm_clone inherit the filter; existing clones must be set themselves. */
WM_API int wm_set_suppress(wm_ctx *ctx, const int32_t *suppress, int n, const int32_t *suppress_first, int n_first);

/* openai-whisper's ApplyTimestampRules (whisper/decoding.py [3p]) for wm_transcribe_greedy, i.e. decoding WITH
 * timestamps (prompt without <|notimestamps|>): timestamps come in pairs, never decrease, the transcript opens with a
 * timestamp no later than max_initial_timestamp_index (e.g. 50 = 1.0 s; < 0: unlimited), and a timestamp is forced
 * whenever the summed probability of the admissible timestamps exceeds the best admissible text token.
 *   timestamp_begin : id of <|0.00|> (50364; 50365 for large-v3);  eot : <|endoftext|>.
 * Evaluated inside the fused logits / arg-max kernels; combine with wm_set_suppress (which should list <|notimestamps|>).
 * enable = 0 switches the rules off.  Same inheritance as wm_set_suppress. */
WM_API int wm_set_timestamp_rules(wm_ctx *ctx, int enable, int32_t timestamp_begin, int32_t eot,
                           int32_t max_initial_timestamp_index);

/* Per-chunk token budgets for the NEXT wm_transcribe_greedy call on this context (consumed by it; n must equal that
 * call's B): chunk i generates at most budgets[i] tokens (clamped to max_new), lens_out[i] <=
 * budgets[i].  A chunk that has reached its budget -- like one that has emitted `eot` -- LEAVES the decode: its caches are
 * not read again, and once every chunk of a decode group is finished no further position is launched for the group
 * (serving: per-request max_tokens; bench.py: a synthetic early-stop workload).  n = 0 clears. */
WM_API int wm_set_token_budgets(wm_ctx *ctx, const int32_t *budgets, int n);

/* ------------------------------------------------- all GPUs of the node, one host process --- */
/* the Swift host dlopens ONE library in ONE process; wm_multi drives n GPUs from it.  Weights are
 * replicated (one wm_ctx per device: fetch each with wm_multi_device_ctx and fill it with wm_load_weights /
 * wm_set_tensor / wm_init_synthetic + wm_finalize, exactly as a single context), a call's chunks are cut into contiguous
 * blocks -- rank r owns [r ceil(B/n), min(B, (r+1) ceil(B/n))) -- each block runs its whole path (front end -> encoder
 * -> greedy decode) on its own GPU from its own host thread, and the only exchange is ONE fixed-stride all-gather of
 * int32 [ceil(B/n)][1 + max_new] (length, tokens) per rank over RCCL / xGMI (ncclCommInitAll).  n = 1 is valid.
 *   devices    : n distinct HIP device ordinals;
 *   pcm        : HOST [B][480000] samples; tokens_out i32 [B][max_new], lens_out i32 [B] (host), as wm_transcribe_greedy. */
typedef struct wm_multi wm_multi;
WM_API int wm_multi_create(const wm_dims *dims, const int *devices, int n, wm_multi **out);
WM_API void wm_multi_destroy(wm_multi *m);
WM_API int wm_multi_size(const wm_multi *m);
WM_API int wm_multi_device_ctx(wm_multi *m, int rank, wm_ctx **out);
WM_API int wm_multi_transcribe_greedy(wm_multi *m, const void *pcm, wm_dtype pcm_dtype, int B, const int32_t *prompt,
                               int n_prompt, int max_new, int32_t eot, int32_t *tokens_out, int32_t *lens_out);
/* Host-only pieces of the above (usable without a GPU; also what the CPU tests pin): the block partition, and the
 * fixed-stride payload [per][1 + max_new] = (length, tokens) each rank contributes to the all-gather. */
WM_API int wm_multi_partition(int n_chunks, int world_size, int rank, int *lo, int *hi);
WM_API int wm_multi_pack_tokens(const int32_t *tokens, const int32_t *lens, int n_local, int per, int max_new,
                         int32_t *payload);
WM_API int wm_multi_unpack_tokens(const int32_t *gathered, int world_size, int per, int max_new, int n_chunks,
                           int32_t *tokens_out, int32_t *lens_out);

/* -------------------------------------------------------------- ids -> text (host only) --- */
/* GPT-2 byte-level BPE de-tokenizer for the ids wm_transcribe_greedy returns (the reference prints
 * a language code and has no tokenizer, Whisper.swift:37-39).  vocab_json_path: the tokenizer's vocab.json
 * ({"piece": id, ...}, pieces over GPT-2's byte alphabet) -- supplied by the host, no vocabulary ships with the library.
 * wm_detokenize: UTF-8 text of ids[0..n): ids without a piece (special tokens, timestamps) are skipped
 * (skip_special != 0) or written as <|id|>.  Writes at most cap - 1 bytes + NUL; *needed (nullable) = bytes required
 * including the NUL, so cap = 0 sizes the buffer.  No GPU involved. */
typedef struct wm_vocab wm_vocab;
WM_API int wm_vocab_load(const char *vocab_json_path, wm_vocab **out);
WM_API void wm_vocab_free(wm_vocab *v);
WM_API int wm_vocab_size(const wm_vocab *v);
WM_API int wm_detokenize(const wm_vocab *v, const int32_t *ids, int n, int skip_special, char *buf, size_t cap,
                  size_t *needed);

/* ------------------------------------------------ WAV reader + 30 s chunker (host only) --- */
/* The step BEFORE the path: the reference records 16 kHz mono 16-bit LinearPCM to query.wav
 * (AudioRecorder.swift:56-61), reads it back through AVFoundation (:74-86) and zero-pads / truncates to one 30 s window
 * (ContentView.swift:57-60).  For a dlopen-only host: the same file format in, and the reference's pad rule applied per
 * window, so a recording of any length becomes [n_chunks][480000] int16 (x = s / 32768 inside the front end) for
 * wm_logmel / wm_transcribe_greedy / wm_multi_transcribe_greedy.  Anything but 16 kHz mono 16-bit PCM is WM_ERR_IO. */
typedef struct wm_wav wm_wav;
WM_API int wm_wav_open(const char *path, wm_wav **out);
WM_API void wm_wav_close(wm_wav *w);
WM_API long wm_wav_num_samples(const wm_wav *w);
WM_API int wm_wav_num_chunks(const wm_wav *w);   /* ceil(samples / 480000), at least 1 */
/* windows [first_chunk, first_chunk + n_chunks) -> out int16 [n_chunks][480000], the last window zero-padded */
WM_API int wm_wav_read_chunks(const wm_wav *w, int first_chunk, int n_chunks, int16_t *out);

/* ------------------------------------------------------------ device memory helpers --- */
/* For callers that keep inputs resident in HBM (bench.py; a Swift host would use them to
 * avoid the 5.7 MB/chunk PCIe round trip of the reference ABI). */
WM_API int wm_dev_malloc(wm_ctx *ctx, size_t bytes, void **dptr);
WM_API int wm_dev_free(wm_ctx *ctx, void *dptr);
WM_API int wm_dev_upload(wm_ctx *ctx, void *dptr, const void *host, size_t bytes);
WM_API int wm_dev_download(wm_ctx *ctx, void *host, const void *dptr, size_t bytes);
WM_API int wm_sync(wm_ctx *ctx);

/* -------------------------------------------------------------------- measurement --- */
/* Per-kernel-family HIP-event timing on the context's stream.  When enabled, every launch
 * of a profiled kernel family is bracketed by hipEventRecord on the launch stream; the
 * totals are read back with wm_profile_get (which synchronises). */
WM_API int wm_profile_enable(wm_ctx *ctx, int on);
WM_API int wm_profile_reset(wm_ctx *ctx);
/* Bias (microseconds) of an event-bracketed launch on this stream, calibrated with a kernel that spins
 * for a known time of the device clock: subtract it from a family's mean launch duration. */
WM_API int wm_profile_overhead_us(wm_ctx *ctx, float *us);
/* Writes a JSON object {"family": {"ms": total_ms, "n": launches}, ...} into buf. */
WM_API int wm_profile_json(wm_ctx *ctx, char *buf, size_t buf_bytes);
/* Wall-clock stage split of the last wm_transcribe_greedy call, in ms (HIP events):
 * [0] front end, [1] encoder + cross-KV projection, [2] decode loop. */
WM_API int wm_last_stage_ms(wm_ctx *ctx, float out3[3]);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_MI355X_H */
