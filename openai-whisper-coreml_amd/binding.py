"""ctypes mirror of the reference's Swift surface on top of the C ABI.

The reference's host is Swift (no Swift toolchain exists in this image or on the GPU
box), so this module plays the Swift caller's role for tests and benchmarks, with the
same names, argument meaning and error behaviour:

    generateSpectrogram(audio)        Whisper/Whisper/stft.swift:8-19
    Whisper(...)                      Whisper/Whisper/Whisper.swift:11-21   (init)
    Whisper.encode(audio)             Whisper/Whisper/Whisper.swift:23-31
    Whisper.decode(audioFeatures)     Whisper/Whisper/Whisper.swift:33-40
    Whisper.LANGUAGES                 Whisper/Whisper/Whisper.swift:12

Everything goes through libwhisper_mi355x.so (include/whisper_mi355x.h).  There is no
CPU fallback: if the library is missing, or no gfx950 device is usable, calls raise.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WM_LIB_PATH") or os.path.join(HERE, "libwhisper_mi355x.so")   # WM_LIB_PATH: A/B builds
# the product objects + the wmdbg_* kernel test hooks (include/whisper_mi355x_debug.h): tests / tools only
DEBUG_LIB_PATH = os.environ.get("WM_DBG_LIB_PATH") or os.path.join(HERE, "libwhisper_mi355x_dbg.so")   # A/B builds (tools/)

WM_OK = 0
WM_I16, WM_F32, WM_F64, WM_BF16 = 0, 1, 2, 3
WM_MEM_HOST, WM_MEM_DEVICE = 0, 1

N_SAMPLES = 16000 * 30  # ContentView.swift:57
N_FRAMES = 3000

_DTYPES = {np.dtype(np.int16): WM_I16, np.dtype(np.float32): WM_F32, np.dtype(np.float64): WM_F64}


class WhisperError(RuntimeError):
    """Plays the role of Swift's `throws` (Whisper.swift:17,23,33)."""

    def __init__(self, status, message):
        super().__init__("wm status %d: %s" % (status, message))
        self.status = status


class wm_dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# ModelDimensions of the checkpoints named in BASELINE.json / SURVEY.md 8a.
MODEL_DIMS = {
    "tiny.en": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=384, n_audio_head=6, n_audio_layer=4,
                    n_vocab=51864, n_text_ctx=448, n_text_state=384, n_text_head=6, n_text_layer=4),
    "base": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=512, n_audio_head=8, n_audio_layer=6,
                 n_vocab=51865, n_text_ctx=448, n_text_state=512, n_text_head=8, n_text_layer=6),
    "small": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=768, n_audio_head=12, n_audio_layer=12,
                  n_vocab=51865, n_text_ctx=448, n_text_state=768, n_text_head=12, n_text_layer=12),
    "large-v2": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=1280, n_audio_head=20,
                     n_audio_layer=32, n_vocab=51865, n_text_ctx=448, n_text_state=1280,
                     n_text_head=20, n_text_layer=32),
    "large-v3": dict(n_mels=128, n_audio_ctx=1500, n_audio_state=1280, n_audio_head=20,
                     n_audio_layer=32, n_vocab=51866, n_text_ctx=448, n_text_state=1280,
                     n_text_head=20, n_text_layer=32),
}

_lib = None
_dbg_lib = None


def load_debug_library():
    """dlopen libwhisper_mi355x_dbg.so: the same objects as the product plus the wmdbg_* hooks."""
    global _dbg_lib
    if _dbg_lib is None:
        _dbg_lib = load_library(DEBUG_LIB_PATH)
    return _dbg_lib


def load_library(path=None):
    """dlopen the product library.  Fails loudly (no fallback) when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise WhisperError(-1, "%s not found: run `python __graft_entry__.py build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    lib = ctypes.CDLL(p)  # RTLD_LOCAL: never interpose another library's symbols
    vp, ip, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    pp = ctypes.POINTER(ctypes.c_void_p)
    lib.generate_spectrogram.argtypes = [vp, vp]
    lib.generate_spectrogram.restype = None
    lib.wm_last_error.restype = ctypes.c_char_p
    sigs = {
        "wm_logmel": [vp, vp, ip, ip, ip, vp, ip, ip],
        "wm_create_frontend": [ip, pp],
        "wm_create": [ctypes.POINTER(wm_dims), ip, pp],
        "wm_clone": [vp, pp],
        "wm_set_tensor": [vp, ctypes.c_char_p, vp, sz],
        "wm_get_tensor": [vp, ctypes.c_char_p, vp, sz],
        "wm_load_weights": [vp, ctypes.c_char_p],
        "wm_init_synthetic": [vp, ctypes.c_uint64],
        "wm_init_synthetic_gain": [vp, ctypes.c_uint64, ctypes.c_float],
        "wm_finalize": [vp],
        "wm_get_dims": [vp, ctypes.POINTER(wm_dims)],
        "wm_encode": [vp, vp, ip, vp, ip],
        "wm_decode_logits": [vp, vp, ip, ip, vp, vp, ip],
        "wm_detect_language": [vp, vp, ip, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, ip],
        "wm_transcribe_greedy": [vp, vp, ip, ip, vp, ip, ip, ctypes.c_int32, vp, vp, ip],
        "wm_set_token_budgets": [vp, vp, ip],
        "wm_set_lanes": [vp, ip],
        "wm_dev_malloc": [vp, sz, pp],
        "wm_dev_free": [vp, vp],
        "wm_dev_upload": [vp, vp, vp, sz],
        "wm_dev_download": [vp, vp, vp, sz],
        "wm_sync": [vp],
        "wm_profile_enable": [vp, ip],
        "wm_profile_reset": [vp],
        "wm_profile_json": [vp, ctypes.c_char_p, sz],
        "wm_profile_overhead_us": [vp, vp],
        "wm_last_stage_ms": [vp, vp],
        "wm_vocab_load": [ctypes.c_char_p, pp],
        "wm_vocab_size": [vp],
        "wm_detokenize": [vp, vp, ip, ip, vp, sz, vp],
        "wm_wav_open": [ctypes.c_char_p, pp],
        "wm_wav_num_chunks": [vp],
        "wm_wav_read_chunks": [vp, ip, ip, vp],
        "wm_multi_create": [ctypes.POINTER(wm_dims), vp, ip, pp],
        "wm_multi_size": [vp],
        "wm_multi_device_ctx": [vp, ip, pp],
        "wm_multi_transcribe_greedy": [vp, vp, ip, ip, vp, ip, ip, ctypes.c_int32, vp, vp],
        "wm_multi_partition": [ip, ip, ip, vp, vp],
        "wm_multi_pack_tokens": [vp, vp, ip, ip, ip, vp],
        "wm_multi_unpack_tokens": [vp, ip, ip, ip, ip, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    lib.wm_destroy.argtypes = [vp]
    lib.wm_destroy.restype = None
    lib.wm_multi_destroy.argtypes = [vp]
    lib.wm_multi_destroy.restype = None
    lib.wm_vocab_free.argtypes = [vp]
    lib.wm_vocab_free.restype = None
    lib.wm_wav_close.argtypes = [vp]
    lib.wm_wav_close.restype = None
    lib.wm_wav_num_samples.argtypes = [vp]
    lib.wm_wav_num_samples.restype = ctypes.c_long
    if path is None:
        _lib = lib
    return lib


def _check(lib, status):
    if status != WM_OK:
        raise WhisperError(status, lib.wm_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def generateSpectrogram(audio):
    """stft.swift:8-19.  audio: 480000 doubles -> 240000 doubles ([80][3000] row-major).

    Inserts 200 zeros at the front (stft.swift:10), appends 200 (stft.swift:11), allocates
    the 80*3000 result (stft.swift:12) and calls the C symbol `generate_spectrogram` with
    raw pointers (stft.swift:13-17).  Like the Swift caller, no length check is made by
    the callee; this wrapper checks because Python has no UB to lean on."""
    lib = load_library()
    a = np.asarray(audio, dtype=np.float64)
    if a.shape != (N_SAMPLES,):
        raise ValueError("generateSpectrogram expects %d samples, got %r" % (N_SAMPLES, a.shape))
    buf = np.zeros(N_SAMPLES + 400, dtype=np.float64)
    buf[200:200 + N_SAMPLES] = a
    result = np.zeros(80 * N_FRAMES, dtype=np.float64)
    lib.generate_spectrogram(_ptr(buf), _ptr(result))
    return result


class Context:
    """Thin RAII wrapper over wm_ctx (front end only unless `dims` is given)."""

    def __init__(self, dims=None, device=0, debug=False):
        self.lib = load_debug_library() if debug else load_library()
        self.handle = ctypes.c_void_p()
        if dims is None:
            _check(self.lib, self.lib.wm_create_frontend(int(device), ctypes.byref(self.handle)))
            self.dims = None
        else:
            d = wm_dims(**dims) if isinstance(dims, dict) else dims
            _check(self.lib, self.lib.wm_create(ctypes.byref(d), int(device), ctypes.byref(self.handle)))
            self.dims = d.as_dict()

    def clone(self):
        """Second context sharing this one's finalised weights (own stream / caches): for overlapping
        independent batches on one GPU from several host threads."""
        c = Context.__new__(Context)
        c.lib = self.lib
        c.handle = ctypes.c_void_p()
        c.dims = dict(self.dims)
        c._parent = self   # keep the weight owner alive
        _check(self.lib, self.lib.wm_clone(self.handle, ctypes.byref(c.handle)))
        return c

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.wm_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- front end ------------------------------------------------------------------
    def logmel(self, pcm, n_mels=80, out_dtype=np.float32):
        """wm_logmel on host arrays.  pcm: [n][480000] int16 / float32 / float64."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        if pcm.ndim != 2 or pcm.shape[1] != N_SAMPLES:
            raise ValueError("pcm must be [n_chunks][%d], got %r" % (N_SAMPLES, pcm.shape))
        if pcm.dtype not in _DTYPES:
            raise ValueError("pcm dtype must be int16/float32/float64, got %s" % pcm.dtype)
        n = pcm.shape[0]
        out = np.empty((n, n_mels, N_FRAMES), dtype=out_dtype)
        _check(self.lib, self.lib.wm_logmel(self.handle, _ptr(pcm), _DTYPES[pcm.dtype], n, n_mels,
                                            _ptr(out), _DTYPES[np.dtype(out_dtype)], WM_MEM_HOST))
        return out

    # ---- device memory --------------------------------------------------------------
    def dev_malloc(self, nbytes):
        p = ctypes.c_void_p()
        _check(self.lib, self.lib.wm_dev_malloc(self.handle, nbytes, ctypes.byref(p)))
        return p

    def dev_free(self, p):
        _check(self.lib, self.lib.wm_dev_free(self.handle, p))

    def upload(self, p, arr):
        arr = np.ascontiguousarray(arr)
        _check(self.lib, self.lib.wm_dev_upload(self.handle, p, _ptr(arr), arr.nbytes))

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        _check(self.lib, self.lib.wm_dev_download(self.handle, _ptr(out), p, out.nbytes))
        return out

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.dev_malloc(arr.nbytes)
        self.upload(p, arr)
        return p

    def sync(self):
        _check(self.lib, self.lib.wm_sync(self.handle))

    # ---- profiling ------------------------------------------------------------------
    def profile_enable(self, on=True):
        _check(self.lib, self.lib.wm_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        _check(self.lib, self.lib.wm_profile_reset(self.handle))

    def profile(self):
        import json
        buf = ctypes.create_string_buffer(1 << 16)
        _check(self.lib, self.lib.wm_profile_json(self.handle, buf, len(buf)))
        return json.loads(buf.value.decode())

    def profile_overhead_us(self):
        v = ctypes.c_float()
        _check(self.lib, self.lib.wm_profile_overhead_us(self.handle, ctypes.byref(v)))
        return float(v.value)

    def last_stage_ms(self):
        out = np.zeros(3, dtype=np.float32)
        _check(self.lib, self.lib.wm_last_stage_ms(self.handle, _ptr(out)))
        return out

    # ---- weights --------------------------------------------------------------------
    def set_tensor(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        _check(self.lib, self.lib.wm_set_tensor(self.handle, name.encode(), _ptr(a), a.size))

    def get_tensor(self, name, shape):
        out = np.empty(shape, dtype=np.float32)
        _check(self.lib, self.lib.wm_get_tensor(self.handle, name.encode(), _ptr(out), out.size))
        return out

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self.set_tensor(k, np.asarray(v, dtype=np.float32))

    def load_weights(self, path):
        _check(self.lib, self.lib.wm_load_weights(self.handle, path.encode()))

    def init_synthetic(self, seed, matrix_gain=1.0):
        """wm_init_synthetic / wm_init_synthetic_gain (matrix_gain 4: the `lively` random-init model of the token tests)."""
        if matrix_gain == 1.0:
            _check(self.lib, self.lib.wm_init_synthetic(self.handle, int(seed)))
        else:
            _check(self.lib, self.lib.wm_init_synthetic_gain(self.handle, int(seed), float(matrix_gain)))

    def set_precision(self, f32):
        """Debug library only (Context(..., debug=True)): route wm_encode / wm_decode_logits of this context through the
        all-fp32 debug model path (wmdbg_set_precision, csrc/f32_path.hip) or back to the product kernels."""
        if not hasattr(self.lib, "wmdbg_set_precision"):
            raise WhisperError(-1, "set_precision needs the debug library: Context(dims, debug=True)")
        self.lib.wmdbg_set_precision.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.lib.wmdbg_set_precision.restype = ctypes.c_int
        _check(self.lib, self.lib.wmdbg_set_precision(self.handle, WM_F32 if f32 else WM_BF16))

    def finalize(self):
        _check(self.lib, self.lib.wm_finalize(self.handle))

    # ---- model (host arrays) --------------------------------------------------------
    def encode_mel(self, mel):
        """wm_encode: f32 [B][n_mels][3000] -> f32 [B][n_audio_ctx][d]."""
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        B = mel.shape[0]
        xa = np.empty((B, self.dims["n_audio_ctx"], self.dims["n_audio_state"]), dtype=np.float32)
        _check(self.lib, self.lib.wm_encode(self.handle, _ptr(mel), B, _ptr(xa), WM_MEM_HOST))
        return xa

    def decode_logits(self, tokens, xa):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        xa = np.ascontiguousarray(xa, dtype=np.float32)
        B, T = tokens.shape
        logits = np.empty((B, T, self.dims["n_vocab"]), dtype=np.float32)
        _check(self.lib, self.lib.wm_decode_logits(self.handle, _ptr(tokens), B, T, _ptr(xa),
                                                   _ptr(logits), WM_MEM_HOST))
        return logits

    def detect_language(self, xa, sot=50258, lang_first=50259, lang_last=50357):
        xa = np.ascontiguousarray(xa, dtype=np.float32)
        B = xa.shape[0]
        idx = np.empty(B, dtype=np.int32)
        _check(self.lib, self.lib.wm_detect_language(self.handle, _ptr(xa), B, sot, lang_first,
                                                     lang_last, _ptr(idx), WM_MEM_HOST))
        return idx

    def detect_language_probs(self, xa, sot=50258, lang_first=50259, lang_last=50357):
        """(lang_idx [B], probs [B][n_lang]): openai-whisper detect_language() -- softmax over the language tokens."""
        xa = np.ascontiguousarray(xa, dtype=np.float32)
        Bn = xa.shape[0]
        idx = np.empty(Bn, dtype=np.int32)
        probs = np.empty((Bn, lang_last - lang_first + 1), dtype=np.float32)
        self.lib.wm_detect_language_probs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_int]
        _check(self.lib, self.lib.wm_detect_language_probs(self.handle, _ptr(xa), Bn, sot, lang_first, lang_last, _ptr(idx),
                                                           _ptr(probs), WM_MEM_HOST))
        return idx, probs

    def set_suppress(self, suppress=(), suppress_first=()):
        """openai-whisper decode() logit filters for transcribe_greedy: `suppress` ids are never generated
        (SuppressTokens), `suppress_first` ids additionally not as the first generated token (SuppressBlank)."""
        a = np.ascontiguousarray(list(suppress), dtype=np.int32)
        b = np.ascontiguousarray(list(suppress_first), dtype=np.int32)
        self.lib.wm_set_suppress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        _check(self.lib, self.lib.wm_set_suppress(self.handle, _ptr(a) if a.size else None, int(a.size),
                                                  _ptr(b) if b.size else None, int(b.size)))

    def set_timestamp_rules(self, enable, timestamp_begin=0, eot=0, max_initial=-1):
        """openai-whisper ApplyTimestampRules for transcribe_greedy (decoding with timestamps)."""
        self.lib.wm_set_timestamp_rules.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int32,
                                                    ctypes.c_int32]
        _check(self.lib, self.lib.wm_set_timestamp_rules(self.handle, 1 if enable else 0, timestamp_begin, eot, max_initial))

    def set_lanes(self, n):
        """Decode groups one transcribe_greedy call keeps in flight (0: default = $WM_LANES or 3; 1: one group per call)."""
        _check(self.lib, self.lib.wm_set_lanes(self.handle, int(n)))

    def set_token_budgets(self, budgets):
        """Per-chunk token budgets of the NEXT transcribe_greedy call (len == its B; consumed by it)."""
        a = np.ascontiguousarray(list(budgets), dtype=np.int32)
        _check(self.lib, self.lib.wm_set_token_budgets(self.handle, _ptr(a) if a.size else None, int(a.size)))

    def transcribe_greedy(self, pcm, prompt, max_new, eot=-1, mem=WM_MEM_HOST, pcm_dtype=None, B=None, budgets=None):
        """pcm: host array [B][480000] (int16/float32/float64), or a device pointer
        (c_void_p) with pcm_dtype and B given when mem == WM_MEM_DEVICE.  budgets: per-chunk token budgets
        (wm_set_token_budgets) for this call."""
        if budgets is not None:
            self.set_token_budgets(budgets)
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        if mem == WM_MEM_HOST:
            pcm = np.ascontiguousarray(pcm)
            B = pcm.shape[0]
            pcm_dtype = _DTYPES[pcm.dtype]
            p = _ptr(pcm)
        else:
            p = pcm
        toks = np.empty((B, max_new), dtype=np.int32)
        lens = np.empty(B, dtype=np.int32)
        _check(self.lib, self.lib.wm_transcribe_greedy(self.handle, p, pcm_dtype, B, _ptr(prompt),
                                                       len(prompt), max_new, eot, _ptr(toks),
                                                       _ptr(lens), mem))
        return toks, lens


class Vocab:
    """wm_vocab: GPT-2 byte-level BPE de-tokenizer over a host-supplied vocab.json (ids -> UTF-8 text, no GPU)."""

    def __init__(self, vocab_json_path):
        self.lib = load_library()
        self.handle = ctypes.c_void_p()
        _check(self.lib, self.lib.wm_vocab_load(str(vocab_json_path).encode(), ctypes.byref(self.handle)))

    def __len__(self):
        return int(self.lib.wm_vocab_size(self.handle))

    def decode(self, ids, skip_special=True):
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1)
        need = ctypes.c_size_t()
        _check(self.lib, self.lib.wm_detokenize(self.handle, _ptr(ids), len(ids), 1 if skip_special else 0, None, 0,
                                                ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        _check(self.lib, self.lib.wm_detokenize(self.handle, _ptr(ids), len(ids), 1 if skip_special else 0, buf,
                                                need.value, None))
        return buf.raw[:need.value - 1].decode("utf-8", "replace")

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.wm_vocab_free(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Wav:
    """wm_wav: 16 kHz mono 16-bit RIFF/WAVE reader + 30 s chunker behind the C ABI (host only)."""

    def __init__(self, path):
        self.lib = load_library()
        self.handle = ctypes.c_void_p()
        _check(self.lib, self.lib.wm_wav_open(str(path).encode(), ctypes.byref(self.handle)))

    @property
    def num_samples(self):
        return int(self.lib.wm_wav_num_samples(self.handle))

    @property
    def num_chunks(self):
        return int(self.lib.wm_wav_num_chunks(self.handle))

    def chunks(self, first=0, n=None):
        n = self.num_chunks - first if n is None else n
        out = np.empty((n, N_SAMPLES), dtype=np.int16)
        _check(self.lib, self.lib.wm_wav_read_chunks(self.handle, int(first), int(n), _ptr(out)))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.wm_wav_close(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiContext:
    """wm_multi: every listed GPU of the node from ONE process (include/whisper_mi355x.h): weights replicated, chunks
    block-partitioned, one RCCL all-gather of the token streams."""

    def __init__(self, dims, devices=(0,)):
        self.lib = load_library()
        self.handle = ctypes.c_void_p()
        d = wm_dims(**dims) if isinstance(dims, dict) else dims
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        _check(self.lib, self.lib.wm_multi_create(ctypes.byref(d), _ptr(devs), len(devs), ctypes.byref(self.handle)))
        self.dims = d.as_dict()
        self.n = int(self.lib.wm_multi_size(self.handle))

    def device_ctx(self, rank):
        """Borrowed Context of one device (owned by the MultiContext: do not close it)."""
        c = Context.__new__(Context)
        c.lib = self.lib
        c.handle = ctypes.c_void_p()
        c.dims = dict(self.dims)
        _check(self.lib, self.lib.wm_multi_device_ctx(self.handle, int(rank), ctypes.byref(c.handle)))
        c.close = lambda: None
        c._owner = self
        return c

    def init_synthetic(self, seed):
        for r in range(self.n):
            c = self.device_ctx(r)
            c.init_synthetic(seed)
            c.finalize()

    def transcribe_greedy(self, pcm, prompt, max_new, eot=-1):
        pcm = np.ascontiguousarray(pcm)
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        Bn = pcm.shape[0]
        toks = np.empty((Bn, max_new), dtype=np.int32)
        lens = np.empty(Bn, dtype=np.int32)
        _check(self.lib, self.lib.wm_multi_transcribe_greedy(self.handle, _ptr(pcm), _DTYPES[pcm.dtype], Bn, _ptr(prompt),
                                                             len(prompt), max_new, eot, _ptr(toks), _ptr(lens)))
        return toks, lens

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.wm_multi_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Whisper:
    """Mirror of `struct Whisper` (Whisper/Whisper/Whisper.swift:11-41).

    The reference constructs CoreML `decoder` then `encoder` from bundled .mlpackages
    (Whisper.swift:17-21); here construction takes the model dimensions and a weight
    source (state dict, flat weight file, or a synthetic seed) and uploads to HBM."""

    LANGUAGES = ["en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar",
                 "sv", "it", "id", "hi", "fi", "vi", "iw", "uk", "el", "ms", "cs", "ro", "da", "hu",
                 "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk", "te", "fa",
                 "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn",
                 "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc",
                 "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn",
                 "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw",
                 "su"]  # Whisper.swift:12 (99 codes, openai-whisper tokenizer order)

    SOT = 50258          # Whisper.swift:35
    LANG_FIRST = 50259   # Whisper.swift:37
    LANG_LAST = 50357

    def __init__(self, dims="small", state_dict=None, weights_path=None, synthetic_seed=None, device=0):
        d = MODEL_DIMS[dims] if isinstance(dims, str) else dims
        self.ctx = Context(d, device=device)
        if state_dict is not None:
            self.ctx.load_state_dict(state_dict)
        elif weights_path is not None:
            self.ctx.load_weights(weights_path)
        elif synthetic_seed is not None:
            self.ctx.init_synthetic(synthetic_seed)
        else:
            raise ValueError("one of state_dict / weights_path / synthetic_seed is required")
        self.ctx.finalize()

    def encode(self, audio):
        """Whisper.swift:23-31: spectrogram -> f32 [1,80,3000] -> encoder -> [1,1500,d]."""
        spec = generateSpectrogram(audio)                                # :24
        array = spec.astype(np.float32).reshape(1, 80, N_FRAMES)         # :25-28 (f64 -> f32)
        return self.ctx.encode_mel(array)                                # :29

    def decode(self, audioFeatures):
        """Whisper.swift:33-40: one decoder step on SOT, arg-max over the language ids.
        The reference prints the code and returns Void; this returns the code."""
        idx = self.ctx.detect_language(audioFeatures, self.SOT, self.LANG_FIRST, self.LANG_LAST)
        return self.LANGUAGES[int(idx[0])]
