"""GPU probe: the reference's own flow (Whisper-small, ONE chunk: spectrogram -> encoder -> one decoder step -> language
arg-max; ContentView.swift:56-63, Whisper.swift:23-40) DEVICE-RESIDENT, split into its three calls: wall ms of each
(min of 10 after a warm-up, each call synchronised) and of the three back to back.

    python tools/gpu_small_flow_probe.py [model=small] [knob=value ...]   (knobs: wmdbg_set_tuning, debug library)"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def main():
    knobs = [a for a in sys.argv[1:] if "=" in a]
    pos = [a for a in sys.argv[1:] if "=" not in a]
    model = pos[0] if pos else "small"
    dims = B.MODEL_DIMS[model]
    c = B.Context(dims, debug=bool(knobs))
    if knobs:
        c.lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        for kv in knobs:
            k, v = kv.split("=")
            assert c.lib.wmdbg_set_tuning(k.encode(), int(v)) == 0, kv
        print("knobs:", " ".join(knobs))
    c.init_synthetic(20240928)
    c.finalize()
    rng = np.random.default_rng(2)
    x = np.clip(0.1 * rng.standard_normal(480000), -1, 1)
    d_x = c.to_device(x)
    d_mel = c.dev_malloc(dims["n_mels"] * 3000 * 4)
    d_xa = c.dev_malloc(1500 * dims["n_audio_state"] * 4)
    d_lang = c.dev_malloc(4)
    nl = 99 if dims["n_vocab"] == 51865 else 100
    steps = {
        "wm_logmel (f64 in, f32 out)": lambda: c.lib.wm_logmel(c.handle, d_x, B.WM_F64, 1, dims["n_mels"], d_mel, B.WM_F32, B.WM_MEM_DEVICE),
        "wm_encode": lambda: c.lib.wm_encode(c.handle, d_mel, 1, d_xa, B.WM_MEM_DEVICE),
        "wm_detect_language": lambda: c.lib.wm_detect_language(c.handle, d_xa, 1, 50258, 50259, 50258 + nl, d_lang, B.WM_MEM_DEVICE),
    }
    tot = 0.0
    for name, fn in steps.items():
        best = None
        for i in range(11):
            c.sync()
            t0 = time.perf_counter()
            assert fn() == 0
            c.sync()
            dt = time.perf_counter() - t0
            if i and (best is None or dt < best):
                best = dt
        tot += best
        print("%-32s %.3f ms" % (name, best * 1e3))
    best = None
    for i in range(11):
        c.sync()
        t0 = time.perf_counter()
        for fn in steps.values():
            assert fn() == 0
        idx = int(c.download(d_lang, (1,), np.int32)[0])
        dt = time.perf_counter() - t0
        if i and (best is None or dt < best):
            best = dt
    print("%-32s %.3f ms (sum of the parts %.3f); language index %d" % ("all three + 4-byte download", best * 1e3, tot * 1e3, idx))
    c.close()


if __name__ == "__main__":
    main()
