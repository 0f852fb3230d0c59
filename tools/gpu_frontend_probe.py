"""GPU probe: the f32 front end's stage 1 at B chunks (device-resident int16 in, f32 out), event-timed per launch by the
in-library profiler: the round-6 kernel (twiddle slices shared through LDS) against the round-1-5 kernel (debug knob
frontend_per_wave_twiddles: every wave fetches its own twiddles from L2), outputs compared bit for bit.

    python tools/gpu_frontend_probe.py [B=56] [n_mels=80]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openai_whisper_coreml_amd as pkg  # noqa: E402

B = pkg.binding


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 56
    n_mels = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    lib = B.load_debug_library()
    lib.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    fe = B.Context(debug=True)
    rng = np.random.default_rng(3)
    s16 = np.round(np.clip(0.1 * rng.standard_normal((nb, 480000)), -1, 1) * 32767).astype(np.int16)
    d_in = fe.to_device(s16)
    d_out = fe.dev_malloc(nb * n_mels * 3000 * 4)
    outs = []
    for knob, name in ((1, "per-wave twiddles from L2 (rounds 1-5)"), (0, "twiddle slices shared through LDS (round 6)")):
        assert lib.wmdbg_set_tuning(b"frontend_per_wave_twiddles", knob) == 0
        for _ in range(3):
            fe.lib.wm_logmel(fe.handle, d_in, 0, nb, n_mels, d_out, 1, 1)
        fe.sync()
        fe.profile_reset()
        fe.profile_enable(True)
        for _ in range(20):
            fe.lib.wm_logmel(fe.handle, d_in, 0, nb, n_mels, d_out, 1, 1)
        fe.sync()
        p = fe.profile()
        fe.profile_enable(False)
        us = p["logmel_stage1_f32"]["ms"] / p["logmel_stage1_f32"]["n"] * 1e3
        alg = nb * (480000 * 2 + n_mels * 3000 * 4)
        print("%d chunks, %d mels, %-44s stage1 %7.1f us (+ stage2 %.1f us): %6.1f GB/s of algorithmic bytes = %.3f of 8 TB/s" % (
            nb, n_mels, name, us, p["logmel_stage2_f32"]["ms"] / p["logmel_stage2_f32"]["n"] * 1e3, alg / us / 1e3, alg / us / 1e3 / 8000))
        outs.append(fe.download(d_out, (nb, n_mels, 3000), np.float32))
    print("bit-identical outputs:", bool(np.array_equal(outs[0], outs[1])))
    lib.wmdbg_set_tuning(b"reset", 0)
    fe.dev_free(d_in)
    fe.dev_free(d_out)
    fe.close()


if __name__ == "__main__":
    main()
