#!/usr/bin/env python3
"""Regenerates the committed log-mel golden vectors.

The reference (Rust) cannot be executed here, so these vectors come from the independent
numpy restatement oracle/logmel_np.py of stft/src/lib.rs:22-122 (numpy pocketfft).  They
pin BOTH the C oracle and the HIP kernels.  A full output is 1.9 MB, so per case we keep
64 sampled frames (all 80 mels) plus whole-tensor checksums.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import logmel_np as L  # noqa: E402


def structured_chunk():
    """Sum of 3 bin-centred cosines (bins 25, 60, 140 -> 1000, 2400, 5600 Hz)."""
    n = np.arange(L.N_SAMPLES, dtype=np.float64)
    x = 0.3 * np.cos(2 * np.pi * 1000 * n / 16000) + 0.2 * np.cos(2 * np.pi * 2400 * n / 16000) \
        + 0.1 * np.cos(2 * np.pi * 5600 * n / 16000)
    return x.astype(np.float32)


def main():
    m80 = np.load(os.path.join(HERE, "m80.npy")).reshape(80, 201)
    rng = np.random.default_rng(7)
    frames = np.sort(np.concatenate([[0, 1, 2, 2997, 2998, 2999],
                                     rng.choice(np.arange(3, 2997), 58, replace=False)]))
    out = {"frames": frames}
    cases = {"noise0": L.synth_chunk(0), "noise1": L.synth_chunk(1), "cos3": structured_chunk()}
    quiet = L.synth_chunk(2).copy()
    quiet[160000:] = 0  # 10 s of audio then silence, like the app (ContentView.swift:47,57-60)
    cases["quiet_tail"] = quiet
    for name, x in cases.items():
        y = L.log_mel(x, m80)
        out[name + "_cols"] = y[:, frames]
        out[name + "_sum"] = np.array([y.sum(), np.abs(y).sum(), (y * y).sum(), y.max(), y.min()])
    np.savez_compressed(os.path.join(HERE, "logmel_golden.npz"), **out)
    print("wrote logmel_golden.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
