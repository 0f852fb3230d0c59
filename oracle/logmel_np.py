"""oracle/logmel_np.py -- CPU ORACLE (test infrastructure, NOT product code).

Independent numpy f64 restatement of /root/reference/stft/src/lib.rs:22-122 (numpy's
pocketfft replaces realfft/rustfft).  Used to (a) cross-check oracle/logmel_ref.c and
(b) generate the committed golden vectors in tests/golden/ (tests/golden/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

N_SAMPLES = 16000 * 30   # lib.rs:37,112
N_FFT = 400              # lib.rs:24
HOP = 160                # lib.rs:50
N_FRAMES = 3000          # lib.rs:52


def hann_periodic():
    """lib.rs:26."""
    i = np.arange(N_FFT, dtype=np.float64)
    return (1.0 - np.cos((i * 2.0 * np.pi) / 400.0)) / 2.0


def reflect_pad(x):
    """stft.swift:10-11 (+200 zeros each side) followed by lib.rs:34-40 (in-place reflect).
    x: (480000,) f64 -> (480400,) f64.  Written with the reference's own index rule."""
    a = np.zeros(N_SAMPLES + 400, dtype=np.float64)
    a[200:200 + N_SAMPLES] = x
    for i in range(200):
        a[i] = a[400 - i]
        j = N_SAMPLES + i + 200
        a[j] = a[200 + (N_SAMPLES - 2) - i]
    return a


def frame_index():
    """lib.rs:52: (0..len-400).step_by(160) -> 3000 frame offsets into the padded buffer."""
    starts = np.arange(0, N_SAMPLES + 400 - N_FFT, HOP)
    assert len(starts) == N_FRAMES
    return starts[:, None] + np.arange(N_FFT)[None, :]


def log_mel(x, filt):
    """x: (480000,) float (any dtype, widened to f64 as ContentView.swift:59 does);
    filt: (n_mels, 201) f32.  Returns (n_mels, 3000) f64 (lib.rs:116-121 layout)."""
    a = reflect_pad(np.asarray(x, dtype=np.float64))
    frames = a[frame_index()] * hann_periodic()[None, :]          # lib.rs:43
    spec = np.fft.rfft(frames, axis=1)                            # lib.rs:44-45
    power = spec.real ** 2 + spec.imag ** 2                       # lib.rs:54
    # lib.rs:60-69, dense, k ascending, f64 accumulation with the f32 filter widened
    f64 = filt.astype(np.float64)
    mel = np.zeros((filt.shape[0], N_FRAMES), dtype=np.float64)
    for k in range(201):
        mel += f64[:, k:k + 1] * power[None, :, k]
    mel = np.log10(np.where(mel > 1e-10, mel, 1e-10))             # lib.rs:71-79
    gmax = mel.max()                                              # lib.rs:82-88
    return (np.maximum(mel, gmax - 8.0) + 4.0) / 4.0              # lib.rs:91-99


def synth_chunk(chunk_idx):
    """BASELINE.md synthetic audio: rng(1234+idx), clip(0.1*N(0,1), -1, 1) as f32."""
    rng = np.random.default_rng(1234 + chunk_idx)
    return np.clip(0.1 * rng.standard_normal(N_SAMPLES), -1.0, 1.0).astype(np.float32)
