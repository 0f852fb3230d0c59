"""The step immediately before the hot path: PCM in, 30 s chunks out.

The reference records 16 kHz mono LinearPCM to `query.wav` and reads it back as Float32
(Whisper/Whisper/AudioRecorder.swift:56-61,74-86), then pads / truncates to ONE 30 s window
(Whisper/Whisper/ContentView.swift:57-60).  Here a recording of any length becomes a list of
independent 30 s windows (last one zero-padded), which is the unit the GPUs shard over."""
import wave

import numpy as np

from .sharding import N_SAMPLES, chunk_pcm

SAMPLE_RATE = 16000


def read_wav_int16(path):
    """16 kHz mono 16-bit RIFF/WAVE -> int16 samples (the format AudioRecorder.swift:56-61 writes)."""
    with wave.open(path, "rb") as w:
        if w.getframerate() != SAMPLE_RATE or w.getnchannels() != 1 or w.getsampwidth() != 2:
            raise ValueError("expected 16 kHz mono 16-bit PCM, got %d Hz, %d ch, %d-bit" % (
                w.getframerate(), w.getnchannels(), 8 * w.getsampwidth()))
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()


def write_wav_int16(path, samples):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(SAMPLE_RATE)
        w.writeframes(np.asarray(samples, dtype="<i2").tobytes())


def wav_to_chunks(path):
    """-> int16 [n_chunks][480000]; sample s maps to s / 32768 inside the front end."""
    return chunk_pcm(read_wav_int16(path))


def transcribe_chunks(ctx, chunks, prompt, max_new, eot=-1, batch=8):
    """Greedy-decode every 30 s chunk (B chunks per launch); returns (tokens [n][max_new], lens [n])."""
    toks, lens = [], []
    for i in range(0, len(chunks), batch):
        t, l = ctx.transcribe_greedy(chunks[i:i + batch], prompt, max_new, eot=eot)
        toks.append(t)
        lens.append(l)
    return np.concatenate(toks), np.concatenate(lens)


__all__ = ["read_wav_int16", "write_wav_int16", "wav_to_chunks", "transcribe_chunks", "N_SAMPLES", "SAMPLE_RATE"]
