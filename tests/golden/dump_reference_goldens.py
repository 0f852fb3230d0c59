#!/usr/bin/env python3
"""Turn "parity unpinned" into "pinned" -- run this ONCE on a machine that has what this image lacks.

The reference (tanmayb123/OpenAI-Whisper-CoreML) holds no tests and no golden outputs, and neither of its two native
halves can be executed in the build container: the Rust `stft` crate needs cargo + its un-vendored crates (realfft 3.0.1,
rustfft 6.0.1, npy, num, lazy_static -- stft/Cargo.lock), the model half needs the `openai-whisper` package and the
"small" checkpoint that whisper_to_cml.py:7 downloads.  Every parity test therefore compares the HIP path with this
repo's own restatements (oracle/).  This script records the REFERENCE's outputs for the seeded inputs the tests already
use, as small fixtures under tests/golden/; tests/test_reference_goldens.py picks them up automatically (it is skipped
while they do not exist) and pins the oracle -- and through it the HIP path -- to the reference itself.

    # front end (needs cargo and network access for the crates; ~1 min: a release build of 5 small crates, ~60 MB of cargo
    # target directory in $TMPDIR, a 150 KB fixture; prints the oracle-vs-crate difference it just observed):
    python tests/golden/dump_reference_goldens.py --stft-crate /path/to/OpenAI-Whisper-CoreML/stft
    # model (needs `pip install openai-whisper`; downloads "small" exactly as whisper_to_cml.py:7 does: 461 MB once, then
    # ~20 s of CPU; a 1 MB fixture; prints the oracle-vs-openai-whisper difference on the checkpoint's own weights):
    python tests/golden/dump_reference_goldens.py --whisper-model small

Nothing here is product code, and nothing of the reference's sources is copied: the crate is built where it lies (as a
cdylib, through an out-of-tree CARGO_TARGET_DIR and a --crate-type override on the command line) and called through the
very symbol the Swift app calls (bridge.h:11); the model is driven through openai-whisper's public API.
"""
import argparse
import ctypes
import glob
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import logmel_np as L  # noqa: E402  (only for the seeded inputs: synth_chunk)

FRAMES_SEED = 7   # same sampled frames as make_golden.py


def sampled_frames():
    rng = np.random.default_rng(FRAMES_SEED)
    return np.sort(np.concatenate([[0, 1, 2, 2997, 2998, 2999], rng.choice(np.arange(3, 2997), 58, replace=False)]))


def dump_stft(crate_dir):
    """cargo rustc --release --crate-type cdylib in the reference crate, then generate_spectrogram(audio, output) --
    stft/src/lib.rs:110-122 -- on the seeded chunks, with stft.swift:10-12's buffer conventions."""
    target = tempfile.mkdtemp(prefix="stft_target_")
    env = dict(os.environ, CARGO_TARGET_DIR=target)
    subprocess.run(["cargo", "rustc", "--release", "--lib", "--crate-type", "cdylib"], cwd=crate_dir, env=env, check=True)
    so = glob.glob(os.path.join(target, "release", "libstft.*"))
    so = [p for p in so if p.endswith((".so", ".dylib"))]
    assert so, "cargo produced no cdylib under %s" % target
    lib = ctypes.CDLL(so[0])
    out = stft_fixture(lib, "rust stft crate, cargo release build, realfft per Cargo.lock")
    path = os.path.join(HERE, "ref_stft_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    compare_with_oracle(out, "the Rust crate")


def compare_with_oracle(ref, who):
    """The comparison the pinned tests will make, printed on the spot: the reference's outputs just recorded against this
    repo's C restatement (oracle/logmel_ref.c, built on demand with gcc) on the same seeded inputs.  What to expect: two f64
    implementations of lib.rs:22-122 that differ only in the FFT (realfft's mixed-radix 400-point plan vs a direct DFT)
    agree to ~1e-12 on the log-mel values and bit for bit on the reflect pads; tests/test_reference_goldens.py gates 1e-9."""
    so = os.path.join(ROOT, "oracle", "liboracle_logmel.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    mine = stft_fixture(ctypes.CDLL(so), "oracle/logmel_ref.c")
    worst = 0.0
    print("oracle (oracle/logmel_ref.c) vs %s, same seeded inputs:" % who)
    for key in sorted(k for k in ref if k.endswith("_cols")):
        case = key[:-5]
        d_cols = float(np.abs(np.asarray(ref[key]) - mine[key]).max())
        d_sum = float(np.abs(np.asarray(ref[case + "_sum"]) - mine[case + "_sum"]).max())
        pads = bool(np.array_equal(ref[case + "_pad_head"], mine[case + "_pad_head"]) and
                    np.array_equal(ref[case + "_pad_tail"], mine[case + "_pad_tail"]))
        worst = max(worst, d_cols)
        print("  %-7s max |log-mel diff| over %d sampled frames x 80 bins: %.3e   checksum diff: %.3e   reflect pads bit-equal: %s"
              % (case, len(ref["frames"]), d_cols, d_sum, pads))
    print("  worst %.3e -- %s (gate of tests/test_reference_goldens.py: 1e-9).  Now commit tests/golden/ref_stft_golden.npz: "
          "`pytest tests/test_reference_goldens.py` stops skipping and pins the oracle, and through it the HIP path (-m gpu), "
          "to the reference itself." % (worst, "PINNED" if worst <= 1e-9 else "ABOVE THE GATE: the restatement differs from the crate"))
    return worst


def stft_fixture(lib, source):
    """The front-end fixture from ANY library that exports the reference's symbol (bridge.h:11).  dump_stft calls it with the
    Rust crate's cdylib; tests/test_reference_goldens.py calls it with the ORACLE's library, so that a change of the schema
    here -- or of what the consuming tests read -- is caught on the CPU, before anyone with cargo runs this script."""
    lib.generate_spectrogram.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.generate_spectrogram.restype = None
    frames = sampled_frames()
    out = {"frames": frames, "source": np.array(source)}
    cases = {"noise0": L.synth_chunk(0), "noise1": L.synth_chunk(1), "zeros": np.zeros(480000, np.float32),
             "ones": np.ones(480000, np.float32)}
    for name, x in cases.items():
        buf = np.zeros(480400, np.float64)                    # stft.swift:10-11: 200 zeros each side
        buf[200:480200] = x.astype(np.float64)                # ContentView.swift:59: Float -> Double
        res = np.zeros(240000, np.float64)                    # stft.swift:12
        lib.generate_spectrogram(buf.ctypes.data_as(ctypes.c_void_p), res.ctypes.data_as(ctypes.c_void_p))
        y = res.reshape(80, 3000)
        out[name + "_cols"] = y[:, frames]
        out[name + "_sum"] = np.array([y.sum(), np.abs(y).sum(), (y * y).sum(), y.max(), y.min()])
        out[name + "_pad_head"] = buf[:200].copy()            # lib.rs:34-40 mutates the caller's buffer
        out[name + "_pad_tail"] = buf[480200:].copy()
    return out


def dump_whisper(name):
    """whisper.load_model(name) on CPU in fp32 -- whisper_to_cml.py:6-8 -- then the two traced graphs on a seeded input:
    encoder (1,80,3000) -> (1,1500,d) (:10-23) and decoder tokens (1,T) + audio features -> logits (:25-43)."""
    import torch
    import whisper
    model = whisper.load_model(name).cpu().float().eval()
    dims = {k: int(v) for k, v in vars(model.dims).items()}
    sd = model.state_dict()
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].cpu().float().numpy().tobytes())
    x = L.synth_chunk(0)
    mel = whisper.log_mel_spectrogram(torch.from_numpy(x))[None]            # whisper's own front end: (1, n_mels, 3000)
    with torch.no_grad():
        xa = model.encoder(mel)
        sot = 50258 if dims["n_vocab"] >= 51865 else 50257
        toks = torch.tensor([[sot, sot + 1, sot + 101, sot + 105]]) if dims["n_vocab"] >= 51865 else torch.tensor([[50257, 50362]])
        logits = model.decoder(toks, xa)
    out = model_fixture(name, dims, h.hexdigest(), mel[0].numpy(), xa.numpy(), toks.numpy(), logits.numpy())
    path = os.path.join(HERE, "ref_whisper_%s_golden.npz" % name.replace(".", "_"))
    np.savez_compressed(path, **out)
    try:   # the same on-the-spot comparison for the model half: the oracle (oracle/whisper_ref.py) on the checkpoint's own weights
        from oracle import whisper_ref as R
        sd_np = {k: v.cpu().float().numpy() for k, v in sd.items()}
        o_xa = R.encode(R.to_torch(sd_np), dims, mel.numpy()).numpy()
        o_lg = R.decode_logits(R.to_torch(sd_np), dims, toks.numpy().astype(np.int32), o_xa).numpy()
        print("oracle (oracle/whisper_ref.py) vs openai-whisper %s: encoder rel-L2 %.3e, logits rel-L2 %.3e (gates of "
              "tests/test_reference_goldens.py: 1e-4)" % (name, R.rel_l2(o_xa, xa.numpy()), R.rel_l2(o_lg, logits.numpy())))
    except Exception as e:   # the fixture is what matters; the comparison is a convenience
        print("on-the-spot oracle comparison skipped:", repr(e))
    print("wrote", path, "-- convert the same checkpoint with openai-whisper-coreml_amd/weights.py:convert_openai_pt and set "
          "WM_REF_WEIGHTS=<flat file> so that tests/test_reference_goldens.py can load it")


def model_fixture(name, dims, state_sha256, mel, xa, toks, logits):
    """The model fixture from numpy arrays: mel (n_mels, 3000), xa (1, 1500, d), toks (1, T), logits (1, T, n_vocab).
    dump_whisper fills it from openai-whisper; the CPU schema test fills it from the oracle."""
    rows = np.array([0, 1, 2, 3, 100, 500, 749, 750, 1000, 1496, 1497, 1498, 1499])
    return {"model": np.array(name), "dims_keys": np.array(sorted(dims)), "dims_vals": np.array([dims[k] for k in sorted(dims)]),
            "state_sha256": np.array(state_sha256), "mel": np.asarray(mel, np.float32), "rows": rows,
            "xa_rows": xa[0, rows], "xa_sum": np.array([float(xa.sum()), float(np.abs(xa).sum())]),
            "tokens": np.asarray(toks, np.int32),
            "logits_lang": logits[0, 0, 50259:50358] if dims["n_vocab"] >= 51865 else np.zeros(0),
            "logits_head": logits[0, :, :256], "logits_argmax": logits[0].argmax(-1),
            "logits_sum": np.array([float(logits.sum()), float(np.abs(logits).sum())])}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stft-crate", help="path to the reference's stft/ crate directory")
    ap.add_argument("--whisper-model", help='openai-whisper model name, e.g. "small" (whisper_to_cml.py:7)')
    a = ap.parse_args()
    if not a.stft_crate and not a.whisper_model:
        ap.error("nothing to do: give --stft-crate and / or --whisper-model")
    if a.stft_crate:
        dump_stft(a.stft_crate)
    if a.whisper_model:
        dump_whisper(a.whisper_model)
