#!/usr/bin/env python3
"""Regenerates tests/golden/model_golden.npz from the fp32 oracle (oracle/whisper_ref.py) on
seeded synthetic weights (tiny geometry so the fixture stays small).  The reference has no
model-side goldens ("parity unpinned"); this fixture pins the ORACLE, which was cross-checked
against `transformers` (tests/test_oracle_model.py).

    python tests/golden/make_model_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import logmel_np as L  # noqa: E402
from oracle import whisper_ref as R  # noqa: E402

W = importlib.import_module("openai_whisper_coreml_amd.weights")


def main():
    seed = 21
    dims = dict(R.TINY_DIMS)
    sd = R.to_torch(W.synthetic_state_dict(dims, seed))
    m80 = np.load(os.path.join(HERE, "m80.npy")).reshape(80, 201)
    mel = L.log_mel(L.synth_chunk(0), m80).astype(np.float32)
    xa = R.encode(sd, dims, mel[None]).numpy()[0]
    rows = np.array([0, 1, 2, 700, 1498, 1499])
    tokens = np.array([10, 21, 3, 500, 77], np.int32)
    lg = R.decode_logits(sd, dims, tokens[None], xa[None]).numpy()[0]
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), seed=seed, mel=mel.astype(np.float16).astype(np.float32),
                        rows=rows, xa_rows=None, tokens=tokens)
    # mel is stored as fp16-rounded f32 to keep the file small: recompute the outputs from exactly that input
    mel16 = mel.astype(np.float16).astype(np.float32)
    xa = R.encode(sd, dims, mel16[None]).numpy()[0]
    lg = R.decode_logits(sd, dims, tokens[None], xa[None]).numpy()[0]
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), seed=seed, mel=mel16.astype(np.float16), rows=rows,
                        xa_rows=xa[rows], tokens=tokens, logits_head=lg[:, :64], logits_sum=lg.sum())
    print("wrote model_golden.npz", os.path.getsize(os.path.join(HERE, "model_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
