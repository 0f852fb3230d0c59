"""GPU probe: full-depth large-v2 (32 + 32 layers, d = 1280) against the torch-fp32 oracle on ONE chunk:
encoder output rel-L2 and teacher-forced decoder logits rel-L2.  ~1 min (weights are downloaded from the GPU)."""
import sys, time, importlib
sys.path.insert(0, '.')
import numpy as np
import torch
import openai_whisper_coreml_amd as pkg
from oracle import whisper_ref as R, logmel_np as L
W = importlib.import_module("openai_whisper_coreml_amd.weights")
name = sys.argv[1] if len(sys.argv) > 1 else "large-v2"
dims = pkg.binding.MODEL_DIMS[name]
t0 = time.time()
ctx = pkg.binding.Context(dims); ctx.init_synthetic(7); ctx.finalize()
pcm = np.stack([L.synth_chunk(21)])
mel = ctx.logmel(pcm, n_mels=dims["n_mels"])
xa = ctx.encode_mel(mel)
print("gpu encode done %.1fs" % (time.time() - t0))
sd = R.to_torch({n: ctx.get_tensor(n, s) for n, s, _ in W.tensor_specs(dims)})
print("weights on host %.1fs" % (time.time() - t0))
torch.set_num_threads(16)
want = R.encode(sd, dims, mel).numpy()
print("oracle encode done %.1fs" % (time.time() - t0))
print("encoder rel-L2 %.3e  max-abs %.3e (|ref| max %.3f)" % (R.rel_l2(xa, want), np.abs(xa - want).max(), np.abs(want).max()))
tok = np.array([[50258, 50259, 50359, 50363]], dtype=np.int32)
got = ctx.decode_logits(tok, want)
ref = R.decode_logits(sd, dims, tok, want).numpy()
print("logits rel-L2 %.3e  max-abs %.3e (|ref| max %.3f)  argmax agree %s" % (
    R.rel_l2(got, ref), np.abs(got - ref).max(), np.abs(ref).max(), (got.argmax(-1) == ref.argmax(-1)).tolist()))
print("total %.1fs" % (time.time() - t0))
