"""GPU probe: how much does a common-mode offset of the decoder's residual stream cost the LayerNorm-folded GEMVs?
(the fold multiplies the RAW bf16 residual: its rounding error scales with |x|, not with |x - mean|)"""
import importlib, sys
import numpy as np
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
from oracle import whisper_ref as R
from oracle import logmel_np as L
W = importlib.import_module("openai_whisper_coreml_amd.weights")
dims = dict(R.TINY_DIMS)
for off, outl in ((0.0, 0.0), (1.0, 0.0), (3.0, 0.0), (10.0, 0.0), (0.0, 30.0), (3.0, 30.0)):
    sd = W.synthetic_state_dict(dims, seed=11)
    rng = np.random.default_rng(0)
    for k in sd:
        if "ln" in k and k.endswith("weight"): sd[k] = (1 + 0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
        if "ln" in k and k.endswith("bias"): sd[k] = (0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
    pe = sd["decoder.positional_embedding"].copy()
    pe += off                      # common-mode offset: every feature, every position
    pe[:, 5] += outl               # one massive-activation feature
    sd["decoder.positional_embedding"] = pe.astype(np.float32)
    ctx = pkg.binding.Context(dims); ctx.load_state_dict(sd); ctx.finalize()
    sdt = R.to_torch({n: ctx.get_tensor(n, s) for n, s, _ in W.tensor_specs(dims)})
    pcm = np.stack([L.synth_chunk(3)])
    mel = ctx.logmel(pcm, out_dtype=np.float32)
    xa = R.encode(sdt, dims, mel).numpy()
    tok = np.array([[10, 21, 5, 7, 100, 200]], np.int32)
    lg = ctx.decode_logits(tok, xa)
    ref = R.decode_logits(sdt, dims, tok, xa).numpy()
    print("offset %5.1f outlier %5.1f : logits rel-L2 %.3e  max|err| %.3e (max |ref| %.2f)" % (off, outl, R.rel_l2(lg, ref), np.abs(lg - ref).max(), np.abs(ref).max()))
    ctx.close()
