"""Parameter naming, synthetic weights, the flat weight file, and checkpoint converters.

Names and shapes are openai-whisper's state-dict keys -- the model that
whisper_to_cml.py:6-8 loads (`whisper.load_model("small")`) and traces.  The reference
ships no weights and none exist offline, so benchmarks and tests use deterministic
synthetic weights (BASELINE.md "Inputs"); `convert_openai_pt` / `convert_hf_safetensors`
let anyone with a real checkpoint produce the flat file `wm_load_weights` reads.

Flat weight file (little endian):
    magic   8 bytes  b"WMI355X1"
    dims    10 x int32   (wm_dims field order)
    count   int32
    repeated `count` times:
        name_len int32, name bytes (utf-8), n_elems int64, data n_elems x float32
"""
import struct

import numpy as np

DIM_FIELDS = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
              "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")
MAGIC = b"WMI355X1"

# parameter kinds for the synthetic generator
K_MATRIX, K_BIAS, K_LN_W, K_LN_B, K_SINUSOID = 0, 1, 2, 3, 4
STD = {K_MATRIX: 0.02, K_BIAS: 0.01}


def tensor_specs(dims):
    """Ordered list of (name, shape, kind).  The ORDER defines the tensor id used by the
    synthetic generator, on the host (here) and on the device (csrc/model.cpp)."""
    d, da = dims["n_text_state"], dims["n_audio_state"]
    out = []

    def block(prefix, dm, cross):
        names = ["attn"] + (["cross_attn"] if cross else [])
        for a in names:
            out.append((f"{prefix}.{a}.query.weight", (dm, dm), K_MATRIX))
            out.append((f"{prefix}.{a}.query.bias", (dm,), K_BIAS))
            out.append((f"{prefix}.{a}.key.weight", (dm, dm), K_MATRIX))      # key has no bias
            out.append((f"{prefix}.{a}.value.weight", (dm, dm), K_MATRIX))
            out.append((f"{prefix}.{a}.value.bias", (dm,), K_BIAS))
            out.append((f"{prefix}.{a}.out.weight", (dm, dm), K_MATRIX))
            out.append((f"{prefix}.{a}.out.bias", (dm,), K_BIAS))
            out.append((f"{prefix}.{a}_ln.weight", (dm,), K_LN_W))
            out.append((f"{prefix}.{a}_ln.bias", (dm,), K_LN_B))
        out.append((f"{prefix}.mlp.0.weight", (4 * dm, dm), K_MATRIX))
        out.append((f"{prefix}.mlp.0.bias", (4 * dm,), K_BIAS))
        out.append((f"{prefix}.mlp.2.weight", (dm, 4 * dm), K_MATRIX))
        out.append((f"{prefix}.mlp.2.bias", (dm,), K_BIAS))
        out.append((f"{prefix}.mlp_ln.weight", (dm,), K_LN_W))
        out.append((f"{prefix}.mlp_ln.bias", (dm,), K_LN_B))

    out.append(("encoder.conv1.weight", (da, dims["n_mels"], 3), K_MATRIX))
    out.append(("encoder.conv1.bias", (da,), K_BIAS))
    out.append(("encoder.conv2.weight", (da, da, 3), K_MATRIX))
    out.append(("encoder.conv2.bias", (da,), K_BIAS))
    out.append(("encoder.positional_embedding", (dims["n_audio_ctx"], da), K_SINUSOID))
    for i in range(dims["n_audio_layer"]):
        block(f"encoder.blocks.{i}", da, False)
    out.append(("encoder.ln_post.weight", (da,), K_LN_W))
    out.append(("encoder.ln_post.bias", (da,), K_LN_B))
    out.append(("decoder.token_embedding.weight", (dims["n_vocab"], d), K_MATRIX))
    out.append(("decoder.positional_embedding", (dims["n_text_ctx"], d), K_MATRIX))
    for i in range(dims["n_text_layer"]):
        block(f"decoder.blocks.{i}", d, True)
    out.append(("decoder.ln.weight", (d,), K_LN_W))
    out.append(("decoder.ln.bias", (d,), K_LN_B))
    return out


def sinusoids(length, channels, max_timescale=10000.0):
    """openai-whisper `sinusoids()`: the encoder's fixed positional embedding."""
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float64))
    t = np.arange(length, dtype=np.float64)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


# ------------------------------------------------------------------ synthetic generator --
def _hash32(x):
    """lowbias32 (integer hash); x: uint32 array.  Mirrored in csrc/model_kernels.hip."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


IH_MEAN = 131070          # 4 * 65535 / 2
IH_STD = 37837.22659      # sqrt(4 * (65536^2 - 1) / 12)


def synthetic_values(seed, tensor_id, n, std, bf16_round=True):
    """Irwin-Hall(4) approximation of N(0, std^2), exact integer arithmetic + one f32
    multiply, so the device generator reproduces it bit for bit.  With bf16_round the
    value is rounded to the nearest bf16 (ties to even) so HBM (bf16) and oracle (f32)
    hold IDENTICAL weights."""
    with np.errstate(over="ignore"):
        key = _hash32(np.array([(seed + tensor_id * 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint32))[0]
        i = np.arange(n, dtype=np.uint32)
        h1 = _hash32(i ^ key)
        h2 = _hash32(h1 + np.uint32(0x85EBCA6B))
    s = ((h1 & np.uint32(0xFFFF)).astype(np.int64) + (h1 >> np.uint32(16)).astype(np.int64)
         + (h2 & np.uint32(0xFFFF)).astype(np.int64) + (h2 >> np.uint32(16)).astype(np.int64))
    scale = np.float32(std / IH_STD)
    v = (s - IH_MEAN).astype(np.float32) * scale
    return bf16_round_f32(v) if bf16_round else v


def bf16_round_f32(v):
    """Round f32 -> bf16 (nearest even) -> back to f32."""
    u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def synthetic_state_dict(dims, seed=0, matrix_gain=1.0):
    """Deterministic synthetic weights (same values as wm_init_synthetic(ctx, seed); with matrix_gain != 1 as
    wm_init_synthetic_gain: every matrix except the positional tables is multiplied by the gain before the bf16 rounding)."""
    sd = {}
    for tid, (name, shape, kind) in enumerate(tensor_specs(dims)):
        n = int(np.prod(shape))
        if kind == K_LN_W:
            sd[name] = np.ones(shape, np.float32)
        elif kind == K_LN_B:
            sd[name] = np.zeros(shape, np.float32)
        elif kind == K_SINUSOID:
            sd[name] = sinusoids(shape[0], shape[1])
        else:
            # matrices live in HBM as bf16 -> generate bf16-representable values; biases are f32
            v = synthetic_values(seed, tid, n, STD[kind], bf16_round=False)
            if kind == K_MATRIX and "positional" not in name:
                v = v * np.float32(matrix_gain)
            sd[name] = (bf16_round_f32(v) if kind == K_MATRIX else v).reshape(shape)
    return sd


# The "lively" random-init recipe (SYNTHETIC WEIGHTS: NOT FOR PRODUCTION USE -- benchmarks and token-level tests only): with
# plain N(0, 0.02^2) matrices every chunk and position decodes to the same token, so token cross-checks are blind.  Scaling
# the matrices makes the token streams depend on the audio and on the decode history -- but the gain that does it depends
# on the model WIDTH (a d-wide product of 0.02-sigma weights amplifies by 0.02 g sqrt(d)): gain 4, calibrated at d = 1280,
# left base with 3-4 distinct token rows out of 32 distinct recordings.  Calibrated on an MI355X with
# tools/gpu_lively_gain_probe.py (profiles/r06_lively_gain.txt): the smallest tried gain at which 32 distinct recordings --
# seeded noise, and tones + noise -- decode to (all but at most one of) 32 distinct rows.  (A larger gain is a harsher numerical
# test, not a livelier model: bf16 operand rounding grows with it -- tiny.en's 4-layer encoder is 5.3e-3 rel-L2 from the fp32
# oracle at gain 8, base's 6-layer one 2.2e-2 at gain 12 (attention so peaked that an operand rounding flips its targets) --
# so the smallest adequate gain is the one recorded: 6 below d = 1024 (31 - 32 distinct rows of 32), 4 from there.)
LIVELY_GAIN_BY_WIDTH = {384: 6.0, 512: 6.0, 768: 6.0, 1024: 4.0, 1280: 4.0}


def lively_gain(dims):
    """Matrix gain of the lively random-init model of this geometry (by decoder width; 4 for widths not calibrated)."""
    return LIVELY_GAIN_BY_WIDTH.get(int(dims["n_text_state"]), 4.0)


# ------------------------------------------------------------------ flat weight file ----
def save_flat(path, dims, sd):
    specs = tensor_specs(dims)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<10i", *[dims[k] for k in DIM_FIELDS]))
        f.write(struct.pack("<i", len(specs)))
        for name, shape, _ in specs:
            a = np.ascontiguousarray(np.asarray(sd[name], dtype=np.float32))
            if tuple(a.shape) != tuple(shape):
                raise ValueError("%s: shape %r != %r" % (name, a.shape, shape))
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)))
            f.write(nb)
            f.write(struct.pack("<q", a.size))
            f.write(a.tobytes())


def load_flat(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not a WMI355X1 weight file")
        dims = dict(zip(DIM_FIELDS, struct.unpack("<10i", f.read(40))))
        (count,) = struct.unpack("<i", f.read(4))
        shapes = {n: s for n, s, _ in tensor_specs(dims)}
        sd = {}
        for _ in range(count):
            (ln,) = struct.unpack("<i", f.read(4))
            name = f.read(ln).decode()
            (n,) = struct.unpack("<q", f.read(8))
            sd[name] = np.frombuffer(f.read(4 * n), dtype=np.float32).reshape(shapes[name]).copy()
    return dims, sd


# ------------------------------------------------------------------ converters -----------
def convert_openai_pt(pt_path, out_path):
    """openai-whisper checkpoint ({"dims": ..., "model_state_dict": ...}) -> flat file.
    (What `whisper.load_model` at whisper_to_cml.py:7 downloads.)"""
    import torch
    ck = torch.load(pt_path, map_location="cpu")
    dims = {k: int(ck["dims"][k]) for k in DIM_FIELDS}
    sd = {k: v.float().numpy() for k, v in ck["model_state_dict"].items()}
    save_flat(out_path, dims, sd)
    return dims


def hf_to_openai_key(k):
    """HF transformers WhisperModel key -> openai-whisper key (None if unused)."""
    if k.startswith("model."):
        k = k[len("model."):]
    rep = [("layers.", "blocks."), ("self_attn_layer_norm", "attn_ln"), ("self_attn.", "attn."),
           ("encoder_attn_layer_norm", "cross_attn_ln"), ("encoder_attn.", "cross_attn."),
           ("final_layer_norm", "mlp_ln"), ("fc1", "mlp.0"), ("fc2", "mlp.2"),
           ("q_proj", "query"), ("k_proj", "key"), ("v_proj", "value"), ("out_proj", "out"),
           ("embed_tokens", "token_embedding"), ("encoder.embed_positions.weight", "encoder.positional_embedding"),
           ("decoder.embed_positions.weight", "decoder.positional_embedding"),
           ("encoder.layer_norm", "encoder.ln_post"), ("decoder.layer_norm", "decoder.ln")]
    for a, b in rep:
        k = k.replace(a, b)
    if k.startswith("proj_out"):
        return None
    return k


def openai_to_hf_state_dict(sd):
    """Inverse mapping (used by the cross-check test against transformers' Whisper)."""
    inv = [("blocks.", "layers."), ("cross_attn_ln", "ENCLN"), ("cross_attn.", "ENCATTN."),
           ("attn_ln", "self_attn_layer_norm"), ("attn.", "self_attn."), ("ENCLN", "encoder_attn_layer_norm"),
           ("ENCATTN.", "encoder_attn."), ("mlp_ln", "final_layer_norm"),
           ("mlp.0", "fc1"), ("mlp.2", "fc2"), ("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"),
           (".out.", ".out_proj."), ("token_embedding", "embed_tokens"),
           ("encoder.positional_embedding", "encoder.embed_positions.weight"),
           ("decoder.positional_embedding", "decoder.embed_positions.weight"),
           ("encoder.ln_post", "encoder.layer_norm"), ("decoder.ln.", "decoder.layer_norm.")]
    out = {}
    for k, v in sd.items():
        for a, b in inv:
            k = k.replace(a, b)
        out["model." + k] = v
    return out


def convert_hf_safetensors(st_path, dims, out_path):
    from safetensors.numpy import load_file
    raw = load_file(st_path)
    sd = {}
    for k, v in raw.items():
        ok = hf_to_openai_key(k)
        if ok is not None:
            sd[ok] = v.astype(np.float32)
    save_flat(out_path, dims, sd)
