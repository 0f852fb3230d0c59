"""oracle/whisper_ref.py -- CPU ORACLE for boundary #2 (test infrastructure, NOT product).

PyTorch fp32 restatement of the model the reference exports to CoreML:
`whisper.load_model("small")` -> `model.encoder`, `model.decoder`
(/root/reference/whisper_to_cml.py:6-8,15,32), i.e. openai-whisper's AudioEncoder /
TextDecoder / ResidualAttentionBlock / MultiHeadAttention,
plus the Swift caller's post-processing (Whisper/Whisper/Whisper.swift:33-40).

PARITY UNPINNED: openai-whisper is an un-vendored, un-pinned dependency of the reference,
no checkpoint exists offline, and the reference holds no tests or golden outputs for this
boundary.  This restatement is anchored instead on (a) the published module structure,
(b) a one-time cross-check against the independent implementation in `transformers`
(tests/test_oracle_model.py), and (c) the shapes / token constants hard-wired in the
reference (whisper_to_cml.py:13,28-29; Whisper.swift:25,34-37).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

N_FRAMES = 3000


def _t(sd, name):
    v = sd[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def to_torch(sd):
    return {k: _t(sd, k).float() for k in sd}


def _ln(x, sd, prefix):
    return F.layer_norm(x.float(), (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def _linear(x, sd, prefix, bias=True):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"] if bias else None)


def _qkv_attention(q, k, v, n_head, mask=None):
    """openai-whisper MultiHeadAttention.qkv_attention: q,k scaled by hd**-0.25 each,
    softmax in fp32."""
    B, T, D = q.shape
    scale = (D // n_head) ** -0.25
    q = q.view(B, T, n_head, -1).permute(0, 2, 1, 3) * scale
    k = k.view(B, k.shape[1], n_head, -1).permute(0, 2, 3, 1) * scale
    v = v.view(B, v.shape[1], n_head, -1).permute(0, 2, 1, 3)
    qk = q @ k
    if mask is not None:
        qk = qk + mask[:T, :T]
    w = F.softmax(qk.float(), dim=-1)
    return (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)


def _mha(x, sd, prefix, n_head, xa=None, mask=None, kv=None):
    q = _linear(x, sd, prefix + ".query")
    if kv is None:
        src = x if xa is None else xa
        k = _linear(src, sd, prefix + ".key", bias=False)   # key has no bias
        v = _linear(src, sd, prefix + ".value")
    else:
        k, v = kv
    return _linear(_qkv_attention(q, k, v, n_head, mask), sd, prefix + ".out")


def _block(x, sd, prefix, n_head, xa=None, mask=None):
    x = x + _mha(_ln(x, sd, prefix + ".attn_ln"), sd, prefix + ".attn", n_head, mask=mask)
    if xa is not None:
        x = x + _mha(_ln(x, sd, prefix + ".cross_attn_ln"), sd, prefix + ".cross_attn", n_head, xa=xa)
    h = F.gelu(_linear(_ln(x, sd, prefix + ".mlp_ln"), sd, prefix + ".mlp.0"))   # exact (erf) GELU
    return x + _linear(h, sd, prefix + ".mlp.2")


@torch.no_grad()
def encode(sd, dims, mel):
    """AudioEncoder.forward: mel f32 [B, n_mels, 3000] -> [B, 1500, d]
    (the traced graph of whisper_to_cml.py:10-23; Swift side Whisper.swift:25-29)."""
    x = torch.as_tensor(mel, dtype=torch.float32)
    x = F.gelu(F.conv1d(x, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    x = x + sd["encoder.positional_embedding"]
    for i in range(dims["n_audio_layer"]):
        x = _block(x, sd, f"encoder.blocks.{i}", dims["n_audio_head"])
    return _ln(x, sd, "encoder.ln_post")


@torch.no_grad()
def decode_logits(sd, dims, tokens, xa):
    """TextDecoder.forward without kv_cache (offset 0), as traced at whisper_to_cml.py:25-43:
    tokens [B, T] int, xa [B, 1500, d] -> logits f32 [B, T, n_vocab]."""
    tokens = torch.as_tensor(tokens, dtype=torch.long)
    xa = torch.as_tensor(xa, dtype=torch.float32)
    T = tokens.shape[1]
    x = sd["decoder.token_embedding.weight"][tokens] + sd["decoder.positional_embedding"][:T]
    mask = torch.full((dims["n_text_ctx"], dims["n_text_ctx"]), float("-inf")).triu_(1)
    for i in range(dims["n_text_layer"]):
        x = _block(x, sd, f"decoder.blocks.{i}", dims["n_text_head"], xa=xa, mask=mask)
    x = _ln(x, sd, "decoder.ln")
    return (x @ sd["decoder.token_embedding.weight"].T).float()


def detect_language(sd, dims, xa, sot=50258, lang_first=50259, lang_last=50357):
    """Whisper.swift:33-40: SOT token -> logits -> first arg-max over the language ids."""
    B = xa.shape[0]
    logits = decode_logits(sd, dims, np.full((B, 1), sot), xa)[:, 0]
    conf = logits[:, lang_first:lang_last + 1]
    return conf.argmax(dim=1).numpy().astype(np.int32), conf.numpy()   # torch argmax: first max


@torch.no_grad()
def timestamp_filter(row, seq, ts_begin, eot, max_initial=-1, sum_rule=True):
    """openai-whisper ApplyTimestampRules (whisper/decoding.py [3p]) for ONE sequence: `row` = logits after the
    suppress filters (1-D float tensor, modified in place), `seq` = the tokens sampled so far.  Returns
    (forced, gap): whether the summed timestamp probability forced a timestamp, and logsumexp(timestamps) - max(text)."""
    last_was = len(seq) >= 1 and seq[-1] >= ts_begin
    penult = len(seq) < 2 or seq[-2] >= ts_begin
    if last_was:
        if penult:
            row[ts_begin:] = float("-inf")      # has to be non-timestamp
        else:
            row[:eot] = float("-inf")           # cannot be normal text tokens
    tss = [t for t in seq if t >= ts_begin]
    if tss:
        last = tss[-1] if (last_was and not penult) else tss[-1] + 1
        row[ts_begin:last] = float("-inf")      # timestamps shouldn't decrease
    if len(seq) == 0:
        row[:ts_begin] = float("-inf")          # suppress generating non-timestamp tokens at the beginning
        if max_initial is not None and max_initial >= 0:
            row[ts_begin + max_initial + 1:] = float("-inf")
    lp = torch.log_softmax(row.float(), dim=-1)
    ts_lp = torch.logsumexp(lp[ts_begin:], dim=-1)
    text_lp = lp[:ts_begin].max()
    forced = bool(ts_lp > text_lp)
    gap = float(ts_lp - text_lp) if torch.isfinite(ts_lp) and torch.isfinite(text_lp) else float("inf")
    if forced and sum_rule:   # sum_rule=False: structural rules only (tests use it to judge near-ties of the sum rule)
        row[:ts_begin] = float("-inf")
    return forced, gap


def greedy(sd, dims, xa, prompt, max_new, eot=-1, suppress=(), suppress_first=(), ts_rules=None):
    """KV-cached greedy decode (the extension BASELINE.json asks for): returns tokens
    [B, max_new] (padded with eot after a stop), lens [B], and the per-step logits (filtered).
    `suppress` / `suppress_first` restate openai-whisper's SuppressTokens / SuppressBlank logit filters
    (whisper/decoding.py [3p]: logits[:, ids] = -inf, the blank filter only at the first sampled position)."""
    xa = torch.as_tensor(xa, dtype=torch.float32)
    B = xa.shape[0]
    L, H = dims["n_text_layer"], dims["n_text_head"]
    cross = []
    for i in range(L):
        p = f"decoder.blocks.{i}.cross_attn"
        cross.append((_linear(xa, sd, p + ".key", bias=False), _linear(xa, sd, p + ".value")))
    self_k = [None] * L
    self_v = [None] * L
    toks = torch.tensor(np.tile(np.asarray(prompt, dtype=np.int64), (B, 1)))
    out = np.full((B, max_new), eot, dtype=np.int32)
    lens = np.zeros(B, dtype=np.int32)
    done = np.zeros(B, dtype=bool)
    all_logits = []
    pos = 0
    cur = toks
    for step in range(max_new):
        T = cur.shape[1]
        x = sd["decoder.token_embedding.weight"][cur] + sd["decoder.positional_embedding"][pos:pos + T]
        n_ctx = pos + T
        mask = torch.full((n_ctx, n_ctx), float("-inf")).triu_(1)[pos:pos + T]
        for i in range(L):
            p = f"decoder.blocks.{i}"
            h = _ln(x, sd, p + ".attn_ln")
            k_new = _linear(h, sd, p + ".attn.key", bias=False)
            v_new = _linear(h, sd, p + ".attn.value")
            self_k[i] = k_new if self_k[i] is None else torch.cat([self_k[i], k_new], 1)
            self_v[i] = v_new if self_v[i] is None else torch.cat([self_v[i], v_new], 1)
            q = _linear(h, sd, p + ".attn.query")
            Bq, Tq, D = q.shape
            sc = (D // H) ** -0.25
            qh = q.view(Bq, Tq, H, -1).permute(0, 2, 1, 3) * sc
            kh = self_k[i].view(Bq, n_ctx, H, -1).permute(0, 2, 3, 1) * sc
            vh = self_v[i].view(Bq, n_ctx, H, -1).permute(0, 2, 1, 3)
            w = F.softmax((qh @ kh + mask).float(), dim=-1)
            a = (w @ vh).permute(0, 2, 1, 3).flatten(start_dim=2)
            x = x + _linear(a, sd, p + ".attn.out")
            x = x + _mha(_ln(x, sd, p + ".cross_attn_ln"), sd, p + ".cross_attn", H, kv=cross[i])
            hh = F.gelu(_linear(_ln(x, sd, p + ".mlp_ln"), sd, p + ".mlp.0"))
            x = x + _linear(hh, sd, p + ".mlp.2")
        x = _ln(x[:, -1:], sd, "decoder.ln")
        logits = (x @ sd["decoder.token_embedding.weight"].T).float()[:, 0]
        if len(suppress):
            logits[:, list(suppress)] = float("-inf")
        if step == 0 and len(suppress_first):
            logits[:, list(suppress_first)] = float("-inf")
        if ts_rules is not None:   # dict(ts_begin=, eot=, max_initial=): decoding WITH timestamps
            for b in range(B):
                timestamp_filter(logits[b], [int(t) for t in out[b, :step]], ts_rules["ts_begin"], ts_rules["eot"],
                                 ts_rules.get("max_initial", -1))
        all_logits.append(logits.numpy())
        nxt = logits.argmax(dim=1)
        for b in range(B):
            if not done[b]:
                out[b, step] = int(nxt[b])
                lens[b] = step + 1
                if eot >= 0 and int(nxt[b]) == eot:
                    done[b] = True
        pos += T
        cur = nxt[:, None]
        if done.all():
            break
    return out, lens, np.stack(all_logits, axis=1)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


TINY_DIMS = dict(n_mels=80, n_audio_ctx=1500, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                 n_vocab=1024, n_text_ctx=448, n_text_state=128, n_text_head=2, n_text_layer=2)


def smoke_check(pkg):
    """Used by __graft_entry__.smoke(): a tiny synthetic model end to end on the GPU
    (front end -> encoder -> 6-step greedy decode) against this oracle."""
    import importlib
    from oracle import logmel_np
    w = importlib.import_module("openai_whisper_coreml_amd.weights")
    dims = dict(TINY_DIMS)
    sd_np = w.synthetic_state_dict(dims, seed=1, matrix_gain=4.0)   # the `lively` recipe: tokens depend on audio and history
    sd = to_torch(sd_np)
    ctx = pkg.binding.Context(dims)
    ctx.load_state_dict(sd_np)
    ctx.finalize()
    n = np.arange(480000, dtype=np.float64)
    pcm = (0.3 * np.sin(2 * np.pi * 570 * n / 16000) * (0.5 + 0.5 * np.sin(2 * np.pi * 0.4 * n / 16000))).astype(np.float32)[None, :]
    mel = ctx.logmel(pcm, out_dtype=np.float32)
    xa_gpu = ctx.encode_mel(mel)
    xa_ref = encode(sd, dims, mel).numpy()
    e = rel_l2(xa_gpu, xa_ref)
    assert e < 3e-2, "encoder rel-L2 %g" % e
    prompt = [1, 2]
    toks, lens = ctx.transcribe_greedy(pcm, prompt, 12, eot=-1)
    ref_t, _, ref_logits = greedy(sd, dims, xa_gpu, prompt, 12)
    # the GPU-chosen token must be (near-)maximal under the fp32 oracle's logits
    for s in range(12):
        if toks[0, s] != ref_t[0, s]:
            gap = ref_logits[0, s].max() - ref_logits[0, s, toks[0, s]]
            assert gap < 0.05, "step %d: token %d vs %d, oracle logit gap %g" % (s, toks[0, s], ref_t[0, s], gap)
            break
    print("smoke: tiny model encoder rel-L2 = %.3g; greedy tokens %s (oracle %s)" % (e, toks[0].tolist(), ref_t[0].tolist()))
    ctx.close()
