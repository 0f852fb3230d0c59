"""Chunk-parallel sharding of a long recording over the GPUs of one node.

Independent 30 s chunks are the natural shard of this path: the front
end's only reduction is a per-chunk max (stft/src/lib.rs:82-88) and the reference never
conditions one window on another (Whisper.swift:33-40), so ranks own disjoint contiguous
blocks of chunks, no data-path collective is needed, and the only exchange is ONE
all-gather of the decoded token streams (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
import numpy as np

N_SAMPLES = 16000 * 30


def chunk_pcm(samples, pad_value=0):
    """Split a mono 16 kHz recording into 30 s chunks; the last one is zero-padded -- the
    reference's rule for a short recording (ContentView.swift:57-60) applied per window."""
    samples = np.asarray(samples)
    n = max(1, -(-len(samples) // N_SAMPLES))
    out = np.full((n, N_SAMPLES), pad_value, dtype=samples.dtype)
    flat = out.reshape(-1)
    flat[:len(samples)] = samples
    return out


def partition(n_chunks, world_size, rank):
    """Contiguous block partition: rank r gets [r*ceil(N/R), min(N, (r+1)*ceil(N/R)))."""
    per = -(-n_chunks // world_size)
    lo = min(n_chunks, rank * per)
    hi = min(n_chunks, (rank + 1) * per)
    return lo, hi


def gather_tokens(dist, tokens, lens, n_chunks, world_size, device=None):
    """All-gather per-rank token streams into chunk order.

    tokens: int32 [n_local][max_new], lens: int32 [n_local] for this rank's block.
    Every rank contributes a fixed-stride [per][1 + max_new] int32 tensor (length + tokens;
    short blocks are padded), i.e. ~13.5 KB per rank for 1 h of audio on 8 GPUs.
    Returns (tokens [n_chunks][max_new], lens [n_chunks]) on every rank."""
    import torch
    per = -(-n_chunks // world_size)
    max_new = tokens.shape[1] if tokens.ndim == 2 else 0
    buf = np.zeros((per, 1 + max_new), dtype=np.int32)
    n_local = len(lens)
    buf[:n_local, 0] = lens
    if n_local:
        buf[:n_local, 1:] = tokens
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world_size)]
    dist.all_gather(out, t)
    allb = torch.cat([o.cpu() for o in out], dim=0).numpy()[:n_chunks]
    return allb[:, 1:].copy(), allb[:, 0].copy()
