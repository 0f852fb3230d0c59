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
    if allb.size and (allb[:, 0].min() < 0 or allb[:, 0].max() > max_new):   # written by other ranks (wm_multi_unpack_tokens)
        raise ValueError("gather_tokens: a rank sent a length outside [0, %d]: corrupt token payload" % max_new)
    return allb[:, 1:].copy(), allb[:, 0].copy()


def plan_groups(n_steps, fuse, inflight):
    """Deterministic decode-group plan of a run of `n_steps` independent batches: consecutive batches are
    decoded together in groups of at most `fuse`, and with few steps the groups shrink so that each of the
    `inflight` lanes still gets one (20 steps, fuse 12, 3 lanes -> [7, 7, 6]).  More groups than lanes come in whole
    rounds of the lanes (72 steps, fuse 16, 3 lanes -> 6 x 12, not 15/15/14/14/14: five groups on three lanes leave one
    lane idle for the last third of the run -- measured 2287 vs 2243 audio-s/s).  A pure function of its arguments, so
    every rank of a job forms the same groups in the same order -- the collective that follows (gather_tokens) therefore
    has the same shape on every rank whatever order the lanes finish in."""
    n_steps, fuse, inflight = int(n_steps), max(1, int(fuse)), max(1, int(inflight))
    if n_steps <= 0:
        return []
    f = min(fuse, max(1, -(-n_steps // inflight)))
    n_groups = -(-n_steps // f)
    if n_groups > inflight:
        n_groups = min(n_steps, -(-n_groups // inflight) * inflight)
    base, rem = divmod(n_steps, n_groups)          # balanced: sizes differ by at most one
    return [base + (1 if g < rem else 0) for g in range(n_groups)]


def run_grouped(plan, inflight, run_group, rows_per_step, max_new, dist=None, world_size=1, device=None):
    """Run the decode groups of `plan` (plan_groups) on `inflight` host threads -- worker w drives lane w --
    and exchange the token streams ONCE, after the last group, with a fixed-stride all-gather.

    run_group(worker, group_index, k) -> (tokens int32 [k * rows_per_step][max_new], lens int32 [k * rows_per_step]).
    Worker w starts with group w (so that a repeated run -- warm-up, then the timed pass -- puts the same group sizes on
    the same lanes and finds their captured decode graphs); after that, which worker takes which group is a race (a
    queue of group indices).  Nothing observable depends on it: the results are stored by group index and concatenated
    in plan order before the collective.
    Returns (per_group results in plan order, gathered (tokens, lens) or None when dist is None)."""
    import queue
    import threading
    n_workers = max(1, int(inflight))
    todo = queue.Queue()
    for g in range(n_workers, len(plan)):
        todo.put(g)
    results = [None] * len(plan)
    errors = []

    def worker(w):
        first = True
        while True:
            if first and w < len(plan):
                g = w                      # deterministic first assignment
            else:
                try:
                    g = todo.get_nowait()
                except queue.Empty:
                    return
            first = False
            try:
                results[g] = run_group(w, g, plan[g])
            except BaseException as e:   # surfaced on the caller's thread: a dead worker must not hang the job
                errors.append(e)
                return

    th = [threading.Thread(target=worker, args=(w,)) for w in range(n_workers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]
    gathered = None
    if dist is not None:
        n_local = sum(plan) * rows_per_step
        if n_local:
            toks = np.concatenate([np.asarray(r[0], np.int32).reshape(-1, max_new) for r in results], axis=0)
            lens = np.concatenate([np.asarray(r[1], np.int32).reshape(-1) for r in results], axis=0)
        else:
            toks, lens = np.zeros((0, max_new), np.int32), np.zeros((0,), np.int32)
        assert toks.shape[0] == n_local and lens.shape[0] == n_local, "run_group returned the wrong number of rows"
        gathered = gather_tokens(dist, toks, lens, n_local * world_size, world_size, device=device)
    return results, gathered


# ---------------------------------------------------------------------------------- host placement (one process per GPU)
def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def numa_node_of_pci(bdf, sysfs_root="/sys"):
    """NUMA node of a PCI device ('0000:05:00.0', as hipDeviceGetPCIBusId prints it); None when the platform reports none
    (-1: single-node boxes, most VMs) or the device is not in sysfs."""
    import os
    try:
        n = int(open(os.path.join(sysfs_root, "bus", "pci", "devices", bdf.lower(), "numa_node")).read().strip())
    except (OSError, ValueError):
        return None
    return n if n >= 0 else None


def numa_cpus_for_rank(bdfs, local_rank, allowed=None, sysfs_root="/sys", min_cpus=4):
    """Cores for rank `local_rank` of a one-process-per-GPU job: the cores of its GPU's NUMA node, split evenly among the
    ranks whose GPUs sit on the same node (in rank order), intersected with `allowed` (the process's current affinity).
    bdfs: PCI bus ids of the job's GPUs, indexed by local rank.  Returns a sorted list, or None = "do not pin" (no NUMA
    information, or the rank's share would be smaller than min_cpus: each rank runs 3 lane threads + the main thread whose
    whole job is launching microsecond-scale graphs).  A pure function of the files under sysfs_root: CPU-tested with a
    fake topology."""
    import os
    nodes = [numa_node_of_pci(b, sysfs_root) for b in bdfs]
    mine = nodes[local_rank] if 0 <= local_rank < len(nodes) else None
    if mine is None:
        return None
    try:
        cpus = _parse_cpulist(open(os.path.join(sysfs_root, "devices", "system", "node", "node%d" % mine, "cpulist")).read())
    except (OSError, ValueError):
        return None
    if allowed is not None:
        cpus = [c for c in cpus if c in set(allowed)]
    peers = [r for r, n in enumerate(nodes) if n == mine]
    k = peers.index(local_rank)
    per = len(cpus) // len(peers)
    if per < min_cpus:
        return sorted(cpus) if len(cpus) >= min_cpus else None   # too few to split: share the node, or leave the scheduler alone
    return sorted(cpus[k * per:(k + 1) * per])


def pin_to_numa(bdfs, local_rank, sysfs_root="/sys"):
    """Apply numa_cpus_for_rank to this process (all current and future threads inherit it).  Returns the core list or None."""
    import os
    if not hasattr(os, "sched_setaffinity"):
        return None
    cpus = numa_cpus_for_rank(bdfs, local_rank, allowed=sorted(os.sched_getaffinity(0)), sysfs_root=sysfs_root)
    if cpus:
        os.sched_setaffinity(0, cpus)
    return cpus
