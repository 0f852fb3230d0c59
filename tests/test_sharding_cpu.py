"""CPU tests of the N > 1 path: chunking, block partition and the token all-gather, run with
world_size 2 over gloo (the GPU path uses the same code over RCCL)."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

S = importlib.import_module("openai_whisper_coreml_amd.sharding")


def test_chunking_pads_last_window():
    x = np.arange(480000 * 2 + 5, dtype=np.int16)
    c = S.chunk_pcm(x)
    assert c.shape == (3, 480000)
    assert np.array_equal(c.reshape(-1)[:len(x)], x) and not c[2, 5:].any()
    assert S.chunk_pcm(np.zeros(0, np.float32)).shape == (1, 480000)       # empty input: one silent chunk
    assert S.chunk_pcm(np.ones(480000, np.float32)).shape == (1, 480000)   # exact fit


def test_partition_covers_everything_once():
    for n in (0, 1, 7, 8, 15, 120, 121):
        for w in (1, 2, 3, 8):
            spans = [S.partition(n, w, r) for r in range(w)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n)), (n, w, spans)
    assert S.partition(120, 8, 3) == (45, 60)      # 1 h of audio on 8 GPUs: 15 chunks per rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, max_new, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = S.partition(n_chunks, world, rank)
    # stand-in for the decoder: chunk i "decodes" to tokens 1000*i + j, with length (i % max_new) + 1
    toks = np.array([[1000 * i + j for j in range(max_new)] for i in range(lo, hi)], np.int32).reshape(hi - lo, max_new)
    lens = np.array([(i % max_new) + 1 for i in range(lo, hi)], np.int32)
    all_t, all_l = S.gather_tokens(dist, toks, lens, n_chunks, world)
    q.put((rank, all_t, all_l))
    dist.barrier()
    dist.destroy_process_group()


def test_token_allgather_world2_gloo():
    for n_chunks in (5, 4, 1):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, n_chunks, 6, q)) for r in range(2)]
        for p in ps:
            p.start()
        res = [q.get(timeout=120) for _ in ps]
        for p in ps:
            p.join(timeout=60)
            assert p.exitcode == 0
        want_t = np.array([[1000 * i + j for j in range(6)] for i in range(n_chunks)], np.int32)
        want_l = np.array([(i % 6) + 1 for i in range(n_chunks)], np.int32)
        for _, t, l in res:
            assert np.array_equal(t, want_t) and np.array_equal(l, want_l)


def test_wav_reader_roundtrip(tmp_path):
    A = importlib.import_module("openai_whisper_coreml_amd.audio")
    x = (np.sin(np.arange(16000 * 31) * 0.01) * 12000).astype(np.int16)        # 31 s -> 2 windows
    p = os.path.join(tmp_path, "query.wav")
    A.write_wav_int16(p, x)
    assert np.array_equal(A.read_wav_int16(p), x)
    c = A.wav_to_chunks(p)
    assert c.shape == (2, 480000) and c.dtype == np.int16 and np.array_equal(c[0], x[:480000])
    assert np.array_equal(c[1, :16000], x[480000:]) and not c[1, 16000:].any()
    import wave
    with wave.open(os.path.join(tmp_path, "bad.wav"), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(b"\0" * 8)
    try:
        A.read_wav_int16(os.path.join(tmp_path, "bad.wav"))
        assert False
    except ValueError:
        pass


# ---------------------------------------------------------------- bench.py's N > 1 scheduling (plan_groups / run_grouped)
def test_plan_groups_is_deterministic_and_balanced():
    assert S.plan_groups(20, 12, 3) == [7, 7, 6]            # the driver's `--steps 20` with 3 lanes
    assert S.plan_groups(72, 12, 3) == [12] * 6
    assert S.plan_groups(72, 16, 3) == [12] * 6             # whole rounds of the lanes, not 15/15/14/14/14
    assert S.plan_groups(100, 16, 3) == [12] + [11] * 8
    assert S.plan_groups(4, 1, 3) == [1] * 4                # ... unless there are not enough steps for that
    assert S.plan_groups(5, 12, 3) == [2, 2, 1]
    assert S.plan_groups(1, 12, 3) == [1]
    assert S.plan_groups(0, 12, 3) == []
    assert S.plan_groups(7, 1, 3) == [1] * 7
    for n in range(1, 60):
        for f in (1, 3, 12):
            for s in (1, 2, 3, 4):
                p = S.plan_groups(n, f, s)
                assert sum(p) == n and max(p) <= f and max(p) - min(p) <= 1


def _grouped_worker(rank, world, port, n_steps, fuse, inflight, nb, max_new, q):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = S.plan_groups(n_steps, fuse, inflight)
    starts = np.concatenate([[0], np.cumsum(plan)])[:-1]
    order = []

    def run_group(w, g, k):
        # stand-in for wm_transcribe_greedy: the ranks finish their groups in DIFFERENT orders (rank-dependent sleeps)
        time.sleep(0.01 * (((g * 7 + rank * 3) % 5) + (w if rank else inflight - w)))
        order.append(g)
        rows = [rank * n_steps * nb + (starts[g] + i) * nb + j for i in range(k) for j in range(nb)]
        toks = np.array([[1000 * r + t for t in range(max_new)] for r in rows], np.int32)
        lens = np.array([(r % max_new) + 1 for r in rows], np.int32)
        return toks, lens

    for _ in range(2):   # two runs back to back, as bench.py does (warm-up, then the timed region)
        res, gathered = S.run_grouped(plan, inflight, run_group, nb, max_new, dist=dist, world_size=world)
    q.put((rank, plan, order, gathered[0], gathered[1]))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_grouping_world2_uneven_groups_gloo():
    """bench.py --steps 20 --inflight 3 (groups 7/7/6) on two ranks whose lanes finish in different orders: one
    fixed-shape all-gather after the run, identical result on both ranks (VERDICT r1 weak #3 / ADVICE bench.py:242)."""
    n_steps, fuse, inflight, nb, max_new, world = 20, 12, 3, 2, 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_grouped_worker, args=(r, world, port, n_steps, fuse, inflight, nb, max_new, q))
          for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = n_steps * nb * world
    want_t = np.array([[1000 * r + t for t in range(max_new)] for r in range(n)], np.int32)
    want_l = np.array([(r % max_new) + 1 for r in range(n)], np.int32)
    for _, plan, order, t, l in res:
        assert plan == [7, 7, 6]
        assert np.array_equal(t, want_t) and np.array_equal(l, want_l)


def test_run_grouped_surfaces_worker_errors():
    def boom(w, g, k):
        raise RuntimeError("lane failed")
    try:
        S.run_grouped([1, 1], 2, boom, 1, 3)
        assert False
    except RuntimeError:
        pass


# ---------------------------------------------------------------- wm_multi: the same split behind the C ABI (host-only parts)
def test_c_abi_partition_and_gather_packing_match_the_python_path():
    """wm_multi (csrc/multi.cpp) is the dlopen-only host's road to all 8 GPUs; its block partition and its fixed-stride
    all-gather payload must be exactly what bench.py's torch.distributed path uses (sharding.partition / gather_tokens)."""
    import ctypes
    import openai_whisper_coreml_amd as pkg
    lib = pkg.load_library()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lo, hi = ctypes.c_int(), ctypes.c_int()
    for n in (0, 1, 7, 8, 15, 120, 121):
        for w in (1, 2, 3, 8):
            for r in range(w):
                assert lib.wm_multi_partition(n, w, r, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == S.partition(n, w, r), (n, w, r)
    assert lib.wm_multi_partition(5, 2, 2, ctypes.byref(lo), ctypes.byref(hi)) != 0          # rank out of range
    # pack each rank's block, concatenate (= what ncclAllGather leaves on every rank), unpack: chunk order restored
    n_chunks, world, max_new = 13, 4, 6
    per = -(-n_chunks // world)
    toks = np.array([[1000 * i + j for j in range(max_new)] for i in range(n_chunks)], np.int32)
    lens = np.array([(i % max_new) + 1 for i in range(n_chunks)], np.int32)
    gathered = np.zeros((world, per, 1 + max_new), np.int32)
    for r in range(world):
        a, b = S.partition(n_chunks, world, r)
        t, l = np.ascontiguousarray(toks[a:b]), np.ascontiguousarray(lens[a:b])
        pay = np.full((per, 1 + max_new), -7, np.int32)
        assert lib.wm_multi_pack_tokens(P(t) if b > a else None, P(l) if b > a else None, b - a, per, max_new, P(pay)) == 0
        assert np.array_equal(pay[:b - a, 0], l) and np.array_equal(pay[:b - a, 1:], t) and not pay[b - a:].any()
        gathered[r] = pay
    out_t = np.zeros((n_chunks, max_new), np.int32)
    out_l = np.zeros(n_chunks, np.int32)
    assert lib.wm_multi_unpack_tokens(P(gathered), world, per, max_new, n_chunks, P(out_t), P(out_l)) == 0
    assert np.array_equal(out_t, toks) and np.array_equal(out_l, lens)
    assert lib.wm_multi_unpack_tokens(P(gathered), world, per, max_new, world * per + 1, P(out_t), P(out_l)) != 0


def test_c_abi_pack_unpack_roundtrip_property():
    """Property test (hypothesis): wm_multi_partition / wm_multi_pack_tokens / wm_multi_unpack_tokens -- the host-only pieces
    of the one-process-all-GPUs split -- reproduce every chunk's (length, tokens) for any chunk count, world size and
    max_new, and agree with sharding.partition (the torch.distributed host)."""
    import ctypes
    import pytest
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    import openai_whisper_coreml_amd as pkg
    lib = pkg.load_library()

    @settings(max_examples=80, deadline=None)
    @given(st.integers(0, 130), st.integers(1, 9), st.integers(0, 12), st.integers(0, 2 ** 31 - 1))
    def check(n_chunks, world, max_new, seed):
        rng = np.random.default_rng(seed)
        toks = rng.integers(0, 51866, size=(n_chunks, max_new)).astype(np.int32)
        lens = rng.integers(0, max_new + 1, size=n_chunks).astype(np.int32)
        per = -(-n_chunks // world)
        gathered = np.zeros((world, per, 1 + max_new), np.int32)
        covered = []
        for r in range(world):
            lo, hi = ctypes.c_int(), ctypes.c_int()
            assert lib.wm_multi_partition(n_chunks, world, r, ctypes.byref(lo), ctypes.byref(hi)) == 0
            assert (lo.value, hi.value) == S.partition(n_chunks, world, r)
            covered += list(range(lo.value, hi.value))
            t = np.ascontiguousarray(toks[lo.value:hi.value])
            l = np.ascontiguousarray(lens[lo.value:hi.value])
            assert lib.wm_multi_pack_tokens(t.ctypes.data_as(ctypes.c_void_p), l.ctypes.data_as(ctypes.c_void_p),
                                            hi.value - lo.value, per, max_new,
                                            gathered[r].ctypes.data_as(ctypes.c_void_p)) == 0
        assert covered == list(range(n_chunks))
        out_t = np.full((n_chunks, max_new), -7, np.int32)
        out_l = np.full(n_chunks, -7, np.int32)
        assert lib.wm_multi_unpack_tokens(gathered.ctypes.data_as(ctypes.c_void_p), world, per, max_new, n_chunks,
                                          out_t.ctypes.data_as(ctypes.c_void_p), out_l.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(out_t, toks) and np.array_equal(out_l, lens)

    check()


def test_c_abi_wav_reader_and_chunker(tmp_path):
    """SURVEY 8f rank 1 behind the C ABI (wm_wav_*): the file AudioRecorder.swift:56-61 writes (16 kHz mono 16-bit PCM) in,
    30 s windows out with ContentView.swift:57-60's zero-pad rule per window -- equal to the Python path (audio.py / the
    standard `wave` module), including extra RIFF chunks before the data, odd-sized chunks, and the error cases."""
    import struct
    import wave
    import pytest
    import openai_whisper_coreml_amd as pkg
    A = importlib.import_module("openai_whisper_coreml_amd.audio")
    rng = np.random.default_rng(5)
    for n in (0, 1, 479999, 480000, 480001, 16000 * 75 + 3):
        x = rng.integers(-32768, 32768, size=n).astype(np.int16)
        p = os.path.join(tmp_path, "a%d.wav" % n)
        A.write_wav_int16(p, x)
        w = pkg.binding.Wav(p)
        want = A.wav_to_chunks(p)
        assert w.num_samples == n and w.num_chunks == want.shape[0]
        assert np.array_equal(w.chunks(), want)
        if w.num_chunks > 1:
            assert np.array_equal(w.chunks(1, 1), want[1:2])
        with pytest.raises(pkg.binding.WhisperError, match="outside"):
            w.chunks(w.num_chunks, 1)
        w.close()
    # a LIST chunk of odd size before the data, and a WAVE_FORMAT_EXTENSIBLE header
    x = rng.integers(-3000, 3000, size=1234).astype(np.int16)
    fmt_ext = struct.pack("<HHIIHHHHIH14s", 0xFFFE, 1, 16000, 32000, 2, 16, 22, 16, 4, 1, b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    body = b"WAVE" + b"LIST" + struct.pack("<I", 5) + b"hello" + b"\x00" + b"fmt " + struct.pack("<I", len(fmt_ext)) + fmt_ext \
        + b"data" + struct.pack("<I", x.nbytes) + x.tobytes()
    p = os.path.join(tmp_path, "ext.wav")
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    w = pkg.binding.Wav(p)
    assert w.num_samples == 1234 and np.array_equal(w.chunks()[0, :1234], x) and not w.chunks()[0, 1234:].any()
    w.close()
    # errors: not RIFF, wrong rate, stereo, 8-bit, missing file
    open(os.path.join(tmp_path, "junk.wav"), "wb").write(b"not a wave file at all")
    with pytest.raises(pkg.binding.WhisperError, match="RIFF"):
        pkg.binding.Wav(os.path.join(tmp_path, "junk.wav"))
    for rate, ch, width, tag in ((44100, 1, 2, "rate"), (16000, 2, 2, "stereo"), (16000, 1, 1, "u8")):
        p = os.path.join(tmp_path, tag + ".wav")
        with wave.open(p, "wb") as f:
            f.setnchannels(ch); f.setsampwidth(width); f.setframerate(rate); f.writeframes(b"\x00" * 64)
        with pytest.raises(pkg.binding.WhisperError, match="16 kHz mono 16-bit"):
            pkg.binding.Wav(p)
    with pytest.raises(pkg.binding.WhisperError, match="cannot open"):
        pkg.binding.Wav(os.path.join(tmp_path, "missing.wav"))


def test_bench_refuses_a_launcher_whose_world_size_differs_from_gpus():
    """bench.py --gpus N must never print a line whose n_gpus is not N: under a launcher that started another number of
    ranks it stops before touching a GPU (this runs on the CPU box)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "--gpus 2" in r.stderr, (r.returncode, r.stderr[-500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
