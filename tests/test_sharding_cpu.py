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
