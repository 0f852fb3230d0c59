"""GPU probe: aggregate streaming rate of the decode kernels when S independent streams run them concurrently
(what the lanes of wm_transcribe_greedy do).  Each stream replays a chain of launches cycling over 32 matrices."""
import ctypes, sys, threading, time
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
lib = None
ctxs = [pkg.binding.Context(debug=True) for _ in range(6)]
lib = ctxs[0].lib

def run_conc(S, fn):
    res = [None] * S
    def work(i):
        res[i] = fn(ctxs[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    t0 = time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    return res, time.perf_counter() - t0

def gemv(N, K, ln, resid, iters=640, B=8):
    def f(c):
        us = ctypes.c_float()
        st = lib.wmdbg_bench_dec_gemv(c.handle, B, N, K, ln, resid, 32, iters, 0, ctypes.byref(us))
        assert st == 0
        return us.value
    return f

def attn(B, nk, ns, iters=320):
    def f(c):
        us = ctypes.c_float()
        st = lib.wmdbg_bench_dec_attention(c.handle, B, 20, nk if nk > 448 else 448, nk, ns, 8, iters, ctypes.byref(us))
        assert st == 0
        return us.value
    return f

cases = [("ln_fc1 5120x1280 B=8", gemv(5120, 1280, 1, 0), 5120 * 1280 * 2), ("ln_fc1 5120x1280 B=16", gemv(5120, 1280, 1, 0, B=16), 5120 * 1280 * 2),
         ("fc2 1280x5120 B=8", gemv(1280, 5120, 0, 1), 1280 * 5120 * 2), ("fc2 1280x5120 B=16", gemv(1280, 5120, 0, 1, B=16), 1280 * 5120 * 2),
         ("cross attn B=8 ns=1", attn(8, 1500, 1), 8 * 20 * 1500 * 64 * 4), ("cross attn B=12 ns=1", attn(12, 1500, 1), 12 * 20 * 1500 * 64 * 4),
         ("cross attn B=16 ns=1", attn(16, 1500, 1), 16 * 20 * 1500 * 64 * 4), ("cross attn B=16 ns=2", attn(16, 1500, 2), 16 * 20 * 1500 * 64 * 4),
         ("self attn B=8 k=224", attn(8, 224, 1), 8 * 20 * 224 * 64 * 4), ("self attn B=16 k=224", attn(16, 224, 1), 16 * 20 * 224 * 64 * 4)]
for name, fn, nbytes in cases:
    line = "%-20s" % name
    for S in (1, 2, 3, 4):
        run_conc(S, fn)  # warm
        res, wall = run_conc(S, fn)
        per = sum(res) / len(res)
        line += "  S=%d %6.2f us/launch %5.0f GB/s |" % (S, per, S * nbytes / per / 1e3)
    print(line)
