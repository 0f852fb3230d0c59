"""GPU probe: does a weight GEMV chain overlap with a cross-attention chain running on another stream?
Per-launch times of each chain alone and together (B = 16, large-v2 shapes)."""
import ctypes, sys, threading, time
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
ctxs = [pkg.binding.Context(debug=True) for _ in range(4)]
lib = ctxs[0].lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16

def gemv(c, N, K, ln, resid, iters):
    us = ctypes.c_float()
    assert lib.wmdbg_bench_dec_gemv(c.handle, B, N, K, ln, resid, 32, iters, 0, ctypes.byref(us)) == 0
    return us.value

def attn(c, iters):
    us = ctypes.c_float()
    assert lib.wmdbg_bench_dec_attention(c.handle, B, 20, 1500, 1500, 1, 8, iters, ctypes.byref(us)) == 0
    return us.value

def together(jobs):
    res = [None] * len(jobs)
    def work(i):
        res[i] = jobs[i](ctxs[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in th]; [t.join() for t in th]
    return res

A = lambda c: attn(c, 400)
G1 = lambda c: gemv(c, 5120, 1280, 1, 0, 1000)
G2 = lambda c: gemv(c, 1280, 5120, 0, 1, 1000)
for name, jobs in [("attn alone", [A]), ("fc1 alone", [G1]), ("fc2 alone", [G2]), ("attn + fc1", [A, G1]), ("attn + fc2", [A, G2]),
                   ("attn + fc1 + fc2", [A, G1, G2]), ("attn + attn", [A, A]), ("attn + attn + fc1", [A, A, G1]), ("fc1 + fc2", [G1, G2])]:
    together(jobs)
    r = together(jobs)
    print("%-20s" % name, "  ".join("%7.2f us" % v for v in r))
