"""GPU probe: throughput with S independent contexts (each: own stream, own weights copy) driven by S host
threads, each running the full batch-8 pipeline -- i.e. software pipelining of whole batches."""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
import openai_whisper_coreml_amd as pkg
B = pkg.binding
dims = B.MODEL_DIMS["large-v2"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctxs = []
for s in range(S):
    c = B.Context(dims); c.init_synthetic(1); c.finalize(); ctxs.append(c)
rng = np.random.default_rng(0)
pcm = (rng.standard_normal((8, 480000)) * 3000).astype(np.int16)
dp = [c.to_device(pcm) for c in ctxs]
prompt = [50258, 50259, 50359, 50363]
def work(i, n):
    for _ in range(n):
        ctxs[i].transcribe_greedy(dp[i], prompt, 224, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=8)
for i in range(S): work(i, 1)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i, steps)) for i in range(S)]
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0
print("streams=%d: %d batches of 8 chunks in %.3f s -> %.1f audio-s/s" % (S, S * steps, dt, S * steps * 240 / dt))
