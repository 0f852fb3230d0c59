"""GPU probe: the cross-attention kernel alone and 2 / 3 concurrent (launch-shape variants: wmdbg_set_tuning keys
xattn_wgs / xattn_lds_pad / xattn_split_below through `bench.py --tuning`; the product reads no environment variable)."""
import ctypes, sys, threading
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
ctxs = [pkg.binding.Context(debug=True) for _ in range(3)]
lib = ctxs[0].lib
def attn(c, B, iters=300):
    us = ctypes.c_float()
    assert lib.wmdbg_bench_dec_attention(c.handle, B, 20, 1500, 1500, 1, 8, iters, ctypes.byref(us)) == 0
    return us.value
def conc(S, B):
    res = [None] * S
    th = [threading.Thread(target=lambda i=i: res.__setitem__(i, attn(ctxs[i], B))) for i in range(S)]
    [t.start() for t in th]; [t.join() for t in th]
    return sum(res) / S
for B in (8, 32, 64):
    line = "B=%2d" % B
    for S in (1, 2, 3):
        conc(S, B)
        us = conc(S, B)
        line += "   S=%d %7.2f us %5.0f GB/s" % (S, us, S * B * 20 * 1500 * 256 / us / 1e3)
    print(line)
