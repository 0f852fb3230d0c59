import ctypes, sys
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
c = pkg.binding.Context(debug=True); lib = c.lib
us = ctypes.c_float()
for B in (16, 32, 64):
    for (name, N, K, ln, resid) in [("fc2", 1280, 5120, 0, 1), ("attn_out", 1280, 1280, 0, 1)]:
        line = "B=%d %-8s" % (B, name)
        for nw in (0, 4, 8, 16):
            st = lib.wmdbg_bench_dec_gemv(c.handle, B, N, K, ln, resid, 32, 320, nw, ctypes.byref(us))
            line += "  nw=%d %6.2f us" % (nw, us.value) if st == 0 else "  nw=%d ERR" % nw
        print(line)
