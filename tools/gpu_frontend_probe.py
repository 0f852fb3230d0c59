"""Throw-away GPU probe: front-end timing via the in-library HIP-event profiler."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import openai_whisper_coreml_amd as pkg
from oracle import logmel_np as L
fe = pkg.binding.Context()
B = 8
x = np.stack([L.synth_chunk(i) for i in range(B)])
s16 = np.round(x * 32767).astype(np.int16)
d_in = fe.to_device(s16)
d_out = fe.dev_malloc(B * 80 * 3000 * 4)
for it in range(3):
    fe.lib.wm_logmel(fe.handle, d_in, 0, B, 80, d_out, 1, 1)
fe.sync()
fe.profile_reset(); fe.profile_enable(True)
t = time.time()
for it in range(20):
    fe.lib.wm_logmel(fe.handle, d_in, 0, B, 80, d_out, 1, 1)
fe.sync(); dt = time.time() - t
print("f32 path B=8 i16->f32: wall %.3f ms/call" % (dt / 20 * 1e3), fe.profile())
xd = x.astype(np.float64)
d_in2 = fe.to_device(xd); d_out2 = fe.dev_malloc(B * 80 * 3000 * 8)
fe.profile_reset()
for it in range(5):
    fe.lib.wm_logmel(fe.handle, d_in2, 2, B, 80, d_out2, 2, 1)
fe.sync(); print("f64 path:", fe.profile())
