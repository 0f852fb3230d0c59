import ctypes, sys
sys.path.insert(0, '.')
import openai_whisper_coreml_amd as pkg
c = pkg.binding.Context(debug=True); lib = c.lib
a, b = ctypes.c_float(), ctypes.c_float()
for grid, us in ((80, 5), (80, 20), (320, 5), (1024, 5)):
    st = lib.wmdbg_bench_graph_branches(c.handle, 100, grid, us, ctypes.byref(a), ctypes.byref(b))
    print("grid=%d spin=%dus x100: one chain %.1f us (%.2f/kernel), two chains %.1f us  ratio %.2f  st=%d" % (grid, us, a.value, a.value/100, b.value, b.value/a.value, st))
