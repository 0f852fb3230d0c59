"""GPU probe (measurement reference only, NOT product code): what the vendor library GEMM (torch.matmul -> hipBLASLt / rocBLAS)
reaches on this box for the encoder's shapes with the same kind of operands (A ~ N(0,1), W ~ N(0, 0.02^2), bf16, f32
accumulate) -- the practical ceiling of a bf16 GEMM at these sizes on real data, next to the 2.5 PFLOP/s datasheet peak
the roofline fractions are quoted against."""
import sys
import time

import torch

shapes = [(8192, 8192, 8192), (84000, 5120, 1280), (84000, 1280, 5120), (84000, 3840, 1280), (84000, 1280, 1280),
          (84000, 2560, 1280), (12000, 5120, 1280)]
dev = torch.device("cuda")
for zeros in (False, True):
    for M, N, K in shapes:
        a = (torch.zeros if zeros else torch.randn)(M, K, device=dev, dtype=torch.float32).to(torch.bfloat16)
        w = ((torch.zeros if zeros else torch.randn)(N, K, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        for _ in range(3):
            c = a @ w.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 16
        e0.record()
        for _ in range(iters):
            c = a @ w.t()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print("%s %6d x %5d x %5d  library %8.1f us %7.0f TF/s" % ("zeros " if zeros else "random", M, N, K, us,
                                                                   2.0 * M * N * K / us / 1e6), flush=True)
        del a, w, c
