#!/usr/bin/env python3
"""Per-lane view of a rocprofv3 --kernel-trace of the multi-lane decode (rocpd sqlite .db): finds the window in
which every lane's stream runs decode kernels, and reports per kernel family the mean duration inside that window,
the busy fraction of each lane, and the HBM bytes the window moved (algorithmic, large-v2) per wall second.
Usage: python tools/rocprof_lanes.py <results.db> [chunks_per_group] [phase_index]"""
import re
import sqlite3
import sys

import numpy as np


def short(n):
    m = re.search(r"(dec_gemv_kernelILi\d+ELi\dELi\dELi\dELb\d|dec_x?rows_attn_kernelILi\dELi\dELb\d|dec_xattn_fq_kernel|gemm256_bf16_kernelILi\d|"
                  r"gemm_bf16_kernelILi\d|enc_attn_kernel|layernorm_kernel|argmax_embed_kernel|dec_\w+?_kernel|logmel_stage\d)", n)
    return m.group(1) if m else n[:40]


def main():
    c = sqlite3.connect(sys.argv[1])
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 56
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute(f"select s.kernel_name, d.stream_id, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    streams = sorted({r[1] for r in rows})
    # decode window: from the last stream's first cross-attention launch of its LAST decode phase to the first stream's last one
    dec = [(short(n), s, a, b) for n, s, a, b in rows]
    is_x = lambda n: n.startswith(("dec_xrows_attn_kernelILi8", "dec_rows_attn_kernelILi8", "dec_xattn_fq_kernel"))   # (rounds 1-4: dec_rows_...)
    per = {s: [(a, b) for n, ss, a, b in dec if ss == s and is_x(n)] for s in streams}
    per = {s: v for s, v in per.items() if len(v) > 100}
    # split each stream's cross-attention launches into phases (gap > 5 ms = an encoder in between)
    phases = {}
    for s, v in per.items():
        cuts = [0] + [i + 1 for i in range(len(v) - 1) if v[i + 1][0] - v[i][1] > 5e6] + [len(v)]
        phases[s] = [(v[cuts[i]][0], v[cuts[i + 1] - 1][1], cuts[i + 1] - cuts[i]) for i in range(len(cuts) - 1)]
        print("stream", s, "decode phases:", [(round((b - a) / 1e6, 1), n) for a, b, n in phases[s]])
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # which decode phase of every stream (0 = warm-up run, 1 = timed run)
    big = {s: p[k] for s, p in phases.items() if len(p) > k}
    w0 = max(a for a, b, n in big.values())
    w1 = min(b for a, b, n in big.values())
    print("window with all %d lanes decoding: %.1f ms" % (len(big), (w1 - w0) / 1e6))
    fam = {}
    busy = {s: 0.0 for s in big}
    for n, s, a, b in dec:
        if a >= w0 and b <= w1 and s in big:
            fam.setdefault(n, []).append((b - a) / 1e3)
            busy[s] += (b - a)
    tot = sum(sum(v) for v in fam.values())
    print("%-44s %8s %10s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "%"))
    for n, v in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
        print("%-44s %8d %10.2f %9.2f %9.2f %6.2f" % (n, len(v), sum(v) / 1e3, np.mean(v), min(v), 100 * sum(v) / tot))
    for s in big:
        print("lane stream %d busy %.1f%% of the window" % (s, 100 * busy[s] / (w1 - w0)))
    # how many cross-attention launches (the HBM-saturating kernel) are in flight over the window
    ev = []
    for n, s, a, b in dec:
        if a >= w0 and b <= w1 and s in big and is_x(n):
            ev += [(a, 1), (b, -1)]
    ev.sort()
    hist, cur, last = {}, 0, w0
    for t, dlt in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur, last = cur + dlt, t
    hist[cur] = hist.get(cur, 0) + (w1 - last)
    print("cross-attention launches in flight: " + ", ".join("%d: %.1f%%" % (k, 100 * v / (w1 - w0)) for k, v in sorted(hist.items())))
    # how far apart the lanes are in the layer sequence: cumulative cross-attention launches (one per layer-step) of every
    # lane at the start of each launch of the first lane
    import bisect
    starts = {s: [a for n, ss, a, b in dec if ss == s and is_x(n) and w0 <= a <= w1] for s in big}
    ref = sorted(big)[0]
    lags = []
    for i, t in enumerate(starts[ref]):
        for s in big:
            if s != ref:
                lags.append(bisect.bisect_left(starts[s], t) - i)
    if lags:
        lags = np.array(lags)
        print("lane lag vs first lane in layer-steps: mean %.1f, |lag| median %.1f, p90 %.1f, max %d" % (
            lags.mean(), np.median(np.abs(lags)), np.percentile(np.abs(lags), 90), np.abs(lags).max()))
    nx = sum(len(v) for n, v in fam.items() if is_x(n))
    d, L, V = 1280, 32, 51865
    layers = nx                                     # one cross-attention launch per (lane, layer, position)
    bytes_ = layers * (14 * d * d * 2 + B * 2 * 1500 * d * 2 + B * 2 * 20 * d * 2)   # self-KV at ~20 rows (short run)
    print("layer-steps in window %d -> %.1f GB algorithmic, %.2f TB/s aggregate" % (layers, bytes_ / 1e9, bytes_ / ((w1 - w0) * 1e-9) / 1e12))


if __name__ == "__main__":
    main()
