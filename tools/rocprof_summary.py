#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or kernel_trace.csv): per-kernel
count / total / avg / min / max in microseconds, plus inter-kernel gap statistics.
Usage: python tools/rocprof_summary.py <results.db | kernel_trace.csv> [top_n]"""
import csv
import sqlite3
import sys

import numpy as np


def load(path):
    rows = []
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        q = f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"
        rows = [(n, int(a), int(b)) for n, a, b in c.execute(q)]
    else:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return rows


def main():
    rows = load(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    by = {}
    for n, a, b in rows:
        by.setdefault(n, []).append((b - a) / 1e3)
    tot = sum(sum(v) for v in by.values())
    print("%-88s %8s %12s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print("%-88s %8d %12.1f %9.2f %9.2f %9.2f %6.2f" % (n[:88], len(v), sum(v), np.mean(v), min(v), max(v),
                                                            100 * sum(v) / tot))
    se = np.array(sorted((a, b) for _, a, b in rows), dtype=np.int64)
    gaps = (se[1:, 0] - se[:-1, 1]) / 1e3
    g = gaps[(gaps > -1e3) & (gaps < 1e3)]
    print("kernels %d  busy %.1f us  span %.1f us  gap median %.2f mean %.2f p90 %.2f us" % (
        len(rows), tot, (se[-1, 1] - se[0, 0]) / 1e3, np.median(g), g.mean(), np.percentile(g, 90)))


if __name__ == "__main__":
    main()
