#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc <COUNTER> --kernel-trace run (rocpd sqlite .db): per kernel, the mean / max of the
counter per dispatch.  For FETCH_SIZE (KiB per dispatch) also prints 2 x value x 1024 = HBM read bytes per launch
(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read).
Usage: python tools/rocprof_pmc_summary.py <results.db> [counter_name]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    cols = lambda t: [r[1] for r in c.execute("pragma table_info(%s)" % t)]
    pe_c, pi_c = cols(pe), cols(pi)
    name_col = "name" if "name" in pi_c else [x for x in pi_c if "name" in x][0]
    ev_col = "event_id" if "event_id" in pe_c else [x for x in pe_c if x.endswith("event_id")][0]
    kd_c = cols(kd)
    kd_ev = "event_id" if "event_id" in kd_c else "id"
    q = (f"select s.kernel_name, d.id, sum(p.value) from {pe} p join {pi} i on p.pmc_id = i.id "
         f"join {kd} d on p.{ev_col} = d.{kd_ev} join {ks} s on d.kernel_id = s.id where i.{name_col} = ? "
         f"group by s.kernel_name, d.id")
    by = {}
    for n, _, v in c.execute(q, (want,)):
        by.setdefault(n, []).append(float(v))
    print("# %s per dispatch (summed over dimensions)" % want)
    print("%-84s %7s %14s %14s %20s" % ("kernel", "calls", "avg", "max", "2x_bytes_per_launch" if want == "FETCH_SIZE" else ""))
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
        avg = sum(v) / len(v)
        extra = "%20.0f" % (2 * avg * 1024) if want == "FETCH_SIZE" else ""
        print("%-84s %7d %14.1f %14.1f %s" % (n[:84], len(v), avg, max(v), extra))


if __name__ == "__main__":
    main()
