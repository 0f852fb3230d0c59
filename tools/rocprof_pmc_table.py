#!/usr/bin/env python3
"""Merge several rocprofv3 --pmc passes (rocpd sqlite .db files, one counter set each) into ONE table per kernel:
mean of every counter per dispatch (summed over dimensions), for the kernels whose name contains one of the filters.

    python tools/rocprof_pmc_table.py --filter gemm256 --filter enc_attn pass1.db pass2.db ...
"""
import sqlite3
import sys


def read(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    cols = lambda t: [r[1] for r in c.execute("pragma table_info(%s)" % t)]
    pe_c, pi_c, kd_c = cols(pe), cols(pi), cols(kd)
    name_col = "name" if "name" in pi_c else [x for x in pi_c if "name" in x][0]
    ev_col = "event_id" if "event_id" in pe_c else [x for x in pe_c if x.endswith("event_id")][0]
    kd_ev = "event_id" if "event_id" in kd_c else "id"
    q = (f"select s.kernel_name, i.{name_col}, d.id, sum(p.value) from {pe} p join {pi} i on p.pmc_id = i.id "
         f"join {kd} d on p.{ev_col} = d.{kd_ev} join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.{name_col}, d.id")
    out = {}
    for kn, cn, _, v in c.execute(q):
        out.setdefault(kn, {}).setdefault(cn, []).append(float(v))
    return out


def main():
    filters, dbs = [], []
    a = sys.argv[1:]
    while a:
        x = a.pop(0)
        if x == "--filter":
            filters.append(a.pop(0))
        else:
            dbs.append(x)
    merged = {}
    for db in dbs:
        for kn, cs in read(db).items():
            if filters and not any(f in kn for f in filters):
                continue
            for cn, v in cs.items():
                # the LARGEST dispatches of the kernel (the full-size launches: a kernel family also runs small shapes)
                v = sorted(v)
                top = v[len(v) // 2:]
                merged.setdefault(kn, {})[cn] = (sum(top) / len(top), len(v))
    for kn, cs in sorted(merged.items()):
        print("## %s" % kn[:110])
        wc = cs.get("SQ_WAVE_CYCLES", (0, 0))[0]
        for cn, (avg, n) in sorted(cs.items()):
            frac = "  %6.3f of SQ_WAVE_CYCLES" % (avg / wc) if wc and cn.startswith("SQ_") and cn != "SQ_WAVE_CYCLES" else ""
            print("   %-34s %16.0f  (upper half of %4d dispatches)%s" % (cn, avg, n, frac))


if __name__ == "__main__":
    main()
