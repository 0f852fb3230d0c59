#!/usr/bin/env python3
"""Record a FETCH_SIZE PMC pass in profiles/pmc_traffic.json, keyed "<family>@<model>:B<group chunks>", together with the
sha256 of the kernel source it was taken on (bench.py reports `traffic: null` when that no longer matches the tree).

    python tools/pmc_traffic_update.py <pmc summary .txt of tools/rocprof_pmc_summary.py> <model> <group chunks> [source note]

Kernel name -> family: dec_xrows_attn_kernel<8,..> (rounds 1-4: dec_rows_attn_kernel<8,..>) = dec_attn_cross, dec_rows_attn_kernel<4,..> = dec_attn_self."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import family_source_sha256  # noqa: E402


def family_of(kernel):
    if "dec_xrows_attn_kernelILi8" in kernel or "dec_rows_attn_kernelILi8" in kernel:   # (round 5: the cross family's own kernel)
        return "dec_attn_cross"
    if "dec_rows_attn_kernelILi4" in kernel:
        return "dec_attn_self"
    return None


def main():
    path, model, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
    note = sys.argv[4] if len(sys.argv) > 4 else os.path.relpath(path, ROOT)
    db_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    db = json.load(open(db_path))
    for line in open(path):
        parts = line.split()
        if len(parts) < 5 or not re.match(r"^\d+$", parts[1]):
            continue
        fam = family_of(parts[0])
        if fam is None:
            continue
        f, sha = family_source_sha256(fam)
        db["%s@%s:B%d" % (fam, model, B)] = {
            "hbm_read_bytes_per_launch": float(parts[4]),
            "source": "%s (2 x FETCH_SIZE x 1024, MI355X_MICROARCH.md HBM section)" % note,
            "kernel_source": "csrc/" + f, "kernel_source_sha256": sha}
        print(fam, parts[4])
    json.dump(db, open(db_path, "w"), indent=1)


if __name__ == "__main__":
    main()
