#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 rocpd database (--pmc ... --kernel-trace).
Usage: tools/rocpd_pmc.py results.db [kernel-name-substring]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=""):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    rows = c.execute(f"select dispatch_id, {namecol}, counter_name, value from counters_collection").fetchall()
    per = defaultdict(lambda: defaultdict(float))
    kname = {}
    for did, kn, cn, v in rows:
        per[did][cn] += v
        kname[did] = kn
    agg = defaultdict(lambda: defaultdict(list))
    for did, d in per.items():
        k = re.sub(r"\(.*$", "", kname[did]).replace("void ", "")[-60:]
        if filt and filt not in k:
            continue
        for cn, v in d.items():
            agg[k][cn].append(v)
    for k, d in sorted(agg.items()):
        print(f"{k}  (dispatches {len(next(iter(d.values())))})")
        for cn, vs in sorted(d.items()):
            print(f"    {cn:28s} {sum(vs) / len(vs):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
