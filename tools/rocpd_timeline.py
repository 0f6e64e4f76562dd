#!/usr/bin/env python3
"""Print the kernel timeline of the LAST train iteration found in a rocprofv3 rocpd database: start offset, duration
and the idle gap before each kernel.  An iteration is delimited by `preprocess_kernel` launches.
Usage: tools/rocpd_timeline.py results.db [iterations_from_end | -index_from_start]"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "")[-56:]


def main(path, back=2):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[0] and "backward" not in r[0]]
    if back < 0:  # negative: absolute iteration index from the start
        lo, hi = starts[-back], starts[-back + 1]
    else:
        lo = starts[-back]
        hi = starts[-back + 1] if back > 1 else len(rows)
    t0 = rows[lo][1]
    prev_end = t0
    print("| t_start us | dur us | gap us | kernel |\n|---:|---:|---:|---|")
    busy = 0
    for name, s, e in rows[lo:hi]:
        print(f"| {(s - t0)/1e3:.1f} | {(e - s)/1e3:.1f} | {(s - prev_end)/1e3:.1f} | `{short(name)}` |")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"\nspan {(prev_end - t0)/1e3:.1f} us, kernels busy {busy/1e3:.1f} us, idle {(prev_end - t0 - busy)/1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
