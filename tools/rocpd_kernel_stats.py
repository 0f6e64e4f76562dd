#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`--kernel-trace --stats` output on ROCm 7.2)
as the per-kernel table `--stats` used to print: calls, total / average / min / max duration.
Usage: tools/rocpd_kernel_stats.py results.db [> profiles/rNN_kernel_stats.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[-70:]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    dur = "duration" if "duration" in cols else "(end - start)"
    rows = c.execute(f"select name, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels "
                     "group by name order by sum(" + dur + ") desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{short(name)}` | {n} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
