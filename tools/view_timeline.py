#!/usr/bin/env python3
"""Development tool: from a rocprofv3 kernel trace (results.db) of tools/views8_probe.py, the kernels of the last few views in
start order with their hardware queue -- which kernels really ran side by side in a pipelined batch.
    tools/view_timeline.py results.db [views=3]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
nviews = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
k1 = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[0] and "backward" not in r[0]]
first = k1[-(nviews + 2)]
last = k1[-2]
t0 = rows[first][1]
short = lambda n: n.replace("gsr::", "").split("<")[0].split("(")[0][:34]  # noqa: E731
lanes = sorted({r[3] for r in rows[first:last]})
print("t_start us | dur us | queue | kernel   (one column per hardware queue)")
for n, s, e, q in rows[first:last]:
    col = lanes.index(q)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  " + " " * (36 * col) + short(n))
