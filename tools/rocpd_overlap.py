#!/usr/bin/env python3
"""Kernel timeline of a run in which several streams overlap (bench.py --views 8): for a window in the middle of the run print
every kernel with start offset, duration, the queue it ran on and how many kernels of OTHER queues ran during it; then, per
kernel name, its mean duration in the window.  Usage: tools/rocpd_overlap.py results.db [window_us=5000]"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("gsr::", "")[-48:]


def main(path, window_us=5000.0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((q for q in ("queue_id", "stream_id", "queue", "stream") if q in cols), None)
    rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    print("columns:", cols, file=sys.stderr)
    mid = rows[len(rows) * 2 // 3][1]
    # start the window at a preprocess_kernel launch
    i0 = next(i for i, r in enumerate(rows) if r[1] >= mid and "preprocess_kernel" in r[0] and "backward" not in r[0])
    t0 = rows[i0][1]
    win = [r for r in rows[i0:] if r[1] - t0 < window_us * 1e3]
    queues = sorted({r[3] for r in win})
    print(f"window {window_us:.0f} us from the run's 2/3 point, {len(win)} kernels, queues {queues}\n")
    print("| t_start us | dur us | queue | overlapping (other queues) | kernel |\n|---:|---:|---:|---|---|")
    for name, s, e, q in win:
        others = [short(n2) for n2, s2, e2, q2 in win if q2 != q and s2 < e and e2 > s]
        print(f"| {(s - t0)/1e3:.1f} | {(e - s)/1e3:.1f} | {queues.index(q)} | {', '.join(others[:4])}{' ...' if len(others) > 4 else ''} | `{short(name)}` |")
    span = max(r[2] for r in win) - t0
    n_views = sum(1 for r in win if "preprocess_kernel" in r[0] and "backward" not in r[0])
    print(f"\nspan {span/1e3:.1f} us, {n_views} views started: {span/1e3/max(n_views,1):.1f} us per view")
    agg = {}
    for name, s, e, q in win:
        agg.setdefault(short(name), []).append((e - s) / 1e3)
    print("\n| kernel | n | mean us in the window |\n|---|---:|---:|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{k}` | {len(v)} | {sum(v)/len(v):.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5000.0)
