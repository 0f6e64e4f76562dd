#!/usr/bin/env python3
"""Development tool: per ROUND of tools/views8_probe.py in a rocprofv3 kernel trace (a round starts with the spin kernels of
multiview._concurrent_streams), the HARDWARE QUEUE of the two view streams (where the K7 launches ran), of the side stream
(the touched-rows plan) and of the launch stream (the accumulate kernel), and the time per view of the round's last steps.
    tools/queue_trace.py results.db"""
import sqlite3
import sys
from collections import Counter

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
segs, cur = [], []
in_sleep = False
for r in rows:
    is_sleep = "sleep" in r[0].lower() or "spin" in r[0].lower()
    if is_sleep and not in_sleep and cur:
        segs.append(cur)
        cur = []
    in_sleep = is_sleep
    if not is_sleep:
        cur.append(r)
if cur:
    segs.append(cur)
print(f"{len(rows)} kernels, {len(segs)} segments")
for si, g in enumerate(segs):
    k7 = [(s, e, q) for n, s, e, q in g if "blend_backward_kernel" in n]
    if len(k7) < 16:
        continue
    last = k7[-48:]
    t0, t1 = last[0][0], last[-1][1]
    per_view = (t1 - t0) / 1e3 / len(last)
    win = [(n, s, e, q) for n, s, e, q in g if t0 <= s <= t1]
    qv = Counter(q for _, _, q in last)
    qs = Counter(q for n, s, e, q in win if "touched_rows" in n or "compact_count" in n)
    qm = Counter(q for n, s, e, q in win if "accumulate" in n)
    qf = Counter(q for n, s, e, q in win if "preprocess_kernel" in n and "backward" not in n)
    print(f"segment {si}: {len(k7)} views; last {len(last)}: {per_view:6.1f} us/view = {1e6 / per_view:6.0f} view-it/s | view streams on queues "
          f"{dict(qv)} (K1: {dict(qf)}), side stream {dict(qs)}, launch stream {dict(qm)}")
