#!/usr/bin/env python3
"""Per-stage hardware counters of a train iteration from rocprofv3 PMC passes -> profiles/traffic_latest.json.

    tools/collect_counters.py --key KEY [--fetch fetch.db] [--write write.db] [--sq sq.db] [--note TEXT] [--md out.md]

Each database is one `rocprofv3 --kernel-trace --pmc <counters>` pass (rocpd format) over `bench.py --train-only ...`, so
every launch belongs to a train iteration: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit the TCC's four
slots together), the SQ instruction counters in a third.  Launches are assigned to the bench line's stages by POSITION in the
iteration (`sort_scan_kernel` runs in two stages): preprocess_kernel opens `preprocess`, the emit kernel `bin`,
blend_forward_kernel `blend_forward`, backward_worklist_kernel `blend_backward`, preprocess_backward_kernel
`preprocess_backward`; whatever follows until the next preprocess_kernel (torch's own kernels) is dropped.  The first
iteration (warm-up) is dropped.

gfx950 corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports half the bytes of
wide coalesced streaming reads, so it is DOUBLED for the two streaming kernels (preprocess_kernel, preprocess_backward_kernel)
and taken as is for the gather / list kernels (their reads are 4- to 16-byte gathers; uncalibrated, as the guide says).

The file is stamped with the hash of the kernel sources (bench.csrc_sha16); bench.py quotes it only while that matches."""
import argparse
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import STAGES, csrc_sha16  # noqa: E402

STREAMING = ("preprocess_kernel", "preprocess_backward_kernel")


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("gsr::", "")


def dispatches(path):
    """[(dispatch_id, kernel name, {counter: value})] in launch order."""
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    per, kname = defaultdict(lambda: defaultdict(float)), {}
    for did, kn, cn, v in c.execute(f"select dispatch_id, {namecol}, counter_name, value from counters_collection"):
        per[did][cn] += v
        kname[did] = short(kn)
    return [(d, kname[d], dict(per[d])) for d in sorted(per)]


def by_stage(rows):
    """-> ({stage: {counter: mean per iteration}}, {kernel: {counter: mean per launch}}, iterations used)"""
    its, cur, stage = [], None, None
    kern = defaultdict(lambda: defaultdict(list))
    for _, k, cs in rows:
        base = k.split("<")[0]
        if base == "preprocess_kernel":
            cur = defaultdict(lambda: defaultdict(float))
            its.append(cur)
            stage = "preprocess"
        elif base in ("emit_groups_kernel", "emit_keys_kernel"):
            stage = "bin"
        elif base == "blend_forward_kernel":
            stage = "blend_forward"
        elif base == "backward_worklist_kernel":
            stage = "blend_backward"
        elif base == "preprocess_backward_kernel":
            stage = "preprocess_backward"
        elif stage == "preprocess_backward":
            stage = None  # behind the iteration's last kernel
        if cur is None or stage is None:
            continue
        for cn, v in cs.items():
            cur[stage][cn] += v
            if len(its) > 1:
                kern[k][cn].append(v)
        if base == "preprocess_backward_kernel":
            stage = "preprocess_backward"
    its = [i for i in its[1:] if "preprocess_backward" in i]  # whole train iterations, warm-up dropped
    out = {}
    for st in STAGES:
        cs = defaultdict(list)
        for i in its:
            for cn, v in i.get(st, {}).items():
                cs[cn].append(v)
        out[st] = {cn: sum(v) / len(v) for cn, v in cs.items()}
    return out, {k: {cn: sum(v) / len(v) for cn, v in d.items()} for k, d in kern.items()}, len(its)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--note", default="")
    ap.add_argument("--md")
    a = ap.parse_args()
    entry = {"per_launch_bytes": {}, "valu_wave_insts": {}, "fetch_bytes": {}, "write_bytes": {}}
    kernels = defaultdict(dict)
    n_it = {}
    if a.fetch:
        # per KERNEL first (the x2 applies to two kernels, not to their stage), then summed by stage through the same walk
        rows = dispatches(a.fetch)
        rows = [(d, k, {"B": (2.0 if k.split("<")[0] in STREAMING else 1.0) * cs.get("FETCH_SIZE", 0.0) * 1024.0}) for d, k, cs in rows]
        st, kn, n_it["fetch"] = by_stage(rows)
        entry["fetch_bytes"] = {s: int(v.get("B", 0)) for s, v in st.items()}
        for k, v in kn.items():
            kernels[k]["fetch_bytes"] = int(v["B"])
    if a.write:
        rows = [(d, k, {"B": cs.get("WRITE_SIZE", 0.0) * 1024.0}) for d, k, cs in dispatches(a.write)]
        st, kn, n_it["write"] = by_stage(rows)
        entry["write_bytes"] = {s: int(v.get("B", 0)) for s, v in st.items()}
        for k, v in kn.items():
            kernels[k]["write_bytes"] = int(v["B"])
    if a.fetch and a.write:
        entry["per_launch_bytes"] = {s: entry["fetch_bytes"][s] + entry["write_bytes"][s] for s in STAGES}
    if a.sq:
        st, kn, n_it["sq"] = by_stage(dispatches(a.sq))
        entry["valu_wave_insts"] = {s: int(v.get("SQ_INSTS_VALU", 0)) for s, v in st.items()}
        entry["salu_wave_insts"] = {s: int(v.get("SQ_INSTS_SALU", 0)) for s, v in st.items()}
        for k, v in kn.items():
            kernels[k]["valu_wave_insts"] = int(v.get("SQ_INSTS_VALU", 0))
    entry["per_kernel"] = {k: kernels[k] for k in sorted(kernels)}
    entry["iterations_averaged"] = n_it
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    sha = csrc_sha16()
    try:
        doc = json.load(open(path))
        if doc.get("csrc_sha16") != sha or "workloads" not in doc:
            doc = None
    except Exception:
        doc = None
    if doc is None:
        doc = {"csrc_sha16": sha, "source": "", "workloads": {}}
    if a.note:
        doc["source"] = a.note
    doc["workloads"][a.key] = entry
    json.dump(doc, open(path, "w"), indent=1)
    lines = [f"## {a.key}  (train iterations averaged: {n_it})", "",
             "| stage | FETCH MB (x2 on the streaming kernels) | WRITE MB | total MB | VALU wave-instructions |", "|---|---:|---:|---:|---:|"]
    for s in STAGES:
        lines.append(f"| {s} | {entry['fetch_bytes'].get(s, 0) / 1e6:.1f} | {entry['write_bytes'].get(s, 0) / 1e6:.1f} | "
                     f"{entry['per_launch_bytes'].get(s, 0) / 1e6:.1f} | {entry['valu_wave_insts'].get(s, 0):.3e} |")
    lines += ["", "| kernel (per launch) | FETCH MB | WRITE MB | VALU wave-instructions |", "|---|---:|---:|---:|"]
    for k, v in sorted(kernels.items(), key=lambda kv: -(kv[1].get("fetch_bytes", 0) + kv[1].get("write_bytes", 0))):
        lines.append(f"| `{k[:70]}` | {v.get('fetch_bytes', 0) / 1e6:.2f} | {v.get('write_bytes', 0) / 1e6:.2f} | {v.get('valu_wave_insts', 0):.3e} |")
    text = "\n".join(lines) + "\n"
    if a.md:
        open(a.md, "a").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
