#!/usr/bin/env python3
"""Development tool: the gradient errors (vs the oracle) of a few tools/fuzz_v2.py configurations, several runs each -- how much
of an excess over the bar is run-to-run (atomic order) and how much is systematic.   python tools/fuzz_v2_repeat.py 256 365 --runs 4"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("seeds", type=int, nargs="+")
ap.add_argument("--runs", type=int, default=4)
a = ap.parse_args()
import test_gpu_parity as tp  # noqa: E402
from gaussianeditor_amd.synth import seed_gradient  # noqa: E402
from helpers import oracle_backward, oracle_forward, v2_fuzz_case  # noqa: E402
from oracle import cpu  # noqa: E402

cpu.build()
for seed in a.seeds:
    case, sm, D = v2_fuzz_case(seed)
    P, W, H = case["sc"]["xyz"].shape[0], case["W"], case["H"]
    f = oracle_forward(cpu, case, scale_modifier=sm)
    G = seed_gradient(H, W, seed) * (H * W)
    g = oracle_backward(cpu, case, f, G, scale_modifier=sm)
    rows = []
    for _ in range(a.runs):
        h = tp._grads_hip(case, G, scale_modifier=sm)
        rows.append({k: tp.rel_err(v, g[k].reshape(v.shape)) for k, v in h.items()})
    print(f"seed {seed} P={P} {W}x{H} D={D} sm={sm} R={f['num_rendered']}")
    for k in rows[0]:
        print(f"   {k:14s} " + " ".join(f"{r[k]:.2e}" for r in rows))
