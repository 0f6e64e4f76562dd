#!/usr/bin/env python3
"""Look at seeds tools/fuzz_parity.py reported: every gradient tensor's error against the oracle, and the row that carries it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tp  # noqa: E402
from helpers import make_case, oracle_backward, oracle_forward, seed_gradient  # noqa: E402
from oracle import cpu  # noqa: E402

cpu.build()
for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(1000 + seed)
    P = int([1, 2, 63, 65, 255, 256][seed] if seed < 6 else rng.integers(300, 6000))
    W = int(rng.choice([1, 2, 15, 17, 31]) if seed % 3 == 0 else rng.integers(1, 400))
    H = int(rng.choice([1, 3, 16, 47]) if seed % 4 == 1 else rng.integers(1, 300))
    D = int(rng.integers(0, 4))
    case = make_case(P, W, H, seed=100 + seed, s0=float(rng.choice([0.01, 0.05, 0.3])), view=int(rng.integers(0, 4)),
                     sh_degree=D, scale_xyz=float(rng.choice([0.2, 1.0, 2.5])))
    sm = float(rng.choice([0.5, 1.0, 1.7]))
    f = oracle_forward(cpu, case, scale_modifier=sm)
    G = seed_gradient(H, W, seed) * (H * W)
    g = oracle_backward(cpu, case, f, G, scale_modifier=sm)
    print(f"seed {seed}: P={P} {W}x{H} D={D} sm={sm} R={f['num_rendered']}")
    for rep in range(2):
        h = tp._grads_hip(case, G, scale_modifier=sm)
        for k, v in h.items():
            ref = g[k].reshape(v.shape).astype(np.float64)
            d = np.abs(v.astype(np.float64) - ref)
            i = np.unravel_index(np.argmax(d), d.shape)
            row = i[0]
            print(f"  run {rep} {k:14s} max|ref| {np.abs(ref).max():.4e}  max err {d.max():.3e} = {d.max() / max(np.abs(ref).max(), 1e-30):.2e} of max;"
                  f" at row {row}: ref {ref[i]:.6e} got {v[i]:.6e}; row radius {f['radii'][row]}, |row ref|max {np.abs(ref[row]).max():.3e}")
