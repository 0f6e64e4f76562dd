#!/usr/bin/env python3
"""Development tool: for one tools/fuzz_v2.py configuration, the rows where the product's gradient is furthest from the oracle's,
with the float64 value and the Gaussian's screen-space record.   python tools/fuzz_v2_inspect.py 365 [--tensor dL_dmeans2D]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("seed", type=int)
ap.add_argument("--tensor", default="dL_dmeans2D")
ap.add_argument("--rows", type=int, default=4)
a = ap.parse_args()
import test_gpu_parity as tp  # noqa: E402
from gaussianeditor_amd.synth import seed_gradient  # noqa: E402
from helpers import oracle_backward, oracle_forward, v2_fuzz_case  # noqa: E402
from oracle import cpu  # noqa: E402
from oracle.torch_ref import render_f64  # noqa: E402

cpu.build()
seed = a.seed
case, sm, D = v2_fuzz_case(seed)
sc = case["sc"]
P, W, H = sc["xyz"].shape[0], case["W"], case["H"]
cam = case["cam"]
f = oracle_forward(cpu, case, scale_modifier=sm)
G = seed_gradient(H, W, seed) * (H * W)
g = oracle_backward(cpu, case, f, G, scale_modifier=sm)
h = tp._grads_hip(case, G, scale_modifier=sm)
d = torch.float64
ins = {k: sc[k].to(d).requires_grad_(True) for k in ("xyz", "scaling", "rotation", "opacity", "features")}
m2 = torch.zeros(P, 3, dtype=d, requires_grad=True)
render_f64(f, ins["xyz"], m2, ins["opacity"], ins["scaling"], ins["rotation"], ins["features"], None, None, cam.world_view_transform,
           cam.full_proj_transform, cam.camera_center, case["bg"], W, H, case["tfx"], case["tfy"], sm, D, dL_dimage=G)
g64 = {"dL_dmeans3D": ins["xyz"].grad, "dL_dmeans2D": m2.grad, "dL_dopacity": ins["opacity"].grad}
k = a.tensor
hp, go, t = h[k].reshape(P, -1).astype(np.float64), g[k].reshape(P, -1).astype(np.float64), g64[k].numpy().reshape(P, -1)
mx = np.abs(go).max()
print(f"seed {seed} P={P} {W}x{H} D={D} sm={sm} R={f['num_rendered']}; {k}: max |oracle| {mx:.4e}; n_contrib per pixel {f['n_contrib'].reshape(-1)[:W * H].tolist()}")
order = np.argsort(-np.abs(hp - go).max(1))[:a.rows]
pl = f["point_list"]
for r in order:
    pos = np.nonzero(pl == r)[0]
    print(f"row {r}: product {hp[r]}  oracle {go[r]}  float64 {t[r]}\n     |p-o|/max {np.abs(hp[r] - go[r]).max() / mx:.2e}  |p-f64|/max {np.abs(hp[r] - t[r]).max() / mx:.2e}  "
          f"|o-f64|/max {np.abs(go[r] - t[r]).max() / mx:.2e}\n     mean2D {f['means2D'][r]} depth {f['depths'][r]:.4f} radius {f['radii'][r]} conic_opacity {f['conic_opacity'][r]} "
          f"list positions {pos.tolist()[:6]} of {len(pl)}")
