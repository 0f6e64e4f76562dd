#!/usr/bin/env python3
"""Delete path, neighbour query: gsr_near_points on the GPU vs the reference's route (both point sets to the host, a scipy
KDTree, distances back: gaussiansplatting/knn.py) at an edit-sized scene.  Prints a small markdown table."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--radius", type=float, default=1.25, help="radius of the masked ball (object) in a N(0, 2^2) cloud")
    ap.add_argument("--thresh", type=float, default=0.1)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from scipy.spatial import KDTree

    from gaussianeditor_amd import near

    rng = np.random.default_rng(11)
    xyz_h = (rng.standard_normal((a.points, 3)) * 2.0).astype(np.float32)
    mask_h = np.linalg.norm(xyz_h - np.float32([0.5, 0, 0]), axis=1) < a.radius
    xyz, mask = torch.from_numpy(xyz_h).cuda(), torch.from_numpy(mask_h).cuda()

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps, out

    t_full, got = timed(lambda: near.get_near_gaussians_by_mask(xyz, mask, a.thresh), a.reps)
    obj, rem = xyz[mask], xyz[~mask]
    t_query, _ = timed(lambda: near.near_points(obj, rem, a.thresh), a.reps)
    t_query_d, _ = timed(lambda: near.near_points(obj, rem, a.thresh, return_dist=True), a.reps)

    def reference_route():  # knn.py: .cpu().numpy(), KDTree(mean), query(k=1), back to the device
        o, r = obj.detach().cpu().numpy(), rem.detach().cpu().numpy()
        d, i = KDTree(o).query(r, k=1)
        return torch.from_numpy(d).to(obj) <= a.thresh

    t0 = time.perf_counter()
    want = reference_route()
    t_ref = time.perf_counter() - t0
    same = bool((near.near_points(obj, rem, a.thresh) == want).all())
    print(f"| points | object | remaining | near (all remaining) | near (in box) | gsr_near_points, mask only | with distances | "
          f"get_near_gaussians_by_mask (quantiles + box + query) | reference route (KDTree on the host) | same mask |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    print(f"| {a.points} | {int(mask_h.sum())} | {int((~mask_h).sum())} | {int(want.sum())} | {int(got.sum())} | "
          f"{t_query * 1e3:.2f} ms | {t_query_d * 1e3:.2f} ms | {t_full * 1e3:.2f} ms | {t_ref * 1e3:.0f} ms | {same} |")


if __name__ == "__main__":
    main()
