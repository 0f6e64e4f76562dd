"""Time simple_knn.distCUDA2 (HIP, through the C ABI) at a few cloud sizes; prints one JSON line per size."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gaussianeditor_amd.simple_knn._C import distCUDA2  # noqa: E402

for P in (100_000, 1_000_000, 6_000_000):
    for kind in ("normal", "surface"):
        g = torch.Generator(device="cuda").manual_seed(P)
        pts = torch.randn(P, 3, device="cuda", generator=g)
        if kind == "surface":
            pts = pts / pts.norm(dim=1, keepdim=True) * (1 + 0.01 * torch.randn(P, 1, device="cuda", generator=g))
        distCUDA2(pts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            distCUDA2(pts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(json.dumps({"op": "distCUDA2", "P": P, "cloud": kind, "ms": round(ms, 3), "Mpoints_per_s": round(P / ms / 1e3, 1)}))
