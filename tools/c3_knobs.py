#!/usr/bin/env python3
"""Stage times of ONE view of the editor's loop shape (BASELINE configs[2]: 1 M Gaussians through a 512 x 512 image) under the
launch-shape knobs the environment carries (GSR_BWD_SEG, GSR_BWD_HALVES, GSR_CK_CHUNKS, GSR_FWD_SPLIT ...): one process per
setting (the library reads them once).   python tools/c3_knobs.py [--width 512 --height 512]"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--scene", default="v1")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--s0", type=float, default=0.01)
    a = ap.parse_args()
    import bench
    from gaussianeditor_amd import options
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene, synth_scene_v2

    dev = torch.device("cuda", 0)
    W, H = a.width, a.height
    sc = synth_scene_v2(a.gaussians, seed=0) if a.scene == "v2" else synth_scene(a.gaussians, seed=0, s0=a.s0)
    params = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
    cam = ring_cameras(8, W, H)[0].to(dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), sc["bg"].to(dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    G = seed_gradient(H, W, 0).to(dev)
    info = {}
    st, R, V, pix = bench.stage_times(dev, params, rs, G, 3, options.current_flags(), 12, info=info)
    knobs = {k: v for k, v in os.environ.items() if k.startswith("GSR_") and k not in ("GSR_REQUIRE_REF",)}
    print(f"{W}x{H} {a.scene} P={a.gaussians} s0={a.s0} R={R}", knobs, {k: round(1e3 * v, 1) for k, v in st.items()}, "sum", round(1e3 * sum(st.values()), 1),
          {k: (round(v, 3) if isinstance(v, float) else v) for k, v in info.items()})


if __name__ == "__main__":
    main()
