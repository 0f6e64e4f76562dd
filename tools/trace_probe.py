#!/usr/bin/env python3
"""Stage times of one GaussianRasterizer.apply_weights view (K1 without colours, binning, K12) for the C5 workload shapes:
1 M Gaussians at 512 x 512 and 1920 x 1080, C = 1 and 3.  For A/B builds through GSR_LIBRARY_PATH (tools/build_variants.sh)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from gaussianeditor_amd import options
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianeditor_amd.synth import ring_cameras, synth_scene

    dev = torch.device("cuda", 0)
    sc = synth_scene(1_000_000, seed=0, s0=0.01)
    params = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
    flags = options.current_flags()
    out = []
    for W, H in ((512, 512), (1920, 1080)):
        cam = ring_cameras(12, W, H)[0].to(dev)
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False)
        for C in (1, 3):
            m = (torch.rand(C, H, W, device=dev) > 0.5).float()
            st, R = bench.trace_stage_times(dev, params, rs, m, flags, 20)
            out.append(f"{W}x{H} C={C}: K12 {1e3 * st['trace_weights']:.1f} us (preprocess {1e3 * st['preprocess']:.1f}, bin {1e3 * st['bin']:.1f}; R {R})")
    print(os.environ.get("GSR_LIBRARY_PATH", "product"), " | ".join(out))


if __name__ == "__main__":
    main()
