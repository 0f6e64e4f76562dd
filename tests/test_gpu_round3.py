"""-m gpu, round 3: the binning paths.

The grouped binning (gsr_binning.hip: group instances -> one radix pass -> 64 x 64 bit transposes) is what every other GPU
test of the suite runs on, bit-exact against the oracle and the reference's own CUB-sorted output.  Here:
  * the tile-pair sort that images beyond 131 072 tiles still take, forced by GSR_BIN_LEGACY=1 in a fresh process (the
    library reads the variable once);
  * batches whose entries exceed the LDS stage of the list-append kernel (every item covers most of its group), which are
    taken in halves / quarters of their items;
  * images of one group, of one tile row, and with 9 / 11-bit group ids.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import hip_state, make_case, oracle_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_legacy_pair_sort_path_is_still_exact():
    env = dict(os.environ, GSR_BIN_LEGACY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-k", "test_forward_all_stages or test_backward_vs_oracle or test_apply_weights or test_empty_and_fully_culled "
                              "or test_4k_image_many_tiles"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def _lists_match(case):
    from gaussianeditor_amd.diff_gaussian_rasterization import _C

    sc, cam = case["sc"], case["cam"]
    f = oracle_forward(__import__("oracle.cpu", fromlist=["cpu"]), case)
    e = torch.empty(0, device=DEV)
    d = lambda t: t.to(DEV)  # noqa: E731
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        d(case["bg"]), d(sc["xyz"]), e, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), 1.0, e,
        d(cam.world_view_transform), d(cam.full_proj_transform), case["tfx"], case["tfy"], case["H"], case["W"],
        d(sc["features"]), case["D"], d(cam.camera_center), False, False)
    st = hip_state(sc["xyz"].shape[0], R, case["W"], case["H"], geom, binning, img)
    assert R == f["num_rendered"]
    assert np.array_equal(st["point_list"], f["point_list"]) and np.array_equal(st["keys"], f["keys"])
    assert np.array_equal(st["ranges"], f["ranges"])
    assert np.array_equal(st["n_contrib"], f["n_contrib"]) and np.array_equal(color.cpu().numpy(), f["color"])
    return R, f


@pytest.mark.parametrize("P,W,H,s0", [
    (3000, 640, 360, 0.6),     # every splat covers the image: each group instance covers 64 tiles (quarter batches)
    (20000, 1024, 1024, 0.15), # ~25 tiles per group instance: half batches, several flushes per chunk
    (5000, 128, 128, 0.05),    # ONE group: nothing to sort
    (5000, 2048, 16, 0.05),    # one tile row, 16 groups
    (200000, 1920, 1080, 0.02),
])
def test_grouped_binning_shapes(oracle, P, W, H, s0):
    R, f = _lists_match(make_case(P, W, H, seed=9, s0=s0))
    print(f"P={P} {W}x{H} s0={s0}: R={R}, {R / max(1, int((f['radii'] > 0).sum())):.1f} tiles per visible Gaussian")
