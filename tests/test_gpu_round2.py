"""-m gpu, round 2: what VERDICT r01 found unpinned or untested.

  * three-way parity reference(nofma) / CPU oracle / HIP product, forward AND backward, with MAX-error assertions
    (pixels whose discrete decisions flip between libm's exp and the specified gsr_expf are counted, bounded, and the
    Gaussians under them masked) -- on the small cases, at 1 M / 1080p (the headline view) and on the 6 M substitute
    of BASELINE configs[1];
  * the L2 boundary render() on the GPU;
  * needle Gaussians (ill-conditioned conics) through the conservative cull;
  * the per-call flags: GSR_FLAG_FAST_EXP parity bars, and that a backward reuses its forward's flags;
  * the gradient exchange on the RCCL backend (world_size 1 always; world_size 2 when two GPUs are visible).
"""
import math
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import hip_state, make_case, oracle_backward, oracle_forward, rel_err, seed_gradient, settings

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRADS = ("dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dmeans2D")


def _np(t):
    return t.detach().cpu().numpy()


def _ref(variant="nofma"):
    from helpers import require_ref

    return require_ref(variant).Reference(variant, DEV)


def _product(case, G, flags=None):
    """Forward + backward of the HIP path through the L1 API -> (color, radii, state dict, grads dict of numpy)."""
    from gaussianeditor_amd import options
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer, _C

    sc, cam = case["sc"], case["cam"]
    P, W, H = sc["xyz"].shape[0], case["W"], case["H"]
    ctx = options.override(flags) if flags is not None else options.override(options.current_flags())
    with ctx:
        leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
        xyz, op, sh, scl, rot = leaf(sc["xyz"]), leaf(sc["opacity"]), leaf(sc["features"]), leaf(sc["scaling"]), leaf(sc["rotation"])
        m2d = torch.zeros_like(xyz, requires_grad=True)
        color, radii, depth = GaussianRasterizer(settings(case, DEV))(xyz, m2d, op, shs=sh, scales=scl, rotations=rot)
        (color * G.to(DEV)).sum().backward()
        # the internal per-pixel state, from a second (deterministic) forward through the _C layer
        e = torch.empty(0, device=DEV)
        d = lambda t: t.to(DEV)  # noqa: E731
        R, c2, _, _, geom, binning, img = _C.rasterize_gaussians(
            d(case["bg"]), d(sc["xyz"]), e, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), 1.0, e,
            d(cam.world_view_transform), d(cam.full_proj_transform), case["tfx"], case["tfy"], H, W, d(sc["features"]),
            case["D"], d(cam.camera_center), False, False)
        assert torch.equal(c2, color.detach())
        st = hip_state(P, R, W, H, geom, binning, img)
    torch.cuda.synchronize()
    grads = dict(dL_dmeans3D=_np(xyz.grad), dL_dopacity=_np(op.grad), dL_dsh=_np(sh.grad), dL_dscales=_np(scl.grad),
                 dL_drotations=_np(rot.grad), dL_dmeans2D=_np(m2d.grad))
    return _np(color), _np(depth), _np(radii), R, st, grads


def _flipped_pixels(nc_a, ft_a, nc_b, ft_b):
    """Pixels whose discrete blend decisions differ between two implementations: a different last contributor, or a
    final transmittance that differs by more than rounding (one skipped / extra alpha >= 1/255 entry moves it by
    >= 0.4 %)."""
    nc_a, nc_b = nc_a.reshape(-1).astype(np.int64), nc_b.reshape(-1).astype(np.int64)
    ft_a, ft_b = ft_a.reshape(-1).astype(np.float64), ft_b.reshape(-1).astype(np.float64)
    rel = np.abs(ft_a - ft_b) / np.maximum(np.maximum(np.abs(ft_a), np.abs(ft_b)), 1e-30)
    return np.nonzero((nc_a != nc_b) | (rel > 1e-4))[0]


def _gaussians_under(pixels, W, f, nc_other):
    """Mask of the Gaussians that are blended at one of `pixels` (flat indices) by either implementation: the entries
    of the pixel's tile list up to its last contributor whose alpha there reaches the threshold (with some slack)."""
    P = f["radii"].shape[0]
    mask = np.zeros(P, bool)
    gx = (W + 15) // 16
    nc_a, nc_b = f["n_contrib"].reshape(-1), np.asarray(nc_other).reshape(-1)
    for p in pixels.tolist():
        px, py = p % W, p // W
        lo, hi = (int(v) for v in f["ranges"][(py // 16) * gx + px // 16])
        n = max(int(nc_a[p]), int(nc_b[p]))
        ids = f["point_list"][lo:min(hi, lo + n)].astype(np.int64)
        co, m = f["conic_opacity"][ids].astype(np.float64), f["means2D"][ids].astype(np.float64)
        dx, dy = m[:, 0] - px, m[:, 1] - py
        power = -0.5 * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
        alpha = co[:, 3] * np.exp(np.minimum(power, 0.0))
        mask[ids[(power <= 1e-6) & (alpha >= 0.5 / 255.0)]] = True
    return mask


ROW_REL, ROW_FLOOR, ROW_FRACTION = 1e-3, 1e-7, 1e-3


def _assert_grads(tag, got, want, masked, tol=1e-5):
    """Two bars per gradient tensor (rows = Gaussians, `masked` rows reported but not asserted):
      1. max |a - b| <= tol * max|b|                      -- the tensor-wide bar of north_star (1e-5);
      2. per ROW: |a - b| <= ROW_FLOOR * max|b| + ROW_REL * |b_row|  -- so that a Gaussian whose own gradient is a tiny
         fraction of the tensor's largest cannot be grossly wrong and hide under bar 1.  A row is a
         cancelling sum over pixels whose binary32 value depends on the summation order (wave reductions + atomics here,
         per-pixel atomics in the reference), so a FEW rows may exceed the per-row bar by rounding alone: at most
         ROW_FRACTION of the rows may, and none of them by more than bar 1."""
    worst = 0.0
    for k in GRADS:
        a = got[k].reshape(got[k].shape[0], -1).astype(np.float64)
        b = want[k].reshape(a.shape).astype(np.float64)
        scale = max(np.abs(b).max(), 1e-30)
        d = np.abs(a - b).max(axis=1)
        err = d / scale
        e_all, e_kept = float(err.max()), float(err[~masked].max()) if (~masked).any() else 0.0
        worst = max(worst, e_kept)
        row = np.abs(b).max(axis=1)
        bad = (d > ROW_FLOOR * scale + ROW_REL * row) & ~masked
        live = (row > 0) & ~masked
        rel = d[live] / np.maximum(row[live], ROW_FLOOR * scale) if live.any() else np.zeros(1)
        print(f"  {tag} {k}: max err {e_kept:.2e} outside the masked rows ({e_all:.2e} with them); per-row relative error "
              f"median {np.median(rel):.1e} p99.9 {np.quantile(rel, 0.999):.1e}; rows over the per-row bar {int(bad.sum())} "
              f"of {int(live.sum())}")
        assert e_kept <= tol, (tag, k, e_kept)
        assert bad.sum() <= ROW_FRACTION * max(1, int(live.sum())) + 2, (tag, k, int(bad.sum()))
    return worst


CASES = [
    # P, W, H, s0, seed, D
    (10000, 256, 256, 0.03, 1, 3),
    (3000, 250, 131, 0.05, 2, 3),
    (20000, 512, 512, 0.02, 4, 3),
    (1_000_000, 1920, 1080, 0.01, 0, 3),  # the headline view of bench.py (synth-v1, seed 0, ring-v1 view 0 of 8)
    (6_000_000, 1920, 1080, 0.01, 0, 3),  # BASELINE configs[1] substitute: 6 M Gaussians, 1080p
    (1_000_000, 1920, 1080, 0.04, 0, 3),  # deep tiles: R ~ 2e7, lists an order of magnitude longer (the regime of real captures)
]


@pytest.mark.parametrize("P,W,H,s0,seed,D", CASES, ids=lambda v: str(v))
def test_three_way_parity_forward_and_backward(oracle, P, W, H, s0, seed, D):
    """reference (its own .cu compiled contraction-free for gfx950) == oracle == product:
    integers bit-exact all three ways; images and ALL gradients by max error <= 1e-5 (of the tensor's max), after
    masking the Gaussians under the pixels whose decisions flip between libm's exp and gsr_expf (counted, bounded).
    Follows backward.cu:399-557 (render) and :144-396 (preprocess)."""
    big = P >= 1_000_000
    case = make_case(P, W, H, seed=seed, s0=s0, view=0, nviews=8 if big else 4, bg=(0.0, 0.0, 0.0) if big else (0.1, 0.2, 0.3))
    G = seed_gradient(H, W, seed) * (1.0 if big else H * W)
    three_way(oracle, case, G, D)


def three_way(oracle, case, G, D):
    """The body of the three-way test for any case dictionary (tests/helpers.py: make_case) -- also run on the synth-v2
    scenes by tests/test_gpu_round5.py."""
    sc, cam = case["sc"], case["cam"]
    P, W, H = sc["xyz"].shape[0], case["W"], case["H"]
    N = W * H
    # --- the reference itself
    R_ = _ref("nofma")
    r = R_.forward(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], None, None,
                   cam.world_view_transform, cam.full_proj_transform, cam.camera_center, case["bg"], W, H, case["tfx"],
                   case["tfy"], 1.0, D)
    gr = {k: _np(v) for k, v in R_.backward(G).items()}
    r_np = {k: _np(v) for k, v in r.items() if isinstance(v, torch.Tensor)}
    del R_
    torch.cuda.empty_cache()
    # --- the oracle
    f = oracle_forward(oracle, case)
    go = oracle_backward(oracle, case, f, G)
    # --- the product
    color, depth, radii, R, st, gp = _product(case, G)

    # integers / indices: bit exact, three ways
    assert r["num_rendered"] == f["num_rendered"] == R
    assert np.array_equal(r_np["radii"], f["radii"]) and np.array_equal(radii, f["radii"])
    assert np.array_equal(r_np["keys"].view(np.uint64), f["keys"]) and np.array_equal(st["keys"], f["keys"])
    assert np.array_equal(r_np["point_list"].view(np.uint32), f["point_list"]) and np.array_equal(st["point_list"], f["point_list"])
    assert np.array_equal(r_np["ranges"].view(np.uint32), f["ranges"]) and np.array_equal(st["ranges"], f["ranges"])
    vis = f["radii"] > 0
    for k in ("means2D", "depths", "conic_opacity", "rgb"):
        assert np.array_equal(r_np[k][vis], f[k][vis]), k  # same IEEE operations in the same order
    # product vs oracle forward: bit exact (same exp by specification)
    assert np.array_equal(st["n_contrib"], f["n_contrib"]) and np.array_equal(st["final_T"], f["final_T"])
    assert np.array_equal(color, f["color"])
    assert np.array_equal(depth, f["depth"])  # out_depth, forward.cu:359, 377: the same accumulation as the colour's

    # reference vs oracle images: identical decisions except at exp-rounding ties
    flips = _flipped_pixels(r_np["n_contrib"].view(np.uint32), r_np["final_T"], f["n_contrib"], f["final_T"])
    dc = np.abs(r_np["color"] - f["color"]).reshape(3, -1)
    keep = np.ones(N, bool)
    keep[flips] = False
    print(f"[{P} @ {W}x{H}] R = {R}; pixels with flipped decisions (libm exp vs gsr_expf): {flips.size} of {N}; "
          f"colour max diff outside them {dc[:, keep].max():.2e} (with them {dc.max():.2e})")
    assert flips.size <= 4 + 2e-4 * N
    assert dc[:, keep].max() <= 1e-5
    cmax = max(1.0, float(np.abs(f["rgb"][vis]).max()))
    assert dc.max() <= 2.1 * cmax / 255.0 + 1e-5  # a flipped 1/255 decision moves a pixel by at most ~alpha (c + C_behind)
    # the depth image (north_star: "RGB/depth ... within 1e-5"; forward.cu:359, 377) against the reference's own, same bar,
    # relative to the largest depth of the view (depths are O(distance to the scene), colours O(1))
    dd = np.abs(r_np["depth"].reshape(-1) - f["depth"].reshape(-1))
    dscale = max(1.0, float(np.abs(r_np["depth"]).max()))
    print(f"  depth image: max diff outside the flipped pixels {dd[keep].max():.2e} (with them {dd.max():.2e}), scale {dscale:.2f}")
    assert dd[keep].max() <= 1e-5 * dscale

    # gradients: max error, flipped pixels' Gaussians masked (their rows are reported, not hidden)
    masked = _gaussians_under(flips, W, f, r_np["n_contrib"].view(np.uint32))
    print(f"  Gaussians under flipped pixels (masked): {int(masked.sum())} of {P}")
    assert masked.sum() <= 0.02 * P + 64
    _assert_grads("oracle vs reference(nofma)", go, gr, masked)
    _assert_grads("product vs reference(nofma)", gp, gr, masked)
    _assert_grads("product vs oracle", gp, go, np.zeros(P, bool))


class _PC:
    """Duck-typed GaussianModel (gaussiansplatting/scene/gaussian_model.py:221-258) on the GPU."""

    def __init__(self, sc, active_sh_degree=3):
        self._sc = {k: v.to(DEV).clone().requires_grad_(True) for k, v in sc.items() if isinstance(v, torch.Tensor) and k != "bg"}
        self.active_sh_degree = active_sh_degree
        self.max_sh_degree = 3

    get_xyz = property(lambda s: s._sc["xyz"])
    get_opacity = property(lambda s: s._sc["opacity"])
    get_scaling = property(lambda s: s._sc["scaling"])
    get_rotation = property(lambda s: s._sc["rotation"])
    get_features = property(lambda s: s._sc["features"])


def test_render_l2_on_gpu(oracle):
    """The L2 boundary, gaussiansplatting/gaussian_renderer/__init__.py:45-150, through the HIP path on cuda:0: the
    returned dict, the image, the gradients that land on the model tensors and on `viewspace_points`, and the
    override_color / convert_SHs_python / semantic_color variants the editor uses."""
    from gaussianeditor_amd.gaussian_renderer import render

    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    case = make_case(10000, 256, 256, seed=0, s0=0.03, nviews=1, bg=(0.0, 0.0, 0.0))
    pc = _PC(case["sc"])
    cam = case["cam"].to(DEV)
    out = render(cam, pc, pipe, case["bg"].to(DEV))
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "depth_3dgs"}
    assert out["render"].is_cuda and out["render"].shape == (3, 256, 256) and out["depth_3dgs"].shape == (1, 256, 256)
    assert out["radii"].dtype == torch.int32 and out["visibility_filter"].dtype == torch.bool
    f = oracle_forward(oracle, case)
    assert np.array_equal(_np(out["render"]), f["color"]) and np.array_equal(_np(out["depth_3dgs"]), f["depth"])
    assert np.array_equal(_np(out["radii"]), f["radii"])
    G = seed_gradient(256, 256, 0) * 256 * 256
    (out["render"] * G.to(DEV)).sum().backward()
    g = oracle_backward(oracle, case, f, G)
    assert rel_err(_np(pc.get_xyz.grad), g["dL_dmeans3D"]) <= 1e-5
    assert rel_err(_np(pc.get_features.grad), g["dL_dsh"]) <= 1e-5
    assert rel_err(_np(pc.get_opacity.grad), g["dL_dopacity"]) <= 1e-5
    assert rel_err(_np(pc.get_scaling.grad), g["dL_dscales"]) <= 1e-5
    assert rel_err(_np(pc.get_rotation.grad), g["dL_drotations"]) <= 1e-5
    # the screen-space gradient lands on the dummy tensor, as add_densification_stats expects (gaussian_model.py:811-815)
    assert rel_err(_np(out["viewspace_points"].grad), g["dL_dmeans2D"]) <= 1e-5
    assert float(out["viewspace_points"].grad[:, 2].abs().max()) == 0.0
    # override_color (the editor's mask render), SH evaluated in PyTorch, and the fused semantic image
    mask = (torch.rand(10000, 1, generator=torch.Generator().manual_seed(1)) > 0.5).float().repeat(1, 3)
    c = render(cam, pc, pipe, case["bg"].to(DEV), override_color=mask.to(DEV))["render"]
    fm = oracle_forward(oracle, case, colors_precomp=mask)
    assert np.array_equal(_np(c), fm["color"])
    b = render(cam, pc, SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=True, debug=False), case["bg"].to(DEV))
    assert float((b["render"].detach() - out["render"].detach()).abs().max()) < 2e-5
    s = render(cam, pc, pipe, case["bg"].to(DEV), semantic_color=mask.to(DEV))
    assert torch.equal(s["semantic"], c.detach()) and torch.equal(s["render"].detach(), out["render"].detach())


def _needle_case(P=4000, W=320, H=240, seed=11):
    """A scene whose first quarter are needles: one scale 30-3000x the others, random orientation, close to the camera
    (screen-space major sigma up to thousands of pixels, conics with rho = det / (A C) down to 1e-7)."""
    case = make_case(P, W, H, seed=seed, s0=0.02)
    sc = case["sc"]
    g = torch.Generator().manual_seed(seed)
    n = P // 4
    sc["scaling"][:n, 0] = 10 ** (-0.5 + 1.5 * torch.rand(n, generator=g))     # 0.3 .. 10 world units long
    sc["scaling"][:n, 1:] = 10 ** (-3.5 + 1.0 * torch.rand(n, 2, generator=g))  # 3e-4 .. 3e-3 thin
    sc["opacity"][:n] = 0.05 + 0.95 * torch.rand(n, 1, generator=g)
    # half of the needles exactly diagonal in the image plane would need the camera frame; random rotations cover it
    return case


@pytest.mark.parametrize("bounds", ["reference", "alpha"])
def test_needle_gaussians_match_oracle(oracle, bounds):
    """the cull box of an ill-conditioned conic was not conservative (binary32 cancellation in
    det(conic)); such entries are no longer culled.  Images, n_contrib and gradients must equal the oracle's with the
    needles in the scene, under both binning rules (forward.cu:335-344 decides per pixel, nothing else may)."""
    import gaussianeditor_amd

    case = _needle_case()
    W, H = case["W"], case["H"]
    f = oracle_forward(oracle, case)
    co = f["conic_opacity"][f["radii"] > 0]
    rho = (co[:, 0] * co[:, 2] - co[:, 1] ** 2) / np.maximum(co[:, 0] * co[:, 2], 1e-30)
    print(f"visible {co.shape[0]}, conics with rho < 1e-3: {int((rho < 1e-3).sum())}, largest radius {int(f['radii'].max())} px")
    assert (rho < 1e-3).sum() >= 50  # the scene really contains what the test is about
    G = seed_gradient(H, W, 3) * (H * W)
    g = oracle_backward(oracle, case, f, G)
    try:
        gaussianeditor_amd.set_tile_bounds(bounds)
        color, depth, radii, R, st, gp = _product(case, G)
    finally:
        gaussianeditor_amd.set_tile_bounds("reference")
    assert np.array_equal(radii, f["radii"])
    assert np.array_equal(color, f["color"]) and np.array_equal(depth, f["depth"])
    assert np.array_equal(st["final_T"], f["final_T"])
    if bounds == "reference":
        assert R == f["num_rendered"] and np.array_equal(st["n_contrib"], f["n_contrib"])
    else:
        assert 0 < R <= f["num_rendered"]
    # Gradients.  The well-conditioned Gaussians (rows >= P/4) meet the usual bar although needles lie in front of
    # and behind them.  A needle's own gradient is a sum over up to 1e5 pixels of terms ~ q dx^2 with |dx| up to
    # thousands of pixels that cancel to a small total: in binary32 its value depends on the summation order (the
    # oracle adds pixel after pixel, the kernels reduce waves and tiles, the reference's atomics any order), so those
    # rows are only required to be finite and to agree in bulk.
    n = case["sc"]["xyz"].shape[0] // 4
    # the reference's own backward on the same scene: how far IT is from the oracle on the needle rows
    from oracle import ref as _refmod

    gr = None
    import os as _os

    if _os.environ.get("GSR_REQUIRE_REF") == "1":
        from helpers import require_ref

        require_ref("nofma")
    if _refmod.available("nofma"):
        sc, cam = case["sc"], case["cam"]
        R_ = _refmod.Reference("nofma", DEV)
        R_.forward(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], None, None,
                   cam.world_view_transform, cam.full_proj_transform, cam.camera_center, case["bg"], W, H, case["tfx"],
                   case["tfy"], 1.0, case["D"])
        gr = {k: _np(v) for k, v in R_.backward(G).items()}
    for k in GRADS:
        a = gp[k].reshape(gp[k].shape[0], -1).astype(np.float64)
        b = g[k].reshape(a.shape).astype(np.float64)
        scale = max(np.abs(b[n:]).max(), 1e-30)
        e_rest = float(np.abs(a[n:] - b[n:]).max() / scale)
        l2 = lambda x, y: float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30))  # noqa: E731
        row = np.abs(a[:n] - b[:n]).max(axis=1) / np.maximum(np.abs(b[:n]).max(axis=1), 1e-30)
        l2_prod = l2(a[:n], b[:n])
        l2_ref = l2(gr[k].reshape(a.shape).astype(np.float64)[:n], b[:n]) if gr is not None else float("nan")
        print(f"  {bounds} {k}: well-conditioned rows max err {e_rest:.2e}; needle rows: median row error {np.median(row):.2e}, "
              f"rel-L2 product vs oracle {l2_prod:.2e}, reference(nofma) vs oracle {l2_ref:.2e}")
        assert np.isfinite(a).all(), k
        assert e_rest <= 1e-5, k
        if gr is not None:  # no further from the oracle than the reference's own atomics put it
            assert l2_prod <= 3.0 * l2_ref + 0.05, k


def test_fast_exp_flag_parity_and_flag_pinning(oracle):
    """GSR_FLAG_FAST_EXP (opt-in): hardware 2^x in the blend loops.  Bars: integer outputs of K1-K5 untouched; images
    within 1e-5 of the oracle outside the (counted, bounded) pixels whose threshold decisions flip; gradients within
    1e-5 outside the Gaussians under those pixels.  And: a backward runs with the flags of ITS forward, whatever the
    default has become in between."""
    import gaussianeditor_amd
    from gaussianeditor_amd import options
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    P, W, H = 20000, 512, 512
    case = make_case(P, W, H, seed=4, s0=0.02)
    G = seed_gradient(H, W, 4) * (H * W)
    f = oracle_forward(oracle, case)
    g = oracle_backward(oracle, case, f, G)
    color, depth, radii, R, st, gp = _product(case, G, flags=options.FLAG_FAST_EXP)
    assert R == f["num_rendered"] and np.array_equal(radii, f["radii"])
    assert np.array_equal(st["keys"], f["keys"]) and np.array_equal(st["point_list"], f["point_list"])
    flips = _flipped_pixels(st["n_contrib"], st["final_T"], f["n_contrib"], f["final_T"])
    keep = np.ones(W * H, bool)
    keep[flips] = False
    dc = np.abs(color - f["color"]).reshape(3, -1)
    print(f"fast exp: flipped pixels {flips.size} of {W * H}; colour max diff outside them {dc[:, keep].max():.2e}, "
          f"with them {dc.max():.2e}; depth {np.abs(depth - f['depth']).reshape(-1)[keep].max():.2e}")
    assert flips.size <= 4 + 2e-4 * W * H
    cmax = max(1.0, float(np.abs(f["rgb"][f["radii"] > 0]).max()))
    assert dc[:, keep].max() <= 1e-5 and dc.max() <= 2.1 * cmax / 255.0 + 1e-5
    masked = _gaussians_under(flips, W, f, st["n_contrib"])
    _assert_grads("fast-exp product vs oracle", gp, g, masked)
    # flag pinning: forward under FAST_EXP, default flipped back before the backward -> the backward still gets FAST_EXP
    from gaussianeditor_amd.diff_gaussian_rasterization import _C

    sc = case["sc"]
    leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
    xyz, op, sh, scl, rot = leaf(sc["xyz"]), leaf(sc["opacity"]), leaf(sc["features"]), leaf(sc["scaling"]), leaf(sc["rotation"])
    m2d = torch.zeros_like(xyz, requires_grad=True)
    seen = []
    orig = _C.rasterize_gaussians_backward

    def spy(*a, flags=None, **kw):
        seen.append(flags)
        return orig(*a, flags=flags, **kw)

    _C.rasterize_gaussians_backward = spy
    try:
        gaussianeditor_amd.set_fast_exp(True)
        try:
            c, _, _ = GaussianRasterizer(settings(case, DEV))(xyz, m2d, op, shs=sh, scales=scl, rotations=rot)
        finally:
            gaussianeditor_amd.set_fast_exp(False)
        assert options.current_flags() == 0
        (c * G.to(DEV)).sum().backward()
    finally:
        _C.rasterize_gaussians_backward = orig
    assert seen == [options.FLAG_FAST_EXP]
    assert np.array_equal(_np(c), color)
    for k, t in (("dL_dmeans3D", xyz), ("dL_dopacity", op), ("dL_dsh", sh), ("dL_dmeans2D", m2d)):
        assert rel_err(_np(t.grad), gp[k]) <= 2e-6, k  # the atomics' run-to-run spread, nothing more


def _run_exchange_check(nproc, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "rccl_exchange_check.py")]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)


def test_gradient_exchange_on_rccl_world1():
    """multiview_step with backend "nccl" (= RCCL), every route, collectives forced although the group has one rank:
    proves each collective call of the step (dtypes, shapes, all_gather_into_tensor, async MAX) against the real
    backend on a single-GPU box."""
    p = _run_exchange_check(1, 29531)
    print(p.stdout[-2000:], p.stderr[-3000:])
    assert p.returncode == 0 and "all routes agree" in p.stdout


def test_gradient_exchange_on_rccl_world2():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU)")
    p = _run_exchange_check(2, 29533)
    print(p.stdout[-2000:], p.stderr[-3000:])
    assert p.returncode == 0 and "all routes agree" in p.stdout


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` must never report a smaller job under that label."""
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY.md section 8(f) rank 4, second half: densification appends (gsr_append_rows)
@pytest.mark.parametrize("P,n", [(0, 5), (1, 0), (1, 1), (1023, 77), (100001, 12345), (5, 4096)])
def test_append_rows_equals_torch_cat(oracle, P, n):
    from gaussianeditor_amd.densify import append_rows

    g = torch.Generator().manual_seed(P + n)
    mk = lambda rows: [torch.randn(rows, 3, generator=g), torch.randn(rows, 15, 3, generator=g), torch.randn(rows, 1, generator=g),  # noqa: E731
                       torch.randn(rows, 4, generator=g), torch.arange(rows, dtype=torch.int64),
                       torch.rand(rows, generator=g) > 0.5, torch.randint(0, 255, (rows, 3), generator=g, dtype=torch.uint8)]
    ts, es = mk(P), mk(n)
    es[1] = None  # zero rows, as for an Adam moment
    es[6] = None
    out = append_rows([t.to(DEV) for t in ts], [None if e is None else e.to(DEV) for e in es], n=n)
    ref = oracle.append_rows([t.numpy() for t in ts], [None if e is None else e.numpy() for e in es], n)
    for o, r, t, e in zip(out, ref, ts, es):
        assert o.dtype == t.dtype and o.shape[0] == P + n
        assert np.array_equal(_np(o), r)
        want = torch.cat((t, torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype) if e is None else e), dim=0)
        assert torch.equal(o.cpu(), want)


def test_cat_tensors_to_optimizer_and_clone_like_reference():
    """cat_tensors_to_optimizer / clone_rows == GaussianModel.cat_tensors_to_optimizer (gaussian_model.py:609-641) after
    the selection of densify_and_clone (:743-748), on an Adam with state: parameters, both moments, Parameter identity in
    the optimizer's state dict; and the grown optimizer keeps working."""
    from gaussianeditor_amd.densify import cat_tensors_to_optimizer, clone_rows
    from gaussianeditor_amd.optim import FusedMaskedAdam

    P = 5003
    gen = torch.Generator().manual_seed(3)
    init = {"xyz": torch.randn(P, 3, generator=gen), "f_dc": torch.randn(P, 1, 3, generator=gen),
            "f_rest": torch.randn(P, 15, 3, generator=gen), "opacity": torch.randn(P, 1, generator=gen),
            "scaling": torch.randn(P, 3, generator=gen), "rotation": torch.randn(P, 4, generator=gen)}

    def build():
        params = {k: v.clone().to(DEV).requires_grad_(True) for k, v in init.items()}
        opt = FusedMaskedAdam([{"params": [p], "lr": 1e-3, "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
        g2 = torch.Generator().manual_seed(1)
        for p in params.values():
            p.grad = torch.randn(p.shape, generator=g2).to(DEV)
        opt.step()
        return opt

    def reference_cat(opt, d):  # the reference's loop, verbatim in behaviour
        res = {}
        for group in opt.param_groups:
            ext = d[group["name"]]
            st = opt.state.get(group["params"][0], None)
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del opt.state[group["params"][0]]
            group["params"][0] = torch.nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
            opt.state[group["params"][0]] = st
            res[group["name"]] = group["params"][0]
        return res

    sel = (torch.rand(P, generator=gen) < 0.13).to(DEV)
    a, b = build(), build()
    new = clone_rows(a, sel)
    reference_cat(b, {g["name"]: g["params"][0][sel] for g in b.param_groups})
    n = int(sel.sum())
    for ga, gb in zip(a.param_groups, b.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert new[ga["name"]] is pa and pa.requires_grad and pa.shape[0] == P + n and torch.equal(pa, pb)
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(a.state[pa][k], b.state[pb][k]) and float(a.state[pa][k][P:].abs().max()) == 0.0
    # the split path hands over tensors computed in torch
    ext = {g["name"]: torch.randn((7,) + tuple(g["params"][0].shape[1:]), generator=gen).to(DEV) for g in a.param_groups}
    cat_tensors_to_optimizer(a, ext)
    reference_cat(b, ext)
    for ga, gb in zip(a.param_groups, b.param_groups):
        assert torch.equal(ga["params"][0], gb["params"][0])
        assert torch.equal(a.state[ga["params"][0]]["exp_avg_sq"], b.state[gb["params"][0]]["exp_avg_sq"])
    for p in (g["params"][0] for g in a.param_groups):
        p.grad = torch.ones_like(p)
    a.step()


def test_images_beyond_65535_tiles_use_wide_tile_keys(oracle):
    """Tile ids are sorted as uint16 when they fit (6 instead of 8 bytes per sorted pair); an image of 257 x 257 = 66049
    tiles takes the uint32 path.  Both against the oracle: keys, lists, ranges bit-exact."""
    from gaussianeditor_amd import _native

    for W, H, wide in ((4112, 4112, True), (4096, 4080, False)):
        assert (int(_native.lib().gsr_sort_key_bits(W, H)) - 32 > 16) == wide
        case = make_case(3000, W, H, seed=6, s0=0.05, nviews=3, view=1)
        f = oracle_forward(oracle, case)
        G = torch.zeros(3, H, W)
        G[:, ::7, ::5] = 1.0
        color, depth, radii, R, st, gp = _product(case, G)
        assert R == f["num_rendered"] and R > 0
        assert np.array_equal(st["keys"], f["keys"]) and np.array_equal(st["point_list"], f["point_list"])
        assert np.array_equal(st["ranges"], f["ranges"]) and np.array_equal(st["n_contrib"], f["n_contrib"])
        assert np.array_equal(color, f["color"])
        g = oracle_backward(oracle, case, f, G)
        for k in GRADS:
            assert rel_err(gp[k], g[k].reshape(gp[k].shape)) <= 1e-5, k


def test_forward_select_and_masked_chunks_mix_bit_exactly(oracle):
    """Round 2, forward blend: chunks whose staged colours / depths are all finite run the select-only serial part, a chunk
    holding a NaN / Inf / overflowing colour runs the masked one, and a quadrant's per-pixel state crosses between the two
    from chunk to chunk.  A deep scene (lists of several chunks per tile) with a sprinkle of such colours must reproduce
    the oracle: integer outputs and the set of non-finite pixels exactly, finite pixels to 1e-5 (they are bit-identical
    in practice: same operations in the same order)."""
    from test_gpu_parity import _run_hip_forward

    case = make_case(20000, 256, 192, seed=23, s0=0.06)
    P = 20000
    gen = torch.Generator().manual_seed(5)
    colors = torch.rand(P, 3, generator=gen)
    pick = torch.randperm(P, generator=gen)
    colors[pick[:60], 0] = float("nan")
    colors[pick[60:120], 1] = float("inf")
    colors[pick[120:180], 2] = -float("inf")
    colors[pick[180:260]] = 3.0e38  # finite, but the chunk's screening sum overflows: masked form, finite result path
    f = oracle_forward(oracle, case, colors_precomp=colors)
    R, color, depth, radii, geom, binning, img = _run_hip_forward(case, colors_precomp=colors)
    st = hip_state(P, R, 256, 192, geom, binning, img)
    assert R == f["num_rendered"] and np.array_equal(st["n_contrib"], f["n_contrib"])
    assert np.abs(st["final_T"] - f["final_T"]).max() <= 1e-6
    col, ref = color.cpu().numpy(), f["color"]
    bad = ~np.isfinite(ref)
    assert np.array_equal(col[bad], ref[bad], equal_nan=True)  # the same NaN, +Inf and -Inf pixels
    assert 0 < int(bad.sum()) < ref.size // 2  # the sprinkle reaches some pixels and leaves most alone
    big = np.abs(ref) > 1e30
    assert np.abs(col[~bad & ~big] - ref[~bad & ~big]).max() <= 1e-5 and rel_err(col[~bad & big], ref[~bad & big]) <= 1e-5
    print("finite pixels bit-identical:", bool(np.array_equal(col[~bad], ref[~bad])))
    assert rel_err(depth.cpu().numpy(), f["depth"]) <= 1e-5
    # lists long enough that quadrants walk several chunks (otherwise the test would not cross between the two forms)
    lens = (st["ranges"][:, 1] - st["ranges"][:, 0])
    assert int(np.percentile(lens, 90)) > 3 * 64
