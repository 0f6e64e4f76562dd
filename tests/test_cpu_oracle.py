"""CPU tests of the oracle itself: golden vectors generated from the reference's Python modules,
structural invariants of the binning, the exactly-specified exp, and the analytic backward against
float64 autograd."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import make_case, oracle_backward, oracle_forward, rel_err, seed_gradient

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gsr_expf_accuracy(oracle):
    x = np.concatenate([np.linspace(-12, 0, 20001), np.linspace(-85, -12, 2001), [-0.0, 0.0, -1e-30]]).astype(np.float32)
    got = oracle.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    near = x >= -12
    assert np.max(np.abs(got[near] - ref[near]) / ref[near]) < 1.0e-6
    assert np.max(np.abs(got[~near] - ref[~near]) / ref[~near]) < 1.0e-5
    assert oracle.expf(np.array([0.0], np.float32))[0] == 1.0
    assert oracle.expf(np.array([-1000.0], np.float32))[0] < 1e-37  # clamped, tiny, never NaN


def test_sh_matches_reference_eval_sh(oracle):
    """Golden: the reference's own eval_sh (utils/sh_utils.py:57-112) pins the oracle's SH->RGB."""
    g = np.load(os.path.join(GOLD, "sh_eval.npz"))
    shs, dirs = g["shs"], g["dirs"]
    P = shs.shape[0]
    # `campos` only feeds the SH direction (forward.cu:25-27), the view/projection matrices are independent of
    # it: put the points on a sphere around campos = 0 so that normalize(p - campos) == dirs, and look at that
    # sphere from a real camera 10 units away so that every point is in front of it and on screen.
    from gaussianeditor_amd.synth import look_at_camera

    campos = np.zeros(3, np.float32)
    means = (dirs * 3.0).astype(np.float32)
    cam = look_at_camera([0.0, 0.0, -10.0], [0.0, 0.0, 0.0], 64, 64, fovy_deg=60.0)
    view, proj = cam.world_view_transform.numpy(), cam.full_proj_transform.numpy()
    tf = math.tan(cam.FoVy / 2)
    for deg in range(4):
        geom = oracle.preprocess(means, np.full((P, 3), 0.01, np.float32), np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32),
                                 np.full((P, 1), 0.5, np.float32), shs, None, None, view, proj, campos, 64, 64, tf, tf,
                                 1.0, deg)
        vis = geom["radii"] > 0
        assert vis.sum() == P
        ref = np.maximum(g[f"rgb_deg{deg}"] + 0.5, 0.0)
        assert np.abs(geom["rgb"][vis] - ref[vis]).max() < 2e-6
        assert np.array_equal(geom["clamped"][vis].astype(bool), (g[f"rgb_deg{deg}"] + 0.5 < 0)[vis]) or \
            np.abs((g[f"rgb_deg{deg}"] + 0.5)[vis][geom["clamped"][vis].astype(bool) != ((g[f"rgb_deg{deg}"] + 0.5) < 0)[vis]]).max() < 1e-6


def test_camera_conventions_match_reference():
    """Golden: synth.py's camera builder reproduces getWorld2View2 / getProjectionMatrix /
    Simple_Camera (scene/cameras.py:92-95) bit for bit."""
    from gaussianeditor_amd import synth

    g = np.load(os.path.join(GOLD, "cameras.npz"))
    for i in range(4):
        R, T = g[f"R{i}"], g[f"T{i}"]
        fovx, fovy = g[f"fov{i}"]
        wv = torch.tensor(synth._world2view(R, T)).transpose(0, 1)
        proj = synth._projection(0.01, 100.0, float(fovx), float(fovy)).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        assert np.array_equal(wv.numpy(), g[f"world_view{i}"])
        assert np.array_equal(proj.numpy(), g[f"proj{i}"])
        assert np.array_equal(full.numpy(), g[f"full_proj{i}"])
        assert np.array_equal(wv.inverse()[3, :3].numpy(), g[f"center{i}"])


def test_projection_agrees_with_camera_model(oracle):
    """The flat-index convention m[4c+r] = M[r][c]: a point on the optical axis lands at the
    principal point, depth equals camera-space z."""
    case = make_case(1, 64, 48, seed=1)
    cam = case["cam"]
    p = torch.zeros(1, 3)  # ring cameras look at the origin
    geom = oracle.preprocess(p, torch.full((1, 3), 0.01), torch.tensor([[1.0, 0, 0, 0]]), torch.tensor([[0.5]]),
                             torch.zeros(1, 16, 3), None, None, cam.world_view_transform, cam.full_proj_transform,
                             cam.camera_center, 64, 48, case["tfx"], case["tfy"], 1.0, 0)
    assert geom["radii"][0] > 0
    assert abs(geom["means2D"][0, 0] - (64 - 1) / 2) < 1e-3 and abs(geom["means2D"][0, 1] - (48 - 1) / 2) < 1e-3
    assert abs(geom["depths"][0] - 4.0) < 1e-5


@pytest.mark.parametrize("P,W,H", [(10000, 256, 256), (3000, 250, 131)])
def test_binning_invariants(oracle, P, W, H):
    case = make_case(P, W, H, seed=3, s0=0.03)
    f = oracle_forward(oracle, case)
    R = f["num_rendered"]
    assert R == int(f["tiles_touched"].sum()) == f["keys"].shape[0]
    bits = oracle.sort_bits(W, H)
    masked = f["keys"] & np.uint64((1 << bits) - 1)
    assert np.all(masked[1:] >= masked[:-1])                      # sorted
    assert np.array_equal(np.sort(f["keys"]), np.sort(f["keys_unsorted"]))  # a permutation of the emitted pairs
    # equal keys keep emission order (stability): within equal keys the Gaussian index is ascending
    same = f["keys"][1:] == f["keys"][:-1]
    assert np.all(f["point_list"][1:][same] > f["point_list"][:-1][same])
    # ranges partition [0, R) in tile order and match the key's tile id
    tiles = (f["keys"] >> np.uint64(32)).astype(np.int64)
    rl = f["ranges"].astype(np.int64)
    for t in np.unique(tiles):
        lo, hi = rl[t]
        assert np.all(tiles[lo:hi] == t) and (lo == 0 or tiles[lo - 1] != t) and (hi == R or tiles[hi] != t)
    assert (rl[:, 1] - rl[:, 0]).sum() == R
    # n_contrib never exceeds its tile's list length; depth-sorted inside each tile
    gx = (W + 15) // 16
    nc = f["n_contrib"].reshape(H, W)
    for ty in range((H + 15) // 16):
        for tx in range(gx):
            lo, hi = rl[ty * gx + tx]
            assert nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].max(initial=0) <= hi - lo
    d = f["depths"][f["point_list"]]
    brk = tiles[1:] == tiles[:-1]
    assert np.all(d[1:][brk] >= d[:-1][brk])
    assert np.all(f["color"] >= 0) and np.all(f["final_T"] <= 1.0) and np.all(f["final_T"] > 0)


def test_backward_matches_float64_autograd(oracle):
    """The oracle's analytic backward (restating backward.cu) against torch.autograd of an independent
    float64 forward (oracle/torch_ref.py)."""
    from oracle.torch_ref import render_f64

    W = H = 64
    case = make_case(300, W, H, seed=2, s0=0.08, view=1, scale_xyz=0.6, bg=(0.2, 0.5, 0.7))
    sc, cam = case["sc"], case["cam"]
    f = oracle_forward(oracle, case)
    G = seed_gradient(H, W, 1) * H * W
    g = oracle_backward(oracle, case, f, G)
    d = torch.float64
    ins = {k: sc[k].to(d).requires_grad_(True) for k in ["xyz", "scaling", "rotation", "opacity", "features"]}
    m2 = torch.zeros(300, 3, dtype=d, requires_grad=True)
    img = render_f64(f, ins["xyz"], m2, ins["opacity"], ins["scaling"], ins["rotation"], ins["features"], None, None,
                     cam.world_view_transform, cam.full_proj_transform, cam.camera_center, case["bg"], W, H, case["tfx"],
                     case["tfy"], 1.0, 3)
    assert np.abs(img.detach().float().numpy() - f["color"]).max() < 2e-6
    (img * G.to(d)).sum().backward()
    for k, t in (("dL_dmeans3D", ins["xyz"]), ("dL_dmeans2D", m2), ("dL_dopacity", ins["opacity"]),
                 ("dL_dscales", ins["scaling"]), ("dL_drotations", ins["rotation"]), ("dL_dsh", ins["features"])):
        assert rel_err(g[k].reshape(t.shape), t.grad.numpy()) < 2e-5, k


def test_backward_precomp_paths_match_autograd(oracle):
    from oracle.torch_ref import render_f64

    W, H = 48, 40
    case = make_case(150, W, H, seed=4, s0=0.1, view=2, scale_xyz=0.5)
    sc, cam = case["sc"], case["cam"]
    cols = torch.rand(150, 3, generator=torch.Generator().manual_seed(5))
    cov = torch.from_numpy(oracle_forward(oracle, case)["cov3D"].copy())
    f = oracle_forward(oracle, case, colors_precomp=cols, cov3D_precomp=cov)
    G = seed_gradient(H, W, 2) * H * W
    g = oracle_backward(oracle, case, f, G, colors_precomp=cols, cov3D_precomp=cov)
    d = torch.float64
    xyz, op = sc["xyz"].to(d).requires_grad_(True), sc["opacity"].to(d).requires_grad_(True)
    c64, v64 = cols.to(d).requires_grad_(True), cov.to(d).requires_grad_(True)
    img = render_f64(f, xyz, None, op, None, None, None, c64, v64, cam.world_view_transform, cam.full_proj_transform,
                     cam.camera_center, case["bg"], W, H, case["tfx"], case["tfy"], 1.0, 0)
    (img * G.to(d)).sum().backward()
    # dL_dcov3D: the reference's convention doubles off-diagonal terms (backward.cu:221-227) exactly as autograd of
    # the 6-vector parametrisation does
    for k, t in (("dL_dmeans3D", xyz), ("dL_dopacity", op), ("dL_dcolors", c64), ("dL_dcov3D", v64)):
        assert rel_err(g[k].reshape(t.shape), t.grad.numpy()) < 2e-5, k


def test_apply_weights_semantics(oracle):
    """K12: cnt counts blended (pixel, instance) pairs x C; with an all-ones mask weights == cnt / C."""
    P, W, H = 800, 96, 80
    case = make_case(P, W, H, seed=6, s0=0.06)
    sc, cam = case["sc"], case["cam"]
    for C in (1, 3):
        w = np.zeros((P, C), np.float32)
        c = np.zeros(P, np.int32)
        oracle.apply_weights(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], None, cam.world_view_transform,
                             cam.full_proj_transform, cam.camera_center, W, H, case["tfx"], case["tfy"],
                             np.ones((C, H, W), np.float32), w, c)
        assert c.sum() > 0 and np.all(c % C == 0)
        assert np.array_equal(w, np.repeat((c // C)[:, None], C, 1).astype(np.float32))
    with pytest.raises(ValueError):
        oracle.apply_weights(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], None, cam.world_view_transform,
                             cam.full_proj_transform, cam.camera_center, W, H, case["tfx"], case["tfy"],
                             np.ones((4, H, W), np.float32), np.zeros((P, 4), np.float32), np.zeros(P, np.int32))


def test_synth_v2_scene_is_what_it_says(oracle):
    """synth-v2 (gaussianeditor_amd/synth.py::synth_scene_v2), the trained-scene-like workload of round 5: deterministic by
    seed, unit quaternions, disks (one scale a tenth of the other two), bimodal opacity -- and, through the oracle's forward
    from a ring camera INSIDE the dome, every tile non-empty and most Gaussians visible (the uniform cube covers a fifth of
    the tiles)."""
    from gaussianeditor_amd.synth import synth_scene, synth_scene_v2

    P, W, H = 40000, 320, 176
    a, b = synth_scene_v2(P, seed=3), synth_scene_v2(P, seed=3)
    same = lambda x, y: torch.equal(x, y) if isinstance(x, torch.Tensor) else x == y  # noqa: E731
    assert all(same(a[k], b[k]) for k in a) and set(a) == set(synth_scene(16, seed=0))
    assert not torch.equal(a["xyz"], synth_scene_v2(P, seed=4)["xyz"])
    assert torch.allclose(a["rotation"].norm(dim=1), torch.ones(P), atol=1e-5)
    s = a["scaling"]
    assert (s > 0).all() and torch.allclose(s[:, 2], 0.1 * s[:, :2].mean(1), rtol=1e-5)
    op = a["opacity"].reshape(-1)
    assert ((op > 0.84) | (op < 0.31)).all() and 0.6 < float((op > 0.84).float().mean()) < 0.7
    case = make_case(P, W, H, seed=3, view=0, nviews=8)
    case["sc"] = a
    f = oracle_forward(oracle, case)
    r = f["ranges"].reshape(-1, 2)
    assert (r[:, 1] > r[:, 0]).all()  # every tile has a list
    assert (f["radii"] > 0).mean() > 0.6
    cube = oracle_forward(oracle, make_case(P, W, H, seed=3, s0=0.01, view=0, nviews=8))
    rc = cube["ranges"].reshape(-1, 2)
    assert (rc[:, 1] > rc[:, 0]).mean() < 0.5  # (the headline's scene, for contrast)


def test_backward_matches_float64_autograd_on_thin_disks(oracle):
    """The oracle's analytic backward against float64 autograd (as above) on a synth-v2 scene: anisotropic Gaussians whose
    thin axis lies along the surface normal -- needle-like conics wherever a disk is seen edge-on -- under a camera inside
    the scene, with a scale_modifier."""
    from gaussianeditor_amd.synth import synth_scene_v2
    from oracle.torch_ref import render_f64

    P, W, H, sm = 400, 64, 48, 1.7
    case = make_case(P, W, H, seed=11, view=3, nviews=8, bg=(0.2, 0.5, 0.7))
    case["sc"] = synth_scene_v2(P, seed=11)
    sc, cam = case["sc"], case["cam"]
    f = oracle_forward(oracle, case, scale_modifier=sm)
    assert (f["radii"] > 0).sum() > 100
    G = seed_gradient(H, W, 1) * H * W
    g = oracle_backward(oracle, case, f, G, scale_modifier=sm)
    d = torch.float64
    ins = {k: sc[k].to(d).requires_grad_(True) for k in ["xyz", "scaling", "rotation", "opacity", "features"]}
    m2 = torch.zeros(P, 3, dtype=d, requires_grad=True)
    img = render_f64(f, ins["xyz"], m2, ins["opacity"], ins["scaling"], ins["rotation"], ins["features"], None, None,
                     cam.world_view_transform, cam.full_proj_transform, cam.camera_center, case["bg"], W, H, case["tfx"],
                     case["tfy"], sm, 3)
    assert np.abs(img.detach().float().numpy() - f["color"]).max() < 5e-6
    (img * G.to(d)).sum().backward()
    # rows whose view-space x/z or y/z is clamped (forward.cu:82-87) are differentiated with the clamped value as a constant
    # by the reference's analytic backward, not by autograd: compared on the rows inside the cone
    pv = torch.cat([sc["xyz"].to(d), torch.ones(P, 1, dtype=d)], 1) @ cam.world_view_transform.to(d)
    inside = (((pv[:, 0] / pv[:, 2]).abs() <= 1.3 * case["tfx"]) & ((pv[:, 1] / pv[:, 2]).abs() <= 1.3 * case["tfy"])).numpy()
    assert inside.sum() > 50
    for k, t, scale in (("dL_dmeans3D", ins["xyz"], 1.0), ("dL_dmeans2D", m2, 1.0), ("dL_dopacity", ins["opacity"], 1.0),
                        ("dL_dscales", ins["scaling"], 1.0 / sm), ("dL_drotations", ins["rotation"], 1.0), ("dL_dsh", ins["features"], 1.0)):
        a, b = g[k].reshape(P, -1)[inside], (t.grad.numpy() * scale).reshape(P, -1)[inside]
        assert rel_err(a, b) < 3e-5, k
