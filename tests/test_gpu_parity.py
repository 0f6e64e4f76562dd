"""-m gpu: the parity tests proper.  The HIP path (through the C ABI, via the drop-in L1 API)
is compared stage by stage with the CPU oracle on identical seeded inputs.

Bars: integers / indices bit-exact; forward floats bit-exact in practice (asserted <= 1e-6
relative, exactness reported); gradients within 1e-5 of the oracle's (relative to the
tensor's max magnitude -- float atomics re-associate the sums)."""
import numpy as np
import pytest
import torch

from helpers import assert_grads_close, hip_state, make_case, oracle_backward, oracle_forward, rel_err, seed_gradient, settings

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _c():
    from gaussianeditor_amd.diff_gaussian_rasterization import _C

    return _C


def _run_hip_forward(case, colors_precomp=None, cov3D_precomp=None, D=None, scale_modifier=1.0, shs=None):
    sc, cam = case["sc"], case["cam"]
    e = torch.empty(0, device=DEV)
    dev = lambda t: t.to(DEV)  # noqa: E731
    shs_t = sc["features"] if shs is None else shs
    args = (dev(case["bg"]), dev(sc["xyz"]), e if colors_precomp is None else dev(colors_precomp), dev(sc["opacity"]),
            e if cov3D_precomp is not None else dev(sc["scaling"]), e if cov3D_precomp is not None else dev(sc["rotation"]),
            scale_modifier, e if cov3D_precomp is None else dev(cov3D_precomp), dev(cam.world_view_transform),
            dev(cam.full_proj_transform), case["tfx"], case["tfy"], case["H"], case["W"],
            e if colors_precomp is not None else dev(shs_t), case["D"] if D is None else D, dev(cam.camera_center),
            False, False)
    return _c().rasterize_gaussians(*args)


def _compare_forward(O, case, **kw):
    f = oracle_forward(O, case, **kw)
    R, color, depth, radii, geom, binning, img = _run_hip_forward(case, **kw)
    P, W, H = case["sc"]["xyz"].shape[0], case["W"], case["H"]
    cov_inputs = None if kw.get("cov3D_precomp") is not None else (case["sc"]["scaling"], case["sc"]["rotation"],
                                                                   kw.get("scale_modifier", 1.0))
    st = hip_state(P, R, W, H, geom, binning, img, cov_inputs)
    vis = f["radii"] > 0
    # --- integers / indices: bit exact
    assert R == f["num_rendered"]
    assert np.array_equal(radii.cpu().numpy(), f["radii"])
    assert np.array_equal(st["tiles_touched"], f["tiles_touched"])
    assert np.array_equal(st["keys"], f["keys"])
    assert np.array_equal(st["point_list"], f["point_list"])
    assert np.array_equal(st["ranges"], f["ranges"])
    assert np.array_equal(st["n_contrib"], f["n_contrib"])
    if kw.get("colors_precomp") is None:
        assert np.array_equal(st["clamped"][vis], f["clamped"][vis])
    # --- per-Gaussian floats
    exact = {}
    for k in ("means2D", "depths", "conic_opacity", "cov3D"):
        if k not in st:  # (cov3D_precomp given: nothing is computed)
            continue
        a, b = st[k][vis], f[k][vis]
        exact[k] = bool(np.array_equal(a, b))
        assert rel_err(a, b) <= 1e-6, k
    a, b = st["rgb"][vis], f["colors_used"][vis]
    exact["rgb"] = bool(np.array_equal(a, b))
    assert rel_err(a, b) <= 1e-6
    # --- images
    col, dep = color.cpu().numpy(), depth.cpu().numpy()
    exact["final_T"] = bool(np.array_equal(st["final_T"], f["final_T"]))
    exact["color"] = bool(np.array_equal(col, f["color"]))
    exact["depth"] = bool(np.array_equal(dep, f["depth"]))
    assert np.abs(st["final_T"] - f["final_T"]).max() <= 1e-6
    assert np.abs(col - f["color"]).max() <= 1e-5
    assert rel_err(dep, f["depth"]) <= 1e-5
    print("bit-exact:", exact)
    return f, (R, color, depth, radii, geom, binning, img)


@pytest.mark.parametrize("P,W,H,s0,seed", [(10000, 256, 256, 0.03, 1), (3000, 250, 131, 0.05, 2), (500, 64, 64, 0.1, 3),
                                           (20000, 512, 512, 0.02, 4)])
def test_forward_all_stages(oracle, P, W, H, s0, seed):
    case = make_case(P, W, H, seed=seed, s0=s0)
    _compare_forward(oracle, case)


@pytest.mark.parametrize("D,M", [(0, 16), (1, 16), (2, 16), (0, 1), (1, 4), (2, 9)])
def test_forward_sh_degrees(oracle, D, M):
    case = make_case(4000, 200, 120, seed=5, s0=0.04)
    shs = case["sc"]["features"][:, :M, :].contiguous()
    _compare_forward(oracle, case, D=D, shs=shs)


def test_forward_colors_precomp_and_cov3d_precomp(oracle):
    case = make_case(5000, 256, 192, seed=6, s0=0.04)
    g = torch.Generator().manual_seed(7)
    cols = torch.rand(5000, 3, generator=g)
    f0 = oracle_forward(oracle, case)
    cov = torch.from_numpy(f0["cov3D"].copy())
    _compare_forward(oracle, case, colors_precomp=cols)
    _compare_forward(oracle, case, cov3D_precomp=cov)
    _compare_forward(oracle, case, colors_precomp=cols, cov3D_precomp=cov)


def test_forward_scale_modifier_and_culling(oracle):
    # camera inside the cloud: many Gaussians behind the near plane, huge splats near the eye
    case = make_case(3000, 160, 160, seed=8, s0=0.05, scale_xyz=4.0)
    f, _ = _compare_forward(oracle, case, scale_modifier=0.7)
    assert (f["radii"] == 0).any() and (f["radii"] > 0).any()


def _grads_hip(case, G, colors_precomp=None, cov3D_precomp=None, D=None, scale_modifier=1.0):
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    sc = case["sc"]
    rs = settings(case, DEV, D=D, scale_modifier=scale_modifier)
    leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
    xyz, op = leaf(sc["xyz"]), leaf(sc["opacity"])
    m2d = torch.zeros_like(xyz, requires_grad=True)
    kw, leaves = {}, dict(dL_dmeans3D=xyz, dL_dopacity=op, dL_dmeans2D=m2d)
    if colors_precomp is None:
        kw["shs"] = leaves["dL_dsh"] = leaf(sc["features"])
    else:
        kw["colors_precomp"] = leaves["dL_dcolors"] = leaf(colors_precomp)
    if cov3D_precomp is None:
        kw["scales"] = leaves["dL_dscales"] = leaf(sc["scaling"])
        kw["rotations"] = leaves["dL_drotations"] = leaf(sc["rotation"])
    else:
        kw["cov3D_precomp"] = leaves["dL_dcov3D"] = leaf(cov3D_precomp)
    color, radii, depth = GaussianRasterizer(rs)(xyz, m2d, op, **kw)
    (color * G.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    return {k: v.grad.cpu().numpy() for k, v in leaves.items()}


@pytest.mark.parametrize("P,W,H,s0,seed", [(10000, 256, 256, 0.03, 1), (3000, 250, 131, 0.05, 2), (20000, 512, 512, 0.02, 4)])
def test_backward_vs_oracle(oracle, P, W, H, s0, seed):
    case = make_case(P, W, H, seed=seed, s0=s0)
    G = seed_gradient(H, W, seed) * (H * W)  # O(1) pixel gradients
    f = oracle_forward(oracle, case)
    g = oracle_backward(oracle, case, f, G)
    h = _grads_hip(case, G)
    for k, v in h.items():
        e = rel_err(v, g[k].reshape(v.shape))
        print(k, "rel err", e)
        assert e <= 1e-5, k
    assert_grads_close(h, g, tag="product vs oracle")  # (+ the per-row bar: a small row must not be grossly wrong)


def test_backward_precomp_paths(oracle):
    case = make_case(4000, 192, 128, seed=9, s0=0.05)
    G = seed_gradient(128, 192, 9) * (128 * 192)
    cols = torch.rand(4000, 3, generator=torch.Generator().manual_seed(3))
    f0 = oracle_forward(oracle, case)
    cov = torch.from_numpy(f0["cov3D"].copy())
    f = oracle_forward(oracle, case, colors_precomp=cols, cov3D_precomp=cov)
    g = oracle_backward(oracle, case, f, G, colors_precomp=cols, cov3D_precomp=cov)
    h = _grads_hip(case, G, colors_precomp=cols, cov3D_precomp=cov)
    for k, v in h.items():
        assert rel_err(v, g[k].reshape(v.shape)) <= 1e-5, k


def test_backward_run_to_run_spread():
    """Float atomics make the backward order-dependent; the spread must stay far below the parity bar."""
    case = make_case(10000, 256, 256, seed=1, s0=0.03)
    G = seed_gradient(256, 256, 1) * (256 * 256)
    a, b = _grads_hip(case, G), _grads_hip(case, G)
    for k in a:
        assert rel_err(a[k], b[k]) <= 2e-6, k


def test_mark_visible(oracle):
    case = make_case(5000, 64, 64, seed=10, scale_xyz=4.0)
    cam = case["cam"]
    ref = oracle.mark_visible(case["sc"]["xyz"], cam.world_view_transform, cam.full_proj_transform)
    got = _c().mark_visible(case["sc"]["xyz"].to(DEV), cam.world_view_transform.to(DEV), cam.full_proj_transform.to(DEV))
    assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), ref)
    assert ref.any() and not ref.all()


@pytest.mark.parametrize("C,W,H", [(1, 256, 256), (1, 250, 131), (2, 128, 96), (3, 100, 100)])
def test_apply_weights(oracle, C, W, H):
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    P = 6000
    case = make_case(P, W, H, seed=11, s0=0.04)
    sc, cam = case["sc"], case["cam"]
    gen = torch.Generator().manual_seed(12)
    mask = (torch.rand(C, H, W, generator=gen) > 0.5).float()  # 0/1 masks: sums are exact in any order
    w_ref = np.zeros((P, C), np.float32)
    c_ref = np.zeros((P,), np.int32)
    oracle.apply_weights(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], None, cam.world_view_transform,
                         cam.full_proj_transform, cam.camera_center, W, H, case["tfx"], case["tfy"], mask, w_ref, c_ref)
    w = torch.zeros((P, C), device=DEV)
    cnt = torch.zeros((P, 1), dtype=torch.int32, device=DEV)
    rast = GaussianRasterizer(settings(case, DEV, D=0))
    for _ in range(2):  # in-place accumulation over two calls
        rast.apply_weights(sc["xyz"].to(DEV), None, sc["opacity"].to(DEV), None, w, sc["scaling"].to(DEV),
                           sc["rotation"].to(DEV), None, cnt, mask.to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(cnt.cpu().numpy().reshape(-1), 2 * c_ref)
    assert np.array_equal(w.cpu().numpy(), 2 * w_ref)
    assert c_ref.sum() > 0


def test_apply_weights_bad_channels():
    from gaussianeditor_amd import _native
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(100, 32, 32, seed=1)
    sc = case["sc"]
    rast = GaussianRasterizer(settings(case, DEV, D=0))
    with pytest.raises(_native.GsrError):
        rast.apply_weights(sc["xyz"].to(DEV), None, sc["opacity"].to(DEV), None, torch.zeros(100, 4, device=DEV),
                           sc["scaling"].to(DEV), sc["rotation"].to(DEV), None,
                           torch.zeros(100, dtype=torch.int32, device=DEV), torch.zeros(4, 32, 32, device=DEV))


def test_empty_and_fully_culled():
    e = torch.empty(0, device=DEV)
    case = make_case(10, 48, 40, seed=1)
    cam = case["cam"]
    dev = lambda t: t.to(DEV)  # noqa: E731
    # P == 0: zeros everywhere (rasterize_points.cu:72)
    out = _c().rasterize_gaussians(dev(case["bg"]), torch.zeros(0, 3, device=DEV), e, torch.zeros(0, 1, device=DEV),
                                   torch.zeros(0, 3, device=DEV), torch.zeros(0, 4, device=DEV), 1.0, e,
                                   dev(cam.world_view_transform), dev(cam.full_proj_transform), case["tfx"], case["tfy"],
                                   40, 48, torch.zeros(0, 16, 3, device=DEV), 3, dev(cam.camera_center), False, False)
    assert out[0] == 0 and float(out[1].abs().max()) == 0.0 and out[1].shape == (3, 40, 48)
    # all Gaussians behind the camera: R == 0, image == background
    sc = case["sc"]
    behind = sc["xyz"] * 0.1 + cam.camera_center * 2.0  # further out along the eye ray: behind the camera
    R, color, depth, radii, *_ = _c().rasterize_gaussians(
        dev(case["bg"]), dev(behind.contiguous()), e, dev(sc["opacity"]), dev(sc["scaling"]), dev(sc["rotation"]), 1.0,
        e, dev(cam.world_view_transform), dev(cam.full_proj_transform), case["tfx"], case["tfy"], 40, 48,
        dev(sc["features"]), 3, dev(cam.camera_center), False, False)
    assert R == 0 and int(radii.abs().max()) == 0
    assert torch.allclose(color, dev(case["bg"])[:, None, None].expand(3, 40, 48))
    assert float(depth.abs().max()) == 0.0


def test_too_many_instances_is_an_error_not_a_crash():
    """70 000 splats that each cover all 32 400 tiles of a 4K image: 2.27e9 (tile, Gaussian) instances, beyond the
    31-bit index space the reference's `int num_rendered` has too (rasterizer_impl.cu:247-250 would overflow silently).
    The library reports GSR_ERR_TOO_MANY after the preprocessing kernel; nothing is allocated for the instances."""
    from gaussianeditor_amd._native import GsrError

    case = make_case(70000, 3840, 2160, seed=9, s0=40.0, scale_xyz=0.05)
    case["sc"]["opacity"].fill_(0.9)
    with pytest.raises(GsrError, match="31-bit"):
        _run_hip_forward(case)
    # the library is usable afterwards
    small = make_case(500, 64, 64, seed=9, s0=0.1)
    R, color, *_ = _run_hip_forward(small)
    assert R > 0 and bool(torch.isfinite(color).all())


def test_validation_errors():
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(10, 32, 32, seed=1)
    sc = case["sc"]
    rast = GaussianRasterizer(settings(case, DEV))
    x, o = sc["xyz"].to(DEV), sc["opacity"].to(DEV)
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rast(x, x, o, scales=sc["scaling"].to(DEV), rotations=sc["rotation"].to(DEV))
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        rast(x, x, o, shs=sc["features"].to(DEV))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        rast(x.reshape(-1), x, o, shs=sc["features"].to(DEV), scales=sc["scaling"].to(DEV), rotations=sc["rotation"].to(DEV))


def test_headline_workload_full_size(oracle):
    """BASELINE.json's metric configuration itself: synth-v1 1,000,000 Gaussians (SH3) at 1920x1080, view 0.
    Every stage against the oracle at full size (the oracle needs a few seconds on the host cores)."""
    import math

    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer
    from gaussianeditor_amd.synth import ring_cameras, synth_scene

    P, W, H = 1_000_000, 1920, 1080
    sc = synth_scene(P, seed=0, s0=0.01)
    cam = ring_cameras(8, W, H)[0]
    case = dict(sc=sc, cam=cam, W=W, H=H, tfx=math.tan(cam.FoVx / 2), tfy=math.tan(cam.FoVy / 2), bg=sc["bg"], D=3)
    f, (R, color, depth, radii, geom, binning, img) = _compare_forward(oracle, case)
    assert R == f["num_rendered"] > 4_000_000
    # size-independent structure: ranges partition [0, R), lists are depth-sorted inside each tile
    st = hip_state(P, R, W, H, geom, binning, img)
    rl = st["ranges"].astype(np.int64)
    assert (rl[:, 1] - rl[:, 0]).sum() == R
    keys = st["keys"]
    assert np.all(keys[1:] >= keys[:-1])
    # gradients at full size
    G = seed_gradient(H, W, 0)
    g = oracle_backward(oracle, case, f, G)
    leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
    xyz, op, sh, scl, rot = leaf(sc["xyz"]), leaf(sc["opacity"]), leaf(sc["features"]), leaf(sc["scaling"]), leaf(sc["rotation"])
    m2d = torch.zeros_like(xyz, requires_grad=True)
    c2, _, _ = GaussianRasterizer(settings(case, DEV))(xyz, m2d, op, shs=sh, scales=scl, rotations=rot)
    assert torch.equal(c2.detach(), color)  # run-to-run identical forward (work-queue order never affects results)
    (c2 * G.to(DEV)).sum().backward()
    for name, t in (("dL_dmeans3D", xyz), ("dL_dopacity", op), ("dL_dsh", sh), ("dL_dscales", scl),
                    ("dL_drotations", rot), ("dL_dmeans2D", m2d)):
        e = rel_err(t.grad.cpu().numpy(), g[name].reshape(t.shape))
        print(name, "rel err @1M/1080p", e)
        assert e <= 1e-5, name


@pytest.mark.parametrize("P,W,H,s0,scale_xyz", [
    (64, 16, 16, 0.5, 0.3),       # one tile, huge splats: every Gaussian covers the whole image
    (1, 8, 8, 0.2, 0.0),          # a single Gaussian, image smaller than a tile
    (300, 33, 17, 0.6, 0.5),      # ragged edge tiles in both directions, tiles_touched = all tiles
    (5000, 640, 360, 0.4, 1.0),   # long lists: ~all Gaussians in every tile (many 64-entry chunks per tile)
    (257, 48, 48, 0.05, 1.0),     # P just over one 256-Gaussian block
    (4097, 64, 64, 0.05, 1.0),    # P just over one 4096-key sort block
])
def test_edge_geometries(oracle, P, W, H, s0, scale_xyz):
    case = make_case(P, W, H, seed=21, s0=s0, scale_xyz=scale_xyz)
    f, _ = _compare_forward(oracle, case)
    G = seed_gradient(H, W, 21) * (H * W)
    g = oracle_backward(oracle, case, f, G)
    h = _grads_hip(case, G)
    for k, v in h.items():
        assert rel_err(v, g[k].reshape(v.shape)) <= 1e-5, k


def sweep_case(seed):
    """The sweep's configuration for `seed` -> (case, scale_modifier, SH degree); also used by tools/fuzz_parity.py."""
    rng = np.random.default_rng(1000 + seed)
    P = int([1, 2, 63, 65, 255, 256][seed] if seed < 6 else rng.integers(300, 6000))
    W = int(rng.choice([1, 2, 15, 17, 31]) if seed % 3 == 0 else rng.integers(1, 400))
    H = int(rng.choice([1, 3, 16, 47]) if seed % 4 == 1 else rng.integers(1, 300))
    D = int(rng.integers(0, 4))
    case = make_case(P, W, H, seed=100 + seed, s0=float(rng.choice([0.01, 0.05, 0.3])), view=int(rng.integers(0, 4)),
                     sh_degree=D, scale_xyz=float(rng.choice([0.2, 1.0, 2.5])))
    sm = float(rng.choice([0.5, 1.0, 1.7]))
    return case, sm, D


@pytest.mark.parametrize("seed", range(12))
def test_random_configuration_sweep(oracle, seed):
    """Seeded random shapes: odd image sizes down to one pixel, any SH degree, P from 1 to a few thousand, random
    splat size / scene scale / scale_modifier / view.  Forward stage by stage and all six gradients against the oracle."""
    case, sm, D = sweep_case(seed)
    P, W, H = case["sc"]["xyz"].shape[0], case["W"], case["H"]
    f, _ = _compare_forward(oracle, case, scale_modifier=sm)
    G = seed_gradient(H, W, seed) * (H * W)
    g = oracle_backward(oracle, case, f, G, scale_modifier=sm)
    h = _grads_hip(case, G, scale_modifier=sm)
    for k, v in h.items():
        assert rel_err(v, g[k].reshape(v.shape)) <= 1e-5, (k, P, W, H, D, sm)
    assert_grads_close(h, g, tag=f"sweep seed {seed}")  # (+ the per-row bar)


def test_degenerate_inputs(oracle):
    """Zero opacity, zero scale (det of the dilated cov2D stays > 0), coincident Gaussians with identical depth
    (the sort must keep them in index order), opacity > 1 (alpha clamp at 0.99)."""
    case = make_case(2000, 96, 96, seed=22, s0=0.08)
    sc = case["sc"]
    sc["opacity"][:200] = 0.0
    sc["opacity"][200:300] = 1.5
    sc["scaling"][300:400] = 0.0
    sc["xyz"][400:600] = sc["xyz"][400:401]          # 200 coincident centres: identical depth bits and tiles
    sc["scaling"][400:600] = sc["scaling"][400:401]
    sc["rotation"][400:600] = sc["rotation"][400:401]
    f, _ = _compare_forward(oracle, case)
    same = f["keys"][1:] == f["keys"][:-1]
    assert same.sum() > 100 and np.all(f["point_list"][1:][same] > f["point_list"][:-1][same])
    G = seed_gradient(96, 96, 22) * (96 * 96)
    g = oracle_backward(oracle, case, f, G)
    h = _grads_hip(case, G)
    for k, v in h.items():
        assert rel_err(v, g[k].reshape(v.shape)) <= 1e-5, k


@pytest.mark.parametrize("P,W,H,s0,seed", [(10000, 256, 256, 0.03, 1), (5000, 640, 360, 0.4, 21), (20000, 512, 512, 0.02, 4),
                                           (3000, 250, 131, 0.05, 2)])
def test_alpha_tile_bounds_leave_results_unchanged(oracle, P, W, H, s0, seed):
    """set_tile_bounds("alpha") (opt-in): a Gaussian is binned only into the tiles its alpha >= 1/255 level set can reach.
    Images, depths, radii, final_T and traced weights must be bit-identical to the reference rule's (hence to the
    oracle's), gradients within the usual 1e-5; the instance list of every tile must be a subsequence of the reference
    rule's list, and on these scenes (faint and elongated splats included) strictly fewer instances are sorted."""
    import gaussianeditor_amd
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(P, W, H, seed=seed, s0=s0)
    sc = case["sc"]
    sc["opacity"][: P // 10] = 0.003                      # below 1/255: never visible in either rule
    sc["opacity"][P // 10: P // 5] *= 0.05                 # faint: a small level set
    sc["scaling"][P // 5: P // 2, 0] *= 4.0                # elongated: the bounding box is far from the square
    G = seed_gradient(H, W, seed) * (H * W)
    f = oracle_forward(oracle, case)
    out = {}
    try:
        for mode in ("reference", "alpha"):
            gaussianeditor_amd.set_tile_bounds(mode)
            R, color, depth, radii, geom, binning, img = _run_hip_forward(case)
            st = hip_state(P, R, W, H, geom, binning, img)
            grads = _grads_hip(case, G)
            if mode == "reference":
                # four more runs under the same rule: the atomics' run-to-run spread (largest of the ten pairs)
                out["again"] = [_grads_hip(case, G) for _ in range(4)]
            w = torch.zeros(P, 1, device=DEV)
            cnt = torch.zeros(P, 1, dtype=torch.int32, device=DEV)
            mask = (torch.rand(1, H, W, generator=torch.Generator().manual_seed(seed)) > 0.5).float()  # 0/1: exact sums
            GaussianRasterizer(settings(case, DEV, D=0)).apply_weights(
                sc["xyz"].to(DEV), None, sc["opacity"].to(DEV), None, w, sc["scaling"].to(DEV), sc["rotation"].to(DEV), None,
                cnt, mask.to(DEV))
            out[mode] = dict(R=R, color=color.cpu().numpy(), depth=depth.cpu().numpy(), radii=radii.cpu().numpy(), st=st,
                             grads=grads, w=w.cpu().numpy(), cnt=cnt.cpu().numpy())
    finally:
        gaussianeditor_amd.set_tile_bounds("reference")
    a, b = out["reference"], out["alpha"]
    assert a["R"] == f["num_rendered"] and 0 < b["R"] < a["R"]
    print(f"instances {a['R']} -> {b['R']} ({b['R'] / a['R']:.3f})")
    for k in ("color", "depth", "radii", "w", "cnt"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["st"]["final_T"], b["st"]["final_T"])
    assert np.array_equal(b["color"], f["color"]) and np.array_equal(b["radii"], f["radii"])
    for k, v in b["grads"].items():
        runs = [a["grads"][k]] + [g_[k] for g_ in out["again"]]
        r0 = runs[0]
        spread = max(rel_err(runs[i], runs[j]) for i in range(len(runs)) for j in range(i))
        print(k, "alpha vs reference", rel_err(v, r0), "reference run to run", spread)
        # (the two rules group different list entries into the backward's 4-entry reductions: a few run-to-run spreads;
        #  this scene's faint, 4x elongated splats have the worst-conditioned sums of the suite -- the scale / rotation
        #  gradients of the SAME rule differ by 2e-6 .. 2.4e-5 of their maximum from one run to the next.  The bar is the
        #  MEASURED spread, now from five runs (ten pairs) so that a quiet pair cannot understate it; the floor stays 2e-5)
        assert rel_err(v, r0) <= max(2e-5, 8.0 * spread), k
    # per tile: the alpha rule's list is the reference rule's list with some entries removed, order kept
    ra, rb = a["st"]["ranges"], b["st"]["ranges"]
    la, lb = a["st"]["point_list"], b["st"]["point_list"]
    for t in range(ra.shape[0]):
        xa, xb = la[ra[t, 0]:ra[t, 1]], lb[rb[t, 0]:rb[t, 1]]
        assert xb.size <= xa.size and np.array_equal(xa[np.isin(xa, xb)], xb), t


def test_nonfinite_inputs_match_oracle(oracle):
    """NaN / Inf in every input array.  The reference has no guards: a NaN position, scale or rotation fails the
    `det == 0` / rectangle tests and the Gaussian is dropped, a NaN opacity or colour flows into the pixels it covers.
    The HIP path makes the same decisions (float -> int conversions saturate and send NaN to 0 on both sides)."""
    case = make_case(3000, 96, 96, seed=31, s0=0.08)
    sc = case["sc"]
    nan, inf = float("nan"), float("inf")
    sc["xyz"][0:20, 0] = nan
    sc["xyz"][20:40] = inf
    sc["scaling"][40:60] = inf
    sc["scaling"][60:80, 1] = nan
    sc["opacity"][80:100] = nan
    sc["opacity"][100:110] = inf
    sc["rotation"][110:130] = nan
    sc["rotation"][130:140] = 0.0
    sc["features"][140:160] = nan
    sc["features"][160:170, 0] = inf
    f = oracle_forward(oracle, case)
    assert int((f["radii"][:80] > 0).sum()) == 0 and int((f["radii"][80:110] > 0).sum()) == 30  # (what the cases do)
    R, color, depth, radii, geom, binning, img = _run_hip_forward(case)
    st = hip_state(3000, R, 96, 96, geom, binning, img)
    assert R == f["num_rendered"]
    assert np.array_equal(radii.cpu().numpy(), f["radii"])
    assert np.array_equal(st["keys"], f["keys"]) and np.array_equal(st["point_list"], f["point_list"])
    assert np.array_equal(st["ranges"], f["ranges"]) and np.array_equal(st["n_contrib"], f["n_contrib"])
    col = color.cpu().numpy()
    assert np.array_equal(np.isnan(col), np.isnan(f["color"])) and 0 < int(np.isnan(col).sum()) < col.size
    assert np.allclose(col, f["color"], rtol=0, atol=1e-5, equal_nan=True)
    assert np.allclose(st["final_T"], f["final_T"], rtol=0, atol=1e-6, equal_nan=True)
    # the backward runs to completion on the same scene; gradients of the untouched, finite Gaussians match
    G = seed_gradient(96, 96, 31) * (96 * 96)
    g = oracle_backward(oracle, case, f, G)
    h = _grads_hip(case, G)
    for k, v in h.items():
        ref = g[k].reshape(v.shape)
        assert np.array_equal(np.isnan(v), np.isnan(ref)), k
        fin = np.isfinite(ref) & np.isfinite(v)
        scale = max(1.0, float(np.abs(ref[fin]).max())) if fin.any() else 1.0
        assert float(np.abs(v[fin] - ref[fin]).max()) <= 1e-5 * scale, k


def test_non_contiguous_and_sliced_inputs(oracle):
    """The L1 API accepts whatever torch hands it: transposed / strided / sliced tensors are made contiguous
    (rasterize_points.cu:80-91 calls .contiguous() on every input)."""
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(3001, 80, 64, seed=23, s0=0.06)
    sc = case["sc"]
    f = oracle_forward(oracle, case)
    big = torch.randn(3001 + 5, 3, device=DEV)
    big[5:] = sc["xyz"].to(DEV)
    xyz = big[5:]                                      # sliced view with a storage offset (15 floats: not 16-B aligned)
    feats = sc["features"].to(DEV).permute(1, 0, 2).contiguous().permute(1, 0, 2)  # non-contiguous (P,M,3)
    rot = sc["rotation"].to(DEV)[:, [0, 1, 2, 3]].t().contiguous().t()              # column-major (P,4)
    color, radii, depth = GaussianRasterizer(settings(case, DEV))(xyz, torch.zeros_like(xyz), sc["opacity"].to(DEV),
                                                                 shs=feats, scales=sc["scaling"].to(DEV), rotations=rot)
    assert np.array_equal(radii.cpu().numpy(), f["radii"])
    assert np.array_equal(color.cpu().numpy(), f["color"])


# ---------------------------------------------------------------------------------------------------------------------
# simple_knn.distCUDA2 (reference simple-knn/simple_knn.cu, called at
# scene/gaussian_model.py:276 to initialise the scales of a new point cloud)
def _dist2(pts):
    from gaussianeditor_amd.simple_knn._C import distCUDA2

    out = distCUDA2(torch.from_numpy(pts).to(DEV))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("P,kind", [(1, "uniform"), (3, "uniform"), (4, "uniform"), (1000, "uniform"), (1025, "uniform"),
                                    (20000, "uniform"), (20000, "clustered"), (5000, "plane"), (3000, "duplicates")])
def test_knn_bit_exact_vs_oracle(oracle, P, kind):
    rng = np.random.default_rng(P + len(kind))
    pts = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    if kind == "clustered":
        pts = (pts * 0.01 + rng.integers(0, 5, (P, 3)).astype(np.float32)).astype(np.float32)
    elif kind == "plane":
        pts[:, 2] = 0.25  # degenerate bounding box along one axis
    elif kind == "duplicates":
        pts[P // 2:] = pts[:P - P // 2]
    want = oracle.knn_mean_dist2(pts)
    got = _dist2(pts)
    assert got.dtype == np.float32 and got.shape == (P,)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_knn_full_size_vs_kdtree():
    from scipy.spatial import cKDTree

    P = 1_000_000
    rng = np.random.default_rng(7)
    pts = rng.standard_normal((P, 3)).astype(np.float32)
    got = _dist2(pts)
    sel = rng.choice(P, 20000, replace=False)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts[sel].astype(np.float64), k=4)
    want = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(got[sel], want, rtol=3e-5, atol=1e-12)
    assert np.isfinite(got).all() and (got > 0).all()


def test_knn_input_validation():
    from gaussianeditor_amd.simple_knn._C import distCUDA2

    assert distCUDA2(torch.zeros(0, 3, device=DEV)).shape == (0,)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 3))  # host tensor
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 4, device=DEV))
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 3, device=DEV, dtype=torch.float64))
    # non-contiguous input is accepted (made contiguous on the way in)
    base = torch.rand(64, 6, device=DEV)
    a = distCUDA2(base[:, ::2])
    b = distCUDA2(base[:, ::2].contiguous())
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------
# the semantic image of a view from the state of its main render (K6 only)
@pytest.mark.parametrize("P,W,H,s0", [(3000, 128, 96, 0.05), (50000, 512, 512, 0.02)])
def test_aux_render_equals_second_full_render(oracle, P, W, H, s0):
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(P, W, H, seed=4, s0=s0, nviews=3, view=1)
    sc = case["sc"]
    dev = lambda t: t.to(DEV)  # noqa: E731
    rs = settings(case, DEV)
    g = torch.Generator().manual_seed(1)
    aux = torch.rand(P, 3, generator=g)
    aux[torch.rand(P, generator=g) > 0.5] = 0.0  # a mask-like colour table
    leaves = [dev(sc[k]).requires_grad_(True) for k in ("xyz", "features", "opacity", "scaling", "rotation")]
    m3, sh, op, scl, rot = leaves
    m2 = torch.zeros_like(m3, requires_grad=True)
    color, radii, depth, sem = GaussianRasterizer(rs)(m3, m2, op, shs=sh, scales=scl, rotations=rot, aux_colors=dev(aux))
    assert sem.shape == (3, H, W) and not sem.requires_grad
    # (1) bit-identical to a full second render with colors_precomp = aux ...
    with torch.no_grad():
        color2, _, _ = GaussianRasterizer(rs)(dev(sc["xyz"]), torch.zeros_like(m3), dev(sc["opacity"]), colors_precomp=dev(aux),
                                               scales=dev(sc["scaling"]), rotations=dev(sc["rotation"]))
    assert torch.equal(sem, color2)
    # ... and to the oracle's image of that render
    f2 = oracle_forward(oracle, case, colors_precomp=aux)
    assert rel_err(sem.cpu().numpy(), f2["color"]) <= 1e-6
    # (2) the main render and its backward are exactly those of a call without the extension
    G = seed_gradient(H, W, 9).to(DEV) * H * W
    grads = torch.autograd.grad([color], leaves + [m2], grad_outputs=[G])
    leaves_b = [dev(sc[k]).requires_grad_(True) for k in ("xyz", "features", "opacity", "scaling", "rotation")]
    m2b = torch.zeros_like(m3, requires_grad=True)
    color_b, radii_b, depth_b = GaussianRasterizer(rs)(leaves_b[0], m2b, leaves_b[2], shs=leaves_b[1], scales=leaves_b[3],
                                                       rotations=leaves_b[4])
    assert torch.equal(color, color_b) and torch.equal(radii, radii_b) and torch.equal(depth, depth_b)
    grads_b = torch.autograd.grad([color_b], leaves_b + [m2b], grad_outputs=[G])
    for ga, gb in zip(grads, grads_b):
        assert rel_err(ga.cpu().numpy(), gb.cpu().numpy()) <= 1e-5  # (float atomics re-associate between runs)


def test_aux_render_empty_and_validation():
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(64, 64, 64, seed=1, s0=0.05)
    sc = case["sc"]
    dev = lambda t: t.to(DEV)  # noqa: E731
    rs = settings(case, DEV)
    # nothing visible: the auxiliary image is the background, like the main one
    far = dev(sc["xyz"] * 0.1 + case["cam"].camera_center * 2)
    out = GaussianRasterizer(rs)(far, torch.zeros_like(far), dev(sc["opacity"]), shs=dev(sc["features"]),
                                 scales=dev(sc["scaling"]), rotations=dev(sc["rotation"]), aux_colors=torch.ones(64, 3, device=DEV))
    assert torch.equal(out[3], out[0])
    with pytest.raises(RuntimeError):
        GaussianRasterizer(rs)(dev(sc["xyz"]), torch.zeros_like(far), dev(sc["opacity"]), shs=dev(sc["features"]),
                               scales=dev(sc["scaling"]), rotations=dev(sc["rotation"]), aux_colors=torch.ones(64, 4, device=DEV))


# ---------------------------------------------------------------------------------------------------------------------
# fused, row-masked Adam (gsr_adam_step) against the oracle and against torch.optim.Adam
def _adam_groups(P, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    return {k: torch.randn(s, generator=g) for k, s in shapes.items()}, lrs, g


@pytest.mark.parametrize("P", [1, 7, 1001, 40000])
def test_fused_adam_vs_oracle_and_torch(oracle, P):
    from gaussianeditor_amd.optim import FusedMaskedAdam

    init, lrs, gen = _adam_groups(P, P)
    mine = {k: v.clone().to(DEV).requires_grad_(True) for k, v in init.items()}
    ref = {k: v.clone().to(DEV).requires_grad_(True) for k, v in init.items()}
    opt = FusedMaskedAdam([{"params": [p], "lr": lrs[k], "name": k} for k, p in mine.items()], lr=0.0, eps=1e-15)
    topt = torch.optim.Adam([{"params": [p], "lr": lrs[k], "name": k} for k, p in ref.items()], lr=0.0, eps=1e-15)
    orc = {k: [v.numpy().copy(), np.zeros(v.shape, np.float32), np.zeros(v.shape, np.float32)] for k, v in init.items()}
    for step in range(1, 5):
        for k in init:
            g = torch.randn(init[k].shape, generator=gen) * (10.0 if step == 2 else 0.1)
            if step == 3:
                g[: P // 2] = 0.0
            mine[k].grad, ref[k].grad = g.to(DEV), g.to(DEV)
            oracle.adam_step(orc[k][0], g.numpy(), orc[k][1], orc[k][2], lrs[k], step, eps=1e-15)
        opt.step()
        topt.step()
        for k in init:
            st = opt.state[mine[k]]
            # vs the oracle: the same operations in the same order -> bit for bit
            assert np.array_equal(mine[k].detach().cpu().numpy(), orc[k][0]), k
            assert np.array_equal(st["exp_avg"].cpu().numpy(), orc[k][1]) and np.array_equal(st["exp_avg_sq"].cpu().numpy(), orc[k][2])
            # vs torch's own kernels (which contract some a*b+c): 1e-6 of the tensor's magnitude
            for a, b in ((mine[k], ref[k]), (st["exp_avg"], topt.state[ref[k]]["exp_avg"]),
                         (st["exp_avg_sq"], topt.state[ref[k]]["exp_avg_sq"])):
                b = b.detach().cpu().numpy()
                np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=1e-5, atol=1e-6 * float(np.abs(b).max()))
    assert int(opt.state[mine["xyz"]]["step"]) == 4


def test_fused_adam_mask_anchor_and_state_surgery(oracle):
    """Row mask + anchor gradient inside the kernel, and the state layout the reference's densification edits."""
    from gaussianeditor_amd.optim import FusedMaskedAdam

    P = 3000
    init, lrs, gen = _adam_groups(P, 3)
    mask = torch.rand(P, generator=gen) > 0.5
    weight = torch.rand(P, generator=gen)
    anchors = {k: v + 0.05 * torch.randn(v.shape, generator=gen) for k, v in init.items()}
    mine = {k: v.clone().to(DEV).requires_grad_(True) for k, v in init.items()}
    masked_fields = ("xyz", "f_dc", "f_rest", "opacity", "scaling")
    opt = FusedMaskedAdam([{"params": [p], "lr": lrs[k], "name": k, "masked": k in masked_fields} for k, p in mine.items()],
                          lr=0.0, eps=1e-15)
    opt.set_row_mask(mask.to(DEV))
    nsel = int(mask.sum())
    scale = {k: 3.0 * 2.0 / (nsel * (v.numel() // P)) for k, v in init.items()}
    for k, p in mine.items():
        opt.set_anchor(p, anchors[k].to(DEV), scale[k], row_weight=(weight * mask).to(DEV))
    orc = {k: [v.numpy().copy(), np.zeros(v.shape, np.float32), np.zeros(v.shape, np.float32)] for k, v in init.items()}
    for step in range(1, 4):
        for k in init:
            g = torch.randn(init[k].shape, generator=gen)
            mine[k].grad = g.to(DEV)
            oracle.adam_step(orc[k][0], g.numpy(), orc[k][1], orc[k][2], lrs[k], step, eps=1e-15, row_mask=mask.numpy(),
                             masked=k in masked_fields, anchor=anchors[k].numpy(), anchor_scale=scale[k],
                             row_weight=(weight * mask).numpy())
        opt.step()
        for k in init:
            assert np.array_equal(mine[k].detach().cpu().numpy(), orc[k][0]), k
    # masked-out rows received no gradient: after three steps from zero moments they have not moved
    out = ~mask.numpy()
    assert np.array_equal(mine["xyz"].detach().cpu().numpy()[out], init["xyz"].numpy()[out])
    # densification-style surgery (gaussian_model.py:591-641): cut the state tensors and keep stepping
    keep = torch.arange(P, device=DEV) % 3 != 0
    for group in opt.param_groups:
        p = group["params"][0]
        st = opt.state.pop(p)
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep].contiguous(), st["exp_avg_sq"][keep].contiguous()
        q = torch.nn.Parameter(p.detach()[keep].contiguous().requires_grad_(True))
        group["params"][0] = q
        opt.state[q] = st
        q.grad = torch.ones_like(q)
    opt.set_row_mask(None)
    opt._anchors.clear()
    opt.step()
    assert int(opt.state[opt.param_groups[0]["params"][0]]["step"]) == 4
    assert all(torch.isfinite(g["params"][0]).all() for g in opt.param_groups)


# ---------------------------------------------------------------------------------------------------------------------
# paths added with the scheduling / readback work
def test_4k_image_many_tiles(oracle):
    """3840x2160: 32 400 tiles -> 15 tile bits (two tile passes of 8 + 7 bits), 47-bit reference key."""
    from gaussianeditor_amd import _native

    assert _native.lib().gsr_sort_key_bits(3840, 2160) == 47
    case = make_case(30000, 3840, 2160, seed=31, s0=0.01)
    f, _ = _compare_forward(oracle, case)
    G = seed_gradient(2160, 3840, 31) * (2160 * 3840)
    g = oracle_backward(oracle, case, f, G)
    h = _grads_hip(case, G)
    for k, v in h.items():
        assert rel_err(v, g[k].reshape(v.shape)) <= 1e-5, k


@pytest.mark.parametrize("kind", ["flat", "shallow", "deep"])
def test_depth_key_ranges(oracle, kind):
    """The depth sort runs ceil(bits/8) passes over the bits in which the smallest and the largest depth key differ
    (2, 3 or 4): one scene per pass count, order checked against the oracle's full 64-bit-key sort."""
    P, W, H = 6000, 160, 120
    case = make_case(P, W, H, seed=33, s0=0.05, nviews=1)
    g = torch.Generator().manual_seed(5)
    cam = case["cam"]
    fwd = -cam.camera_center / cam.camera_center.norm()  # the camera looks at the origin
    side = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]))
    side = side / side.norm()
    up = torch.linalg.cross(side, fwd)
    u, v = torch.rand(P, generator=g) - 0.5, torch.rand(P, generator=g) - 0.5
    if kind == "flat":      # one plane facing the camera: depths equal up to a few ulps -> 2 passes
        d = torch.zeros(P)
    elif kind == "shallow":  # depth 3.9 .. 4.1
        d = (torch.rand(P, generator=g) - 0.5) * 0.2
    else:                   # depth 0.25 .. 60: the keys differ in the exponent bits -> 4 passes
        d = torch.exp(torch.rand(P, generator=g) * 5.5 - 1.4) - 4.0
    spread = (4.0 + d) * 0.4
    case["sc"]["xyz"] = (u * spread)[:, None] * side[None] + (v * spread)[:, None] * up[None] + d[:, None] * fwd[None]
    case["sc"]["xyz"] = case["sc"]["xyz"].contiguous()
    f, _ = _compare_forward(oracle, case)
    depth_bits = np.unique(f["keys"] & np.uint64(0xffffffff))
    span = int(depth_bits.max()) ^ int(depth_bits.min())
    assert {"flat": span < (1 << 16), "shallow": (1 << 16) <= span < (1 << 24), "deep": span >= (1 << 24)}[kind], hex(span)


def test_backward_twice_and_two_streams(oracle):
    """(a) the saved state of a forward supports any number of backward calls (work queues re-arm themselves);
    (b) renders issued on two streams from two host threads do not share any hidden state."""
    import threading

    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(20000, 320, 200, seed=35, s0=0.03)
    sc = case["sc"]
    dev = lambda t: t.to(DEV)  # noqa: E731
    rs = settings(case, DEV)
    leaves = [dev(sc[k]).requires_grad_(True) for k in ("xyz", "features", "opacity", "scaling", "rotation")]
    m2 = torch.zeros_like(leaves[0], requires_grad=True)
    color, radii, depth = GaussianRasterizer(rs)(leaves[0], m2, leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4])
    G = seed_gradient(200, 320, 35).to(DEV) * (200 * 320)
    g1 = torch.autograd.grad([color], leaves, grad_outputs=[G], retain_graph=True)
    g2 = torch.autograd.grad([color], leaves, grad_outputs=[G], retain_graph=True)
    g3 = torch.autograd.grad([color], leaves, grad_outputs=[2 * G])
    for a, b, c in zip(g1, g2, g3):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5
        assert rel_err(2 * a.cpu().numpy(), c.cpu().numpy()) <= 1e-5

    cases = [make_case(15000 + 1000 * i, 256, 192, seed=40 + i, s0=0.03) for i in range(2)]
    expect = [oracle_forward(oracle, c)["color"] for c in cases]
    results = [None, None]

    def work(i):
        st = torch.cuda.Stream(device=DEV)
        with torch.cuda.stream(st):
            for _ in range(5):
                out = _run_hip_forward(cases[i])
            st.synchronize()
        results[i] = out[1].cpu().numpy()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for r, e in zip(results, expect):
        assert rel_err(r, e) <= 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# multi-GPU exchange support: colour gradients instead of SH gradients (gsr_preprocess_backward_rgb, gsr_sh_grad_compose)
@pytest.mark.parametrize("D", [0, 1, 2, 3])
def test_sh_grad_rebuilt_from_colour_gradients(oracle, D):
    from gaussianeditor_amd.multiview import GradBucket, allreduce_view_grads, render_view_grads

    P, W, H, views = 6000, 200, 150, 3
    direct_sum, rgbs, cams = None, [], []
    m3 = None
    for v in range(views):
        case = make_case(P, W, H, seed=50, s0=0.04, view=v, nviews=views)
        sc = case["sc"]
        rs = settings(case, DEV, D=D)
        G = (seed_gradient(H, W, 60 + v) * H * W).to(DEV)
        args = [sc[k].to(DEV) for k in ("xyz", "opacity", "features", "scaling", "rotation")]
        m3 = args[0]
        # direct: the backward's own dL_dsh
        b_direct = GradBucket(P, 16, DEV, sh_exchange="direct")
        _, _, _, g_direct = render_view_grads(rs, *args, G, b_direct)
        direct_sum = g_direct["sh"].clone() if direct_sum is None else direct_sum + g_direct["sh"]
        # rgb mode: 3 floats per Gaussian, everything else identical
        b_rgb = GradBucket(P, 16, DEV, sh_exchange="rgb")
        _, _, _, g_rgb = render_view_grads(rs, *args, G, b_rgb)
        assert g_rgb["sh"] is None
        for k in ("means3D", "opacities", "scales", "rotations", "means2D"):
            assert rel_err(g_rgb[k].cpu().numpy(), g_direct[k].cpu().numpy()) <= 1e-5, k
        assert allreduce_view_grads(b_rgb, None) == "local"  # one process: the rebuilt gradient of this view alone
        # (two separate backward runs: the blend's float atomics re-associate, so across RUNS the bar is 1e-5; the
        #  composition itself is checked bit for bit against the oracle below and in tests/test_cpu_multiview.py)
        assert rel_err(b_rgb.views["sh"].cpu().numpy(), g_direct["sh"].cpu().numpy()) <= 1e-5
        rgbs.append(b_rgb.rgb.clone())
        cams.append(rs.campos.reshape(3).clone())
    # the batch: rebuilt sum over the views == the views' SH gradients accumulated one after the other
    from gaussianeditor_amd.diff_gaussian_rasterization import _C

    rebuilt = _C.sh_grad_compose(m3, torch.stack(cams), torch.stack(rgbs), D, 16)
    assert rel_err(rebuilt.cpu().numpy(), direct_sum.cpu().numpy()) <= 1e-5
    ref = oracle.sh_grad_compose(m3.cpu().numpy(), torch.stack(cams).cpu().numpy(), torch.stack(rgbs).cpu().numpy(), D, 16)
    assert np.array_equal(rebuilt.cpu().numpy(), ref)


@pytest.mark.parametrize("D,s0,P", [(3, 0.01, 20000), (1, 0.05, 20000), (2, 0.02, 1023), (3, 0.02, 5121)])
def test_touched_rows_exchange_kernels(oracle, D, s0, P):
    """The multi-GPU exchange in its touched-rows form, run for three views in ONE process: every view packs its
    non-zero rows into a message (gsr_view_message_plan / _pack), gsr_view_messages_accumulate adds the messages per
    Gaussian in view order.  The sums equal torch's sequential accumulation of the dense per-view gradients bit for
    bit, the rebuilt SH gradient equals gsr_sh_grad_compose over the same views bit for bit (and the oracle's)."""
    from gaussianeditor_amd.diff_gaussian_rasterization import _C
    from gaussianeditor_amd.multiview import _ROW_SEGS, GradBucket, render_view_grads

    W, H, views, M = 200, 150, 3, 16
    buckets, plans, dense_sum, rgbs, cams, m3 = [], [], None, [], [], None
    for v in range(views):
        case = make_case(P, W, H, seed=70, s0=s0, view=v, nviews=views)
        sc = case["sc"]
        rs = settings(case, DEV, D=D)
        G = (seed_gradient(H, W, 80 + v) * H * W).to(DEV)
        args = [sc[k].to(DEV) for k in ("xyz", "opacity", "features", "scaling", "rotation")]
        m3 = args[0]
        b = GradBucket(P, M, DEV, sh_exchange="rgb")
        render_view_grads(rs, *args, G, b)
        grads5 = [b.views[name] for name in _ROW_SEGS]
        plan, count = _C.view_message_plan(grads5, b.rgb)
        any_nz = torch.cat([g.reshape(P, -1) for g in grads5] + [b.rgb], dim=1).ne(0).any(dim=1)
        assert count == int(any_nz.sum()) and 0 < count <= P
        buckets.append(b)
        plans.append((plan, count, any_nz.nonzero().view(-1)))
        cur = [g.clone() for g in grads5]
        dense_sum = cur if dense_sum is None else [a + c for a, c in zip(dense_sum, cur)]  # (0 + g0) + g1 + g2
        rgbs.append(b.rgb.clone())
        cams.append(rs.campos.reshape(3).clone())
    cap = max(c for _, c, _ in plans) + 5  # (padding rows beyond every count)
    words, nb = _C.view_message_words(P, cap), (P + 1023) // 1024
    assert words == 4 + nb + 18 * cap
    messages = torch.full((views, words + 7), float("nan"), device=DEV)[:, :words]  # rows 'stride' apart, stride > words
    for v, (b, (plan, count, want)) in enumerate(zip(buckets, plans)):
        _C.view_message_pack(plan, [b.views[name] for name in _ROW_SEGS], b.rgb, cams[v], cap, messages[v])
        msg = messages[v]
        assert torch.equal(msg[:3], cams[v]) and int(msg.view(torch.int32)[3]) == count
        boff = msg.view(torch.int32)[4:4 + nb].long()
        assert torch.equal(boff, torch.searchsorted(want, torch.arange(nb, device=DEV) * 1024))
        off = 4 + nb
        assert torch.equal(msg.view(torch.int32)[off:off + count].long(), want)
        off += cap
        for name, k in zip(_ROW_SEGS + ("rgb",), (3, 3, 4, 3, 1, 3)):
            src = b.rgb if name == "rgb" else b.views[name]
            assert torch.equal(msg[off:off + k * cap].view(cap, k)[:count], src.reshape(P, k)[want]), name
            off += k * cap
    dense = [torch.full((P, k), float("nan"), device=DEV) for k in (3, 3, 4, 3, 1)] + [torch.full((P, M, 3), float("nan"), device=DEV)]
    _C.view_messages_accumulate(messages, P, cap, D, M, m3, dense)
    for got, ref in zip(dense[:5], dense_sum):
        assert torch.equal(got, ref.reshape(got.shape))
    composed = _C.sh_grad_compose(m3, torch.stack(cams), torch.stack(rgbs), D, M)
    assert torch.equal(dense[5], composed)
    ref = oracle.sh_grad_compose(m3.cpu().numpy(), torch.stack(cams).cpu().numpy(), torch.stack(rgbs).cpu().numpy(), D, M)
    assert np.array_equal(dense[5].cpu().numpy(), ref)
    # more views than one LDS batch holds (8): the three messages repeated, eleven in all
    reps = [v % views for v in range(11)]
    many = messages[reps].contiguous()
    dense11 = [torch.full((P, k), float("nan"), device=DEV) for k in (3, 3, 4, 3, 1)] + [torch.full((P, M, 3), float("nan"), device=DEV)]
    _C.view_messages_accumulate(many, P, cap, D, M, m3, dense11)
    seq = None
    for v in reps:
        cur = [buckets[v].views[name].reshape(P, -1) for name in _ROW_SEGS]
        seq = [c.clone() for c in cur] if seq is None else [a + c for a, c in zip(seq, cur)]
    for got, ref11 in zip(dense11[:5], seq):
        assert torch.equal(got, ref11.reshape(got.shape))
    assert torch.equal(dense11[5], _C.sh_grad_compose(m3, torch.stack([cams[v] for v in reps]),
                                                      torch.stack([rgbs[v] for v in reps]), D, M))
    # only the rows some view sent (gsr_view_messages_accumulate_rows): those equal the dense sums bit for bit and are
    # marked valid; nothing else is written; and the fused Adam that takes the mask (gsr_adam_step_rows: invalid rows are
    # zero gradients that are never read -- here they are NaN) equals the step on the dense, zero-filled gradients
    sparse = [torch.full((P, k), float("nan"), device=DEV) for k in (3, 3, 4, 3, 1)] + [torch.full((P, M, 3), float("nan"), device=DEV)]
    valid = torch.full((P,), 7, dtype=torch.uint8, device=DEV)
    _C.view_messages_accumulate(messages, P, cap, D, M, m3, sparse, row_valid=valid)
    sent = torch.zeros(P, dtype=torch.bool, device=DEV)
    for v in range(views):
        sent[messages[v].view(torch.int32)[4 + nb:4 + nb + int(messages[v].view(torch.int32)[3])].long()] = True
    assert torch.equal(valid.bool(), sent) and int(valid.max()) == 1
    for got, ref in zip(sparse, dense):
        assert torch.equal(got[sent], ref[sent]) and bool(got[~sent].isnan().all()) and not bool(ref[~sent].any())
    one = [torch.full((P, k), float("nan"), device=DEV) for k in (3, 3, 4, 3, 1)] + [None]  # one view: fewer rows are valid
    valid1 = torch.empty(P, dtype=torch.uint8, device=DEV)
    _C.view_messages_accumulate(messages[:1], P, cap, D, M, m3, one, row_valid=valid1)
    sent1 = torch.zeros(P, dtype=torch.bool, device=DEV)
    sent1[messages[0].view(torch.int32)[4 + nb:4 + nb + int(messages[0].view(torch.int32)[3])].long()] = True
    assert torch.equal(valid1.bool(), sent1)
    for got, name in zip(one[:5], _ROW_SEGS):
        ref1 = buckets[0].views[name].reshape(got.shape)
        assert torch.equal(got[sent1], ref1[sent1]) and bool(got[~sent1].isnan().all()) and not bool(ref1[~sent1].any()), name
    from gaussianeditor_amd.optim import FusedMaskedAdam

    gen = torch.Generator(device=DEV).manual_seed(3)
    names = ("xyz", "scaling", "rotation", "m2", "opacity", "sh")
    pa = {n: torch.randn(t.shape, device=DEV, generator=gen) for n, t in zip(names, dense)}
    pb = {n: t.clone() for n, t in pa.items()}
    anchor = {n: t + 0.01 for n, t in pa.items()}
    row_mask = torch.rand(P, device=DEV, generator=gen) > 0.3
    opts = []
    for params, grads, gv in ((pa, dense, None), (pb, sparse, valid)):
        ps = {n: torch.nn.Parameter(t) for n, t in params.items()}
        opt = FusedMaskedAdam([{"params": [ps[n]], "lr": 1e-3 * (i + 1), "name": n, "masked": n in ("xyz", "sh")}
                               for i, n in enumerate(names)], lr=0.0, eps=1e-15)
        opt.set_row_mask(row_mask)
        opt.set_anchor(ps["scaling"], anchor["scaling"], 0.5)
        opt.set_grad_valid(gv)
        for step in range(2):
            for n, gt in zip(names, grads):
                ps[n].grad = gt.reshape(ps[n].shape)
            opt.step()
        opts.append(ps)
    for n in names:
        assert torch.equal(opts[0][n].detach(), opts[1][n].detach()), n
        assert bool(torch.isfinite(opts[1][n]).all())
    # without an SH target only the five dense segments are written; a single message is a valid batch
    dense2 = [torch.full((P, k), float("nan"), device=DEV) for k in (3, 3, 4, 3, 1)] + [None]
    _C.view_messages_accumulate(messages[:1], P, cap, D, M, m3, dense2)
    for got, name in zip(dense2[:5], _ROW_SEGS):
        assert torch.equal(got, buckets[0].views[name].reshape(got.shape)), name


# ---------------------------------------------------------------------------------------------------------------------
# prune = one compaction for all tensors (gsr_compact_plan / gsr_compact_apply)
@pytest.mark.parametrize("P,frac", [(1, 1.0), (5, 0.0), (1023, 0.5), (1025, 0.5), (100000, 0.9), (100000, 0.01)])
def test_compact_rows_equals_boolean_indexing(oracle, P, frac):
    from gaussianeditor_amd.densify import compact_rows

    g = torch.Generator().manual_seed(P)
    keep = torch.rand(P, generator=g) < frac
    ts = [torch.randn(P, 3, generator=g), torch.randn(P, 15, 3, generator=g), torch.randn(P, 1, generator=g),
          torch.randn(P, 4, generator=g), torch.arange(P, dtype=torch.int64), keep.clone(), torch.randn(P, generator=g),
          torch.randint(0, 255, (P, 3), generator=g, dtype=torch.uint8)]  # incl. 1-byte and 3-byte rows
    out = compact_rows([t.to(DEV) for t in ts], keep.to(DEV))
    ref = oracle.compact_rows([t.numpy() for t in ts], keep.numpy())
    for o, r, t in zip(out, ref, ts):
        assert o.dtype == t.dtype and tuple(o.shape[1:]) == tuple(t.shape[1:])
        assert np.array_equal(o.cpu().numpy(), r)
        assert torch.equal(o.cpu(), t[keep])


def test_prune_optimizer_like_reference():
    """prune_optimizer == GaussianModel._prune_optimizer (gaussian_model.py:568-591) on an Adam with state."""
    from gaussianeditor_amd.densify import prune_optimizer
    from gaussianeditor_amd.optim import FusedMaskedAdam

    P = 5000
    init, lrs, gen = _adam_groups(P, 9)

    def build():
        params = {k: v.clone().to(DEV).requires_grad_(True) for k, v in init.items()}
        opt = FusedMaskedAdam([{"params": [p], "lr": lrs[k], "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
        g2 = torch.Generator().manual_seed(1)
        for p in params.values():
            p.grad = torch.randn(p.shape, generator=g2).to(DEV)
        opt.step()
        return opt

    keep = (torch.rand(P, generator=gen) > 0.25).to(DEV)
    a, b = build(), build()
    new = prune_optimizer(a, keep)
    for group in b.param_groups:  # the reference's loop
        st = b.state.get(group["params"][0], None)
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep], st["exp_avg_sq"][keep]
        del b.state[group["params"][0]]
        group["params"][0] = torch.nn.Parameter(group["params"][0][keep].requires_grad_(True))
        b.state[group["params"][0]] = st
    for ga, gb in zip(a.param_groups, b.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert new[ga["name"]] is pa and pa.requires_grad and torch.equal(pa, pb)
        assert torch.equal(a.state[pa]["exp_avg"], b.state[pb]["exp_avg"])
        assert torch.equal(a.state[pa]["exp_avg_sq"], b.state[pb]["exp_avg_sq"])
    for p in (g["params"][0] for g in a.param_groups):
        p.grad = torch.ones_like(p)
    a.step()  # the pruned optimizer keeps working


def test_c_abi_host_without_torch(tmp_path):
    """examples/c_abi_render.cpp links libgsr_hip.so from plain C++ (hipMalloc'd buffers, no torch, no Python) and must
    produce the very image the drop-in binding renders -- the boundary carries no hidden torch state."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_abi_render")
    if not os.path.exists(exe):
        pytest.fail("examples/c_abi_render is not built (run __graft_entry__.build())")
    case = make_case(20000, 320, 200, seed=11, s0=0.03)
    sc, cam = case["sc"], case["cam"]
    P, M = sc["features"].shape[0], sc["features"].shape[1]
    scene = tmp_path / "scene.bin"
    with open(scene, "wb") as f:
        np.array([P, case["D"], M, case["W"], case["H"]], np.int32).tofile(f)
        np.array([case["tfx"], case["tfy"], 1.0], np.float32).tofile(f)
        for t in (case["bg"], sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"],
                  cam.world_view_transform, cam.full_proj_transform, cam.camera_center):
            t.contiguous().numpy().astype(np.float32).tofile(f)
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(scene), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    R, color, depth, radii, *_ = _run_hip_forward(case)
    W, H = case["W"], case["H"]
    with open(out, "rb") as f:
        R_c = int(np.fromfile(f, np.int64, 1)[0])
        color_c = np.fromfile(f, np.float32, 3 * H * W).reshape(3, H, W)
        depth_c = np.fromfile(f, np.float32, H * W).reshape(1, H, W)
        radii_c = np.fromfile(f, np.int32, P)
    assert R_c == R and R > 0
    assert np.array_equal(radii_c, radii.cpu().numpy())
    assert np.array_equal(color_c, color.cpu().numpy())
    assert np.array_equal(depth_c, depth.cpu().numpy())
