"""-m gpu, round 5: the fuzz sweep's three outliers pinned against the reference's own backward and a float64 restatement;
degenerate depth distributions through the depth sort; the arena's failed growth; the synth-v2 workload's parity case."""
import numpy as np
import pytest
import torch

from helpers import hip_state, make_case, oracle_backward, oracle_forward, require_ref, seed_gradient

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRADS = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")


def _np(t):
    return t.detach().cpu().numpy()


def _sweep_case(seed):
    """The configuration tests/test_gpu_parity.py::test_random_configuration_sweep draws for `seed` (same generator calls)."""
    rng = np.random.default_rng(1000 + seed)
    P = int([1, 2, 63, 65, 255, 256][seed] if seed < 6 else rng.integers(300, 6000))
    W = int(rng.choice([1, 2, 15, 17, 31]) if seed % 3 == 0 else rng.integers(1, 400))
    H = int(rng.choice([1, 3, 16, 47]) if seed % 4 == 1 else rng.integers(1, 300))
    D = int(rng.integers(0, 4))
    case = make_case(P, W, H, seed=100 + seed, s0=float(rng.choice([0.01, 0.05, 0.3])), view=int(rng.integers(0, 4)),
                     sh_degree=D, scale_xyz=float(rng.choice([0.2, 1.0, 2.5])))
    return case, float(rng.choice([0.5, 1.0, 1.7])), D


# tools/fuzz_parity.py, seeds 12 .. 3 188 (round 4): every integer and forward float bit-exact in all 3 177 configurations;
# in these three ONE gradient tensor sat 18-29 % over the suite's 1e-5 bar on a single large Gaussian.
FUZZ_OUTLIERS = (2977, 3019, 3186)


@pytest.mark.parametrize("seed", FUZZ_OUTLIERS)
def test_fuzz_outliers_are_no_further_from_float64_than_the_reference(oracle, seed):
    """The sweep's outliers, four ways (see `four_way`)."""
    case, sm, D = _sweep_case(seed)
    four_way(oracle, case, sm, D, seed)


class RestatementMismatch(AssertionError):
    """four_way: the float64 restatement rendered another image than the binary32 forward (it cannot judge the gradients)."""


def four_way(oracle, case, sm, D, seed, image_tol=1e-5):
    """Four-way on the sweep's outliers: reference(no contraction) backward / oracle / product / float64 autograd.

    The product's gradients must be no further from the float64 restatement than the reference's OWN backward
    (backward.cu:417-556 compiled for gfx950) and the oracle are -- the needle test's pattern: a Gaussian that covers
    thousands of pixels sums terms whose binary32 total depends on the summation order (the oracle adds pixel after pixel,
    the kernels reduce waves and tiles, the reference's atomics any order).  Two conventions of the reference's analytic
    backward that autograd does not share are taken out first: dL_dscales lacks the factor scale_modifier
    (backward.cu:computeCov3D differentiates `mod * scale` by the product), and a Gaussian whose view-space x/z or y/z is
    clamped to 1.3 tan(fov) (forward.cu:82-87) is differentiated with the clamped value as a constant -- those rows are
    compared with the reference and the oracle only."""
    import test_gpu_parity as tp
    from oracle.torch_ref import render_f64

    sc, cam, W, H = case["sc"], case["cam"], case["W"], case["H"]
    P = sc["xyz"].shape[0]
    f, _ = tp._compare_forward(oracle, case, scale_modifier=sm)  # every stage of the forward, bit for bit
    G = seed_gradient(H, W, seed) * (H * W)
    go = oracle_backward(oracle, case, f, G, scale_modifier=sm)
    gp = tp._grads_hip(case, G, scale_modifier=sm)
    R_ = require_ref("nofma").Reference("nofma", DEV)
    R_.forward(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], None, None, cam.world_view_transform,
               cam.full_proj_transform, cam.camera_center, case["bg"], W, H, case["tfx"], case["tfy"], sm, D)
    gr = {k: _np(v) for k, v in R_.backward(G).items()}
    # float64 autograd (CPU, tile by tile)
    d = torch.float64
    ins = {k: sc[k].to(d).requires_grad_(True) for k in ("xyz", "scaling", "rotation", "opacity", "features")}
    m2 = torch.zeros(P, 3, dtype=d, requires_grad=True)
    img = render_f64(f, ins["xyz"], m2, ins["opacity"], ins["scaling"], ins["rotation"], ins["features"], None, None,
                     cam.world_view_transform, cam.full_proj_transform, cam.camera_center, case["bg"], W, H, case["tfx"],
                     case["tfy"], sm, D, dL_dimage=G)
    # (the float64 restatement renders the same image; scenes whose pixels sum thousands of translucent layers -- tools/fuzz_v2.py
    #  -- differ from the binary32 forward by more than the suite's cases do: the caller then widens the sanity bound)
    worst_px = float(np.abs(img.float().numpy() - f["color"]).max())
    if not worst_px < image_tol:  # (a pixel where float64 decides an alpha / transmittance threshold the other way: no judge)
        raise RestatementMismatch(f"the float64 restatement's image differs from the binary32 forward's by {worst_px:.2e}")
    g64 = {"dL_dmeans3D": ins["xyz"].grad, "dL_dmeans2D": m2.grad, "dL_dopacity": ins["opacity"].grad,
           "dL_dscales": ins["scaling"].grad / sm, "dL_drotations": ins["rotation"].grad, "dL_dsh": ins["features"].grad}
    pv = torch.cat([sc["xyz"].to(d), torch.ones(P, 1, dtype=d)], 1) @ cam.world_view_transform.to(d)
    inside = ((pv[:, 0] / pv[:, 2]).abs() <= 1.3 * case["tfx"]) & ((pv[:, 1] / pv[:, 2]).abs() <= 1.3 * case["tfy"])
    inside = inside.numpy()
    print(f"seed {seed}: P={P} {W}x{H} D={D} scale_modifier={sm} R={f['num_rendered']}; rows inside the clamp cone {int(inside.sum())}")
    for k in GRADS:
        a = gp[k].reshape(P, -1).astype(np.float64)
        o = go[k].reshape(P, -1).astype(np.float64)
        r = gr[k].reshape(P, -1).astype(np.float64)
        t = (torch.zeros_like(ins["xyz"]) if g64[k] is None else g64[k]).numpy().reshape(P, -1)
        so = max(np.abs(o).max(), 1e-30)
        e_po, e_ro = np.abs(a - o).max() / so, np.abs(r - o).max() / so
        st = max(np.abs(t[inside]).max(), 1e-30)
        e_pf, e_rf, e_of = (np.abs(x[inside] - t[inside]).max() / st for x in (a, r, o))
        out = ~inside
        e_po_c, e_ro_c = ((np.abs(x[out] - o[out]).max() / so if out.any() else 0.0) for x in (a, r))
        print(f"  {k:14s} vs float64 (rows inside the cone): product {e_pf:.2e} reference {e_rf:.2e} oracle {e_of:.2e} | vs oracle "
              f"(all rows): product {e_po:.2e} reference {e_ro:.2e}; (clamped rows): product {e_po_c:.2e} reference {e_ro_c:.2e}")
        assert np.isfinite(a).all(), k
        # THE criterion: against the float64 value all three carry binary32 summation noise of the same size (on seed 2977 --
        # a 360 x 3 image under splats that cover all of it -- 2.5e-5 of the tensor's maximum for dL_dmeans3D, each of them);
        # the product must be no further from it than the reference's own backward and the oracle, with the needle test's slack
        assert e_pf <= max(1e-5, 3.0 * max(e_rf, e_of)), k
        # rows float64 cannot judge (outside the clamp cone): against the oracle, next to the reference's distance from it
        assert e_po_c <= max(2e-5, 5.0 * e_ro_c), k
        assert e_po <= 1e-4, k  # (and nothing is wildly off anywhere)


def _depth_wall_case(P, W, H, kind, seed=3):
    """Degenerate depth distributions for the depth sort (rasterizer_impl.cu:229-271 sorts 64-bit keys whatever they are),
    seen by a camera that looks down the world's z axis, so that a point's view-space depth is its z + 4 exactly:
    `wall`: every Gaussian at EXACTLY the same depth (all depth keys equal: the order is the index order);
    `two`: such a wall and a handful of Gaussians far behind it (the key range spans octaves, the wall shares one value);
    `steps`: sixteen walls."""
    import math

    from gaussianeditor_amd.synth import look_at_camera

    case = make_case(P, W, H, seed=seed, s0=0.02)
    cam = look_at_camera([0.0, 0.0, -4.0], [0.0, 0.0, 0.0], W, H)
    case.update(cam=cam, tfx=math.tan(cam.FoVx / 2), tfy=math.tan(cam.FoVy / 2))
    xyz = case["sc"]["xyz"]
    g = torch.Generator().manual_seed(seed)
    if kind == "wall":
        xyz[:, 2] = 0.25
    elif kind == "two":
        xyz[:, 2] = 0.25
        far = torch.randperm(P, generator=g)[:32]
        xyz[far, 2] = 30.0 + 50.0 * torch.rand(32, generator=g)
        xyz[far, :2] *= 4.0
    else:
        xyz[:, 2] = -0.5 + 0.0625 * torch.randint(0, 16, (P,), generator=g).float()
    return case


@pytest.mark.parametrize("kind", ["wall", "two", "steps"])
def test_depth_sort_with_walls_of_equal_depth(oracle, kind):
    """Sorted keys, point list, ranges, n_contrib and the image bit-identical to the oracle when (nearly) all depth keys are
    equal -- ties must come out in ascending Gaussian index, as the reference's stable sort leaves them -- and gradients
    within the bar."""
    import test_gpu_parity as tp
    from gaussianeditor_amd.diff_gaussian_rasterization import _C

    P, W, H = 60000, 640, 360
    case = _depth_wall_case(P, W, H, kind)
    sc, cam = case["sc"], case["cam"]
    f = oracle_forward(oracle, case)
    e = torch.empty(0, device=DEV)
    d = lambda t: t.to(DEV)  # noqa: E731
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        d(case["bg"]), d(sc["xyz"]), e, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), 1.0, e,
        d(cam.world_view_transform), d(cam.full_proj_transform), case["tfx"], case["tfy"], H, W, d(sc["features"]), case["D"],
        d(cam.camera_center), False, False)
    st = hip_state(P, R, W, H, geom, binning, img)
    vis = f["radii"] > 0
    keys = f["depths"][vis].view(np.uint32)
    print(f"{kind}: visible {int(vis.sum())}, distinct depth keys {len(np.unique(keys))}, R = {R}")
    assert len(np.unique(keys)) <= (1 if kind == "wall" else 40) and int(vis.sum()) > P // 2
    assert R == f["num_rendered"] and np.array_equal(st["keys"], f["keys"]) and np.array_equal(st["point_list"], f["point_list"])
    same = f["keys"][1:] == f["keys"][:-1]
    assert same.sum() > R // 2 and np.all(f["point_list"][1:][same] > f["point_list"][:-1][same])
    assert np.array_equal(st["ranges"], f["ranges"]) and np.array_equal(st["n_contrib"], f["n_contrib"])
    assert np.array_equal(_np(color), f["color"]) and np.array_equal(_np(depth), f["depth"])
    G = seed_gradient(H, W, 5) * (H * W)
    g = oracle_backward(oracle, case, f, G)
    for k, v in tp._grads_hip(case, G).items():
        assert tp.rel_err(v, g[k].reshape(v.shape)) <= 1e-5, k


def test_row_arena_survives_a_failed_growth(monkeypatch):
    """RowArena._reserve installed the new capacity and offsets before the allocation that can fail; a caller that
    caught the error was left with the new layout over the old, smaller buffer.  Now the arena is unchanged."""
    from gaussianeditor_amd.arena import RowArena

    g = torch.Generator(device=DEV).manual_seed(1)
    P = 3000
    t = dict(xyz=torch.randn(P, 3, device=DEV, generator=g), rot=torch.randn(P, 4, device=DEV, generator=g))
    arena = RowArena(t, headroom=1.0)
    before = {k: arena[k].clone() for k in t}
    cap, ptr = arena.capacity, arena["xyz"].data_ptr()
    real = torch.empty

    def failing_empty(*a, **kw):
        if kw.get("dtype") == torch.uint8 and a and isinstance(a[0], int) and a[0] > 2 * 7 * 4 * cap:
            raise torch.OutOfMemoryError("simulated")
        return real(*a, **kw)

    monkeypatch.setattr(torch, "empty", failing_empty)
    with pytest.raises(RuntimeError, match="cannot allocate"):
        arena.append({"xyz": torch.zeros(5000, 3, device=DEV), "rot": torch.zeros(5000, 4, device=DEV)})
    monkeypatch.setattr(torch, "empty", real)
    assert arena.capacity == cap and arena.P == P and arena["xyz"].data_ptr() == ptr
    for k in t:
        assert torch.equal(arena[k], before[k])
    # ... and it still works: a growth that succeeds, then the rows are where they belong
    n = arena.append({"xyz": torch.ones(5000, 3, device=DEV), "rot": None})
    assert n == P + 5000 and torch.equal(arena["xyz"][:P], before["xyz"]) and float(arena["xyz"][P:].min()) == 1.0
    assert float(arena["rot"][P:].abs().max()) == 0.0


@pytest.mark.parametrize("P", [500_000, 1_000_000])
def test_three_way_parity_on_synth_v2(oracle, P):
    """synth-v2 (gaussianeditor_amd/synth.py: disks on surfaces, bimodal opacity, the camera inside the scene -- every one of
    the 8 160 tiles non-empty, most visible Gaussians receive a gradient; bench.py --scene v2) through the three-way test of
    round 2: reference(no contraction) / oracle / product, integers bit-exact, images and all six gradients within 1e-5."""
    import math

    from gaussianeditor_amd.synth import ring_cameras, synth_scene_v2
    from test_gpu_round2 import three_way

    W, H = 1920, 1080
    sc = synth_scene_v2(P, seed=0)
    cam = ring_cameras(8, W, H)[0]
    case = dict(sc=sc, cam=cam, W=W, H=H, tfx=math.tan(cam.FoVx / 2), tfy=math.tan(cam.FoVy / 2), bg=torch.zeros(3), D=3)
    f = oracle_forward(oracle, case)
    rl = f["ranges"].reshape(-1, 2)
    vis = f["radii"] > 0
    print(f"synth-v2 P={P}: visible {int(vis.sum())}, R = {f['num_rendered']}, empty tiles {int((rl[:, 1] == rl[:, 0]).sum())}")
    assert (rl[:, 1] > rl[:, 0]).all() and vis.mean() > 0.6  # what the workload is for
    three_way(oracle, case, seed_gradient(H, W, 0), 3)


@pytest.mark.parametrize("extra,views", [([], 8), (["--views", "16"], 16)])
def test_bench_eight_ranks_share_one_gpu(extra, views):
    """`bench.py --gpus 8` exactly as the driver's scaling run launches it (torch.distributed.run, one process per rank), on a
    box with ONE GPU: GSR_BENCH_SHARED_GPU=1 puts all eight ranks on GPU 0 over gloo, so the first real 8-GPU run cannot die
    on a rank-count assumption -- views sharded over eight ranks (weak: one each; strong: the fixed batch of 16, two per rank,
    pipelined), eight-way touched-rows exchange, max-over-ranks timing, ONE line from rank 0 with the `multi_gpu` block of
    eight entries.  (The numbers of such a run mean nothing.)"""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GSR_BENCH_SHARED_GPU"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--gaussians",
                        "40000", "--width", "640", "--height", "368", "--steps", "3", "--warmup", "1", "--prewarm", "2",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=1200, cwd=root, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 3 and line["value"] > 0
    assert line["scaling"] == ("strong" if extra else "weak")
    cfg, mg = line["config"], line["multi_gpu"]
    assert cfg["views_per_step"] == views and cfg["views_per_rank"] == views // 8 and cfg["parallelism"] == "dp8-views"
    assert mg["world_size"] == 8 and mg["shared_gpu_test_run"] and len(mg["per_rank"]["step_ms_gpu"]) == 8
    assert cfg["grad_exchange_route"] in ("rows", "dense")
    if cfg["grad_exchange_route"] == "rows":
        assert len(cfg["grad_exchange_rows_per_view"]) == views and all(0 < c <= 40000 for c in cfg["grad_exchange_rows_per_view"])
        assert all(b > 0 for b in mg["per_rank"]["bytes_sent_per_step"])


def test_views_rendered_through_the_l0_entry_points_equal_the_l1_autograd_route(monkeypatch):
    """`multiview._view_forward` / `_view_backward` call `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` directly
    since round 5 (the autograd engine's thread hand-over cost the launch thread 280 us per view and bound the two-stream batch,
    profiles/r05_c_view_pipelining.md).  Same native calls with the same arguments: every forward output and the gradients the
    backward writes without atomics are bit-identical to the L1 route's, the atomically accumulated ones agree to the blend
    backward's run-to-run spread -- without a bucket and with one ("direct" and "rgb" SH exchange)."""
    import gaussianeditor_amd.multiview as mv
    from helpers import settings

    W, H, P = 320, 200, 20000
    case = make_case(P, W, H, seed=5, s0=0.03)
    dev = torch.device(DEV)
    sc = case["sc"]
    params = [sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")]
    G = seed_gradient(H, W, 3).to(dev)
    rs = settings(case, dev)

    def run(autograd, bucket):
        monkeypatch.setattr(mv, "_VIEW_AUTOGRAD", autograd)
        color, radii, depth, grads = mv.render_view_grads(rs, *params, G, bucket)
        torch.cuda.synchronize(dev)
        return _np(color), _np(radii), _np(depth), {k: (None if v is None else _np(v).copy()) for k, v in grads.items()}

    for mode in (None, "direct", "rgb"):
        b0 = None if mode is None else mv.GradBucket(P, 16, dev, sh_exchange=mode)
        b1 = None if mode is None else mv.GradBucket(P, 16, dev, sh_exchange=mode)
        c0, r0, d0, g0 = run(False, b0)
        c1, r1, d1, g1 = run(True, b1)
        assert np.array_equal(c0, c1) and np.array_equal(r0, r1) and np.array_equal(d0, d1)
        assert set(g0) == set(g1)
        for k in g0:
            if g0[k] is None or g1[k] is None:
                assert g0[k] is None and g1[k] is None, (mode, k)
                continue
            scale = max(float(np.abs(g1[k]).max()), 1e-30)
            assert float(np.abs(g0[k] - g1[k]).max()) <= 5e-6 * scale, (mode, k)
        if mode == "rgb":
            assert float(np.abs(_np(b0.rgb) - _np(b1.rgb)).max()) <= 5e-6 * max(float(np.abs(_np(b1.rgb)).max()), 1e-30)


@pytest.mark.parametrize("seed", [365, 110])
def test_deep_translucent_lists_keep_the_gradients_inside_the_bar(oracle, seed):
    """Two configurations of tools/fuzz_v2.py (synth-v2 scenes through tiny images) that exposed product-specific precision
    losses in round 5, both along lists of thousands of translucent entries:
      365 (2 x 16 pixels, ONE tile with a list of 3 231 entries, so forward checkpoints + backward list segments are on): the
          colour behind a segment was C_final - C(checkpoint), two sums near 1 with 6e-8 roundings, divided by a small
          transmittance -- gradients of Gaussians in front of a dense segment 1.1e-5 .. 2.3e-5 off where the reference's own
          backward is 1e-6 from float64.  The forward now accumulates every segment's colour from zero: <= 1.5e-6;
      110 (1 x 151): T / (1 - alpha) with the raw v_rcp_f32 accumulated 2.6e-5 on dL_dscales; with one Newton step 1.6e-5,
          which is where the reference's own backward sits on this scene (four-way: tools/fuzz_v2.py --judge --only 110)."""
    import test_gpu_parity as tp
    from helpers import v2_fuzz_case

    case, sm, D = v2_fuzz_case(seed)
    f, _ = tp._compare_forward(oracle, case, scale_modifier=sm)  # every stage of the forward, bit for bit
    H, W = case["H"], case["W"]
    G = seed_gradient(H, W, seed) * (H * W)
    g = oracle_backward(oracle, case, f, G, scale_modifier=sm)
    h = tp._grads_hip(case, G, scale_modifier=sm)
    errs = {k: tp.rel_err(v, g[k].reshape(v.shape)) for k, v in h.items()}
    print(f"seed {seed}: R = {f['num_rendered']}, errors {errs}")
    if seed == 365:
        assert f["num_rendered"] > 2048 and f["ranges"].reshape(-1, 2).shape[0] == 1  # (one tile, long list: segments are on)
        assert max(errs.values()) <= 3e-6
    else:
        assert max(errs[k] for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh")) <= 1e-5
        assert max(errs.values()) <= 2e-5  # (2.6e-5 - 2.7e-5 with the raw reciprocal)
