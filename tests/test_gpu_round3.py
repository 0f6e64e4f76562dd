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


def test_scene_from_ply_renders_like_the_oracle_and_benches(oracle, tmp_path):
    """The one-command asset path for BASELINE configs[1] (a trained point_cloud.ply): a scene written in the reference's
    save_ply layout (gaussiansplatting/scene/gaussian_model.py:410-445), loaded back as load_ply does (:455-533), activated
    as the model's getters do and rendered on the GPU equals the oracle's render of the same loaded tensors; and
    `bench.py --ply` takes the file."""
    import json
    import math

    from gaussianeditor_amd import scene_ply
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    case = make_case(30000, 640, 360, seed=13, s0=0.03)
    sc, cam = case["sc"], case["cam"]
    path = str(tmp_path / "point_cloud.ply")
    scene_ply.save_gaussians_ply(path, sc["xyz"], sc["features"][:, :1], sc["features"][:, 1:], torch.logit(sc["opacity"]),
                                 torch.log(sc["scaling"]), sc["rotation"] * 1.7)  # (the file holds UNNORMALISED quaternions)
    loaded = scene_ply.activated(scene_ply.load_gaussians_ply(path))
    assert loaded["features"].shape == (30000, 16, 3)
    f = oracle.forward(loaded["xyz"], loaded["scaling"], loaded["rotation"], loaded["opacity"], loaded["features"], None, None,
                       cam.world_view_transform, cam.full_proj_transform, cam.camera_center, case["bg"], 640, 360, case["tfx"],
                       case["tfy"], 1.0, 3)
    d = lambda t: t.to(DEV)  # noqa: E731
    rs = GaussianRasterizationSettings(360, 640, case["tfx"], case["tfy"], d(case["bg"]), 1.0, d(cam.world_view_transform),
                                       d(cam.full_proj_transform), 3, d(cam.camera_center), False, False)
    color, radii, depth = GaussianRasterizer(rs)(d(loaded["xyz"]), torch.zeros(30000, 3, device=DEV), d(loaded["opacity"]),
                                                 shs=d(loaded["features"]), scales=d(loaded["scaling"]), rotations=d(loaded["rotation"]))
    assert np.array_equal(radii.cpu().numpy(), f["radii"]) and np.array_equal(color.cpu().numpy(), f["color"])
    assert np.array_equal(depth.cpu().numpy(), f["depth"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ply", path, "--steps", "3", "--warmup", "1", "--width", "640",
                        "--height", "360", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["gaussians"] == 30000 and "point_cloud.ply" in line["config"]["workload"] and line["value"] > 0
    assert math.isfinite(line["forward_ms"]) and line["config"]["num_rendered"] > 0


def test_row_arena_prune_and_append_bit_exact():
    """gaussianeditor_amd/arena.py: compaction into the other half and in-place appends equal `tensor[mask]` / `torch.cat`
    bit for bit for the row shapes of a Gaussian model (gaussiansplatting/scene/gaussian_model.py:568-641), views stay
    inside ONE buffer until an append overflows it, and the optimizer keeps its Parameter objects and moments."""
    from gaussianeditor_amd.arena import OptimizerArena, RowArena

    g = torch.Generator(device=DEV).manual_seed(5)
    P = 10007
    t = dict(xyz=torch.randn(P, 3, device=DEV, generator=g), f_rest=torch.randn(P, 15, 3, device=DEV, generator=g),
             opacity=torch.randn(P, 1, device=DEV, generator=g), rot=torch.randn(P, 4, device=DEV, generator=g),
             radii=torch.randint(0, 99, (P,), device=DEV, generator=g, dtype=torch.int32),
             mask=torch.rand(P, device=DEV, generator=g) > 0.5)
    want = {k: v.clone() for k, v in t.items()}
    A = RowArena(t, headroom=1.2)
    base = A._buf.data_ptr()
    for step in range(4):
        keep = torch.rand(A.P, device=DEV, generator=g) > 0.1
        A.compact(keep)
        want = {k: v[keep] for k, v in want.items()}
        n = 300 + 7 * step
        ext = {"xyz": torch.randn(n, 3, device=DEV, generator=g), "f_rest": torch.randn(n, 15, 3, device=DEV, generator=g),
               "rot": torch.randn(n, 4, device=DEV, generator=g), "mask": torch.rand(n, device=DEV, generator=g) > 0.5}
        A.append(ext, n=n)  # opacity and radii: zero rows
        want = {k: torch.cat((v, ext[k] if k in ext else torch.zeros((n,) + tuple(v.shape[1:]), dtype=v.dtype, device=DEV)))
                for k, v in want.items()}
        for k in want:
            assert A[k].dtype == want[k].dtype and torch.equal(A[k], want[k]), (step, k)
        assert A.P == want["xyz"].shape[0] and A.allocations == 1 and A._buf.data_ptr() == base
    A.append({"xyz": torch.zeros(P * 2, 3, device=DEV)}, n=P * 2)  # does not fit: the arena grows once
    assert A.allocations == 2 and torch.equal(A["xyz"][:want["xyz"].shape[0]], want["xyz"]) and float(A["opacity"][-1]) == 0.0
    A.compact(torch.zeros(A.P, dtype=torch.bool, device=DEV))
    assert A.P == 0 and A["f_rest"].shape == (0, 15, 3)
    # with an optimizer: the Parameters and the state dictionary survive, the moments follow the rows
    from gaussianeditor_amd.optim import FusedMaskedAdam

    p1 = torch.nn.Parameter(torch.randn(500, 3, device=DEV, generator=g))
    p2 = torch.nn.Parameter(torch.randn(500, 1, device=DEV, generator=g))
    opt = FusedMaskedAdam([dict(params=[p1], lr=1e-2, name="xyz"), dict(params=[p2], lr=1e-2, name="opacity")], lr=0.0, eps=1e-15)
    (p1.sum() * 2 + (p2 ** 2).sum()).backward()
    opt.step()
    m_before, x_before = opt.state[p1]["exp_avg"].clone(), p1.detach().clone()
    oa = OptimizerArena(opt, extra=dict(denom=torch.ones(500, 1, device=DEV)))  # (the groups now hold Parameters on arena views)
    assert torch.equal(oa.params()["xyz"].detach(), x_before) and torch.equal(opt.state[oa.params()["xyz"]]["exp_avg"], m_before)
    keep = torch.rand(500, device=DEV, generator=g) > 0.3
    new = oa.prune(keep)
    q1 = new["xyz"]
    assert opt.param_groups[0]["params"][0] is q1 and q1.grad is None and q1.is_leaf and len(opt.state) == 2
    assert torch.equal(q1.detach(), x_before[keep]) and torch.equal(opt.state[q1]["exp_avg"], m_before[keep])
    new = oa.append({"xyz": torch.full((5, 3), 7.0, device=DEV), "opacity": torch.full((5, 1), 8.0, device=DEV)})
    q1, q2 = new["xyz"], new["opacity"]
    assert q1.shape[0] == int(keep.sum()) + 5 and float(q1.detach()[-1, 0]) == 7.0 and float(opt.state[q1]["exp_avg"][-5:].abs().sum()) == 0.0
    assert float(oa.extra["denom"][-5:].abs().sum()) == 0.0 and float(oa.extra["denom"][0]) == 1.0
    (q1.sum() + q2.sum()).backward()
    opt.step()  # the grown optimizer keeps stepping
    assert int(opt.state[q1]["step"]) == 2 and len(opt.state) == 2


@pytest.mark.parametrize("mode", ["direct", "rgb"])
def test_persistent_gradient_rows_equal_the_dense_backward(mode):
    """GradBucket(persistent_rows=True): the backward rewrites a zero gradient row only when it does not already hold the
    zeros of an earlier backward (gsr_preprocess_backward_rows).  Over a sequence of different views -- Gaussians go from
    touched to untouched and back -- every gradient tensor equals the one a fresh bucket (every row written) gets, every time (up to the run-to-run
    spread of the blend backward's float atomics; the all-zero rows are the same rows);
    and the state says "zero" exactly for the rows that are zero."""
    from gaussianeditor_amd.multiview import GradBucket, render_view_grads
    from helpers import seed_gradient, settings

    P, W, H = 60000, 640, 480
    d = lambda t: t.to(DEV)  # noqa: E731
    def _same(a, c, what):
        # two runs of the blend backward differ in the last bits (float atomics): same zero rows, values within 1e-5 of max
        assert not bool(a.isnan().any()), what
        za, zc = (a.reshape(P, -1) == 0).all(dim=1), (c.reshape(P, -1) == 0).all(dim=1)
        assert torch.equal(za, zc), what
        assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max()) + 1e-30, what

    keep = GradBucket(P, 16, DEV, sh_exchange=mode, persistent_rows=True)
    keep.flat.fill_(float("nan"))  # nothing may survive from before the first backward
    if keep.rgb is not None:
        keep.rgb.fill_(float("nan"))
    zero_rows_seen = 0
    for step, view in enumerate([0, 3, 0, 5, 5, 1]):
        case = make_case(P, W, H, seed=11, s0=0.02, view=view, nviews=8)
        sc = case["sc"]
        G = d(seed_gradient(H, W, 40 + step)) * H * W
        fresh = GradBucket(P, 16, DEV, sh_exchange=mode)
        outs = []
        for b in (keep, fresh):
            _, radii, _, grads = render_view_grads(settings(case, DEV), d(sc["xyz"]), d(sc["opacity"]), d(sc["features"]),
                                                   d(sc["scaling"]), d(sc["rotation"]), G, b)
            outs.append((grads, radii))
        for name in ("means3D", "sh", "opacities", "scales", "rotations", "means2D"):
            a, c = outs[0][0][name], outs[1][0][name]
            assert (a is None) == (c is None), name
            if a is not None:
                _same(a, c, (step, name))
        if mode == "rgb":
            _same(keep.rgb, fresh.rgb, (step, "rgb"))
        rows = torch.cat([fresh.views[n].reshape(P, -1) for n in ("means3D", "scales", "rotations")] +
                         ([fresh.rgb] if mode == "rgb" else [fresh.views["sh"].reshape(P, -1)]), dim=1)
        nz = (rows != 0).any(dim=1)
        st = keep.row_state.bool()
        assert bool((st | ~nz).all())          # a row with a non-zero entry is marked
        zero_rows_seen += int((~st).sum())
    assert zero_rows_seen > P  # (most rows are zero for every view: the state is not just "all dirty")
    keep.invalidate_rows()
    assert int(keep.row_state.min()) == 1


def test_bench_line_carries_the_contract_fields_and_the_gradient_row_mode():
    """`bench.py` (small run): one JSON line with the contract's keys, `roofline` and the note how gradient rows are written;
    `--persistent-grads` is a labelled development mode, not the default."""
    import json

    for extra, word in (([], "every row"), (["--persistent-grads"], "persistent")):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gaussians", "50000", "--steps", "5", "--warmup", "2",
                            "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT,
                           env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
        assert p.returncode == 0, p.stderr[-2000:]
        line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in line, key
        assert line["n_gpus"] == 1 and line["steps"] == 5 and line["value"] > 0 and word in line["config"]["grad_rows"]
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_bench_two_ranks_share_one_gpu_and_take_the_exchange_path():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), on a box with ONE GPU:
    `GSR_BENCH_SHARED_GPU=1` puts both ranks on GPU 0 with the gloo backend, so every line of the N > 1 bench path runs --
    views sharded by rank, the touched-rows exchange, max-over-ranks timing, rank 0 printing the one line.  (The numbers of
    such a run mean nothing; RCCL itself is covered at world size 1 and, where two GPUs exist, 2.)"""
    import json
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GSR_BENCH_SHARED_GPU"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gaussians",
                        "50000", "--width", "640", "--height", "368", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    cfg = line["config"]
    assert cfg["views_per_step"] == 2 and cfg["parallelism"] == "dp2-views" and cfg["grad_exchange"] == "rgb"
    assert cfg["grad_exchange_route"] in ("rows", "dense")
    if cfg["grad_exchange_route"] == "rows":
        assert len(cfg["grad_exchange_rows_per_view"]) == 2 and all(0 < c <= 50000 for c in cfg["grad_exchange_rows_per_view"])
