"""CPU, world_size 2 over gloo: the multi-view data-parallel step (one view per rank, one flat SUM
all-reduce of the gradient bucket, MAX all-reduce of the radii) equals a single process rendering
both views and summing.  The native library is replaced by the oracle stand-in on CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import make_case, oracle_backward, oracle_forward, rel_err, seed_gradient, settings

P, W, H = 1200, 64, 48


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _segments(bucket):
    """The bucket's segments in layout order, concatenated WITHOUT the alignment padding between them."""
    return np.concatenate([v.detach().numpy().reshape(-1) for v in bucket.flat_views().values()])


def _worker_rgb(rank, world, port, out_dir, rows, s0, P=P):
    """The "rgb" exchange: colour gradients all-gathered (dense, or as packed touched rows), SH gradient rebuilt on every rank."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pytest as _pt

    import oracle_backend
    from gaussianeditor_amd.multiview import GradBucket, allreduce_view_grads, render_view_grads

    mpatch = _pt.MonkeyPatch()
    oracle_backend.install(mpatch)
    try:
        case = make_case(P, W, H, seed=5, s0=s0, view=rank, nviews=world)
        sc = case["sc"]
        sparse_rows = rows == "sparse_rows"  # the touched-rows route, writing only the rows some view touched
        if sparse_rows:
            rows = True
        bucket = GradBucket(P, 16, "cpu", sparse_rows=sparse_rows)  # "auto" -> "rgb" because two ranks run
        assert bucket.sh_exchange == "rgb" and bucket.flat.numel() >= P * 14
        assert all(v.data_ptr() % 16 == 0 for v in bucket.views.values())  # whatever P is
        G = seed_gradient(H, W, 100 + rank) * H * W
        if rows == "auto":  # the whole step as bench.py runs it: the radii's MAX all-reduce overlaps the backward
            from gaussianeditor_amd.multiview import multiview_step

            touched_of = []
            orig = allreduce_view_grads.__globals__["_C"].view_message_plan_blend

            def spy(touched):  # the whole step plans from the blend backward's row mask (uint8 (P,)), between K7 and K8+K9
                assert tuple(touched.shape) == (P,) and touched.dtype == torch.uint8
                touched_of.append(float((touched != 0).float().mean()))
                return orig(touched)

            mpatch.setattr(allreduce_view_grads.__globals__["_C"], "view_message_plan_blend", spy)
            params = {k: sc[k] for k in ("xyz", "opacity", "features", "scaling", "rotation")}
            color, radii, depth, grads = multiview_step(settings(case, "cpu"), params, G, bucket)
            assert grads["sh"] is None
            mode, touched = bucket.last_route, touched_of[0]
        else:
            color, radii, depth, grads = render_view_grads(settings(case, "cpu"), sc["xyz"], sc["opacity"], sc["features"],
                                                           sc["scaling"], sc["rotation"], G, bucket)
            assert grads["sh"] is None
            rows_of = torch.cat([v.reshape(P, -1) for v in bucket.flat_views().values()] + [bucket.rgb], dim=1)
            touched = float((rows_of != 0).any(dim=1).float().mean())
            if sparse_rows:  # poison what must not be read afterwards: the rows that are zero on this rank
                own = rows_of.any(dim=1)
            mode = allreduce_view_grads(bucket, radii, sparse=(rank >= 0), rows=rows)
            if sparse_rows:
                valid = bucket.row_valid.bool()
                assert mode == "rows", mode
                assert bool((valid | ~own).all()), "a row this rank touched is not marked valid"
                assert int(valid.sum()) > 0  # (in this small scene the two views together reach nearly every Gaussian; the
                #                              GPU suite checks a view pair that leaves 86 % of the rows invalid)
                for name in ("means3D", "scales", "rotations", "means2D", "opacities"):
                    assert not bool(bucket.views[name][~valid].any())  # (stale = this rank's own zeros)
                bucket.views["sh"][~valid] = 0.0  # uninitialised by contract: what the consumer takes them for
        np.savez(os.path.join(out_dir, f"rgb_rank{rank}.npz"), flat=_segments(bucket), sh=bucket.views["sh"].numpy(),
                 radii=radii.numpy(), touched=np.array([touched]), rows_route=np.array([mode == "rows"]))
    finally:
        mpatch.undo()
        dist.destroy_process_group()


def _worker(rank, world, port, out_dir, P=P):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pytest as _pt

    import oracle_backend
    from gaussianeditor_amd.multiview import GradBucket, multiview_step

    mpatch = _pt.MonkeyPatch()
    oracle_backend.install(mpatch)
    try:
        case = make_case(P, W, H, seed=5, s0=0.07, view=rank, nviews=world)
        sc = case["sc"]
        bucket = GradBucket(P, 16, "cpu", sh_exchange="direct")  # the SH gradient travels inside the all-reduced bucket
        G = seed_gradient(H, W, 100 + rank) * H * W
        params = {k: sc[k] for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        from gaussianeditor_amd.multiview import allreduce_view_grads, render_view_grads

        color, radii, depth, grads = render_view_grads(settings(case, "cpu"), params["xyz"], params["opacity"],
                                                       params["features"], params["scaling"], params["rotation"], G, bucket)
        local = bucket.flat.clone()
        mode_sparse = allreduce_view_grads(bucket, radii.clone(), sparse=True, sparse_threshold=1.0)
        sparse_result = bucket.flat.clone()
        bucket.flat.copy_(local)
        mode_dense = allreduce_view_grads(bucket, radii, sparse=False)
        assert (mode_sparse, mode_dense) == ("sparse", "dense")
        # the packed exchange of the union rows reproduces the dense all-reduce
        assert torch.allclose(sparse_result, bucket.flat, rtol=0, atol=0) or \
            float((sparse_result - bucket.flat).abs().max()) <= 1e-6 * float(bucket.flat.abs().max())
        touched = float((local.view(-1) != 0).float().mean())
        assert 0.0 < touched < 1.0
        # the gradients are views of the flat bucket: no copies between the backward and the collective
        assert grads["sh"].data_ptr() == bucket.views["sh"].data_ptr()
        assert bucket.flat.numel() >= P * (14 + 3 * 16) and (P % 4 or bucket.flat.numel() == P * (14 + 3 * 16))
        # ... and they hold what autograd returned, for EVERY segment (P % 4 != 0 used to leave all but means3D empty)
        for name in ("means3D", "sh", "opacities", "scales", "rotations", "means2D"):
            assert grads[name].data_ptr() == bucket.views[name].data_ptr(), name
            assert float(bucket.views[name].abs().max()) > 0.0, name
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=_segments(bucket), radii=radii.numpy())
    finally:
        mpatch.undo()
        dist.destroy_process_group()


@pytest.mark.parametrize("P", [1200, 1201, 1203])  # P % 4 != 0: every segment start is padded to 16 bytes
def test_two_rank_allreduce_matches_single_process(oracle, tmp_path, P):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), P), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # replicas hold identical reduced buffers
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["radii"], r1["radii"])
    # single-process reference: sum over the two views
    tot, rad = None, None
    for v in range(world):
        case = make_case(P, W, H, seed=5, s0=0.07, view=v, nviews=world)
        f = oracle_forward(oracle, case)
        g = oracle_backward(oracle, case, f, seed_gradient(H, W, 100 + v) * H * W)
        flat = np.concatenate([g[k].reshape(-1) for k in ("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations",
                                                          "dL_dmeans2D", "dL_dopacity")])
        tot = flat if tot is None else tot + flat
        rad = f["radii"] if rad is None else np.maximum(rad, f["radii"])
    assert rel_err(r0["flat"], tot) < 1e-6
    assert np.array_equal(r0["radii"], rad)


@pytest.mark.parametrize("rows,s0,P", [(False, 0.07, 1200), (True, 0.07, 1200), ("auto", 0.07, 1200), ("auto", 0.004, 1200),
                                       (False, 0.07, 1201), (True, 0.07, 1201), ("auto", 0.07, 1202),
                                       ("sparse_rows", 0.004, 1201)])
def test_two_rank_rgb_exchange_matches_single_process(oracle, tmp_path, rows, s0, P):
    world = 2
    mp.spawn(_worker_rgb, args=(world, _free_port(), str(tmp_path), rows, s0, P), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rgb_rank0.npz"), np.load(tmp_path / "rgb_rank1.npz")
    # replicas agree bit for bit, including the SH gradient each of them rebuilt on its own
    for k in ("flat", "sh", "radii"):
        assert np.array_equal(r0[k], r1[k]), k
    tot, sh = None, None
    for v in range(world):
        case = make_case(P, W, H, seed=5, s0=s0, view=v, nviews=world)
        f = oracle_forward(oracle, case)
        g = oracle_backward(oracle, case, f, seed_gradient(H, W, 100 + v) * H * W)
        flat = np.concatenate([g[k].reshape(-1) for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dmeans2D",
                                                          "dL_dopacity")])
        tot = flat if tot is None else tot + flat
        sh = g["dL_dsh"] if sh is None else sh + g["dL_dsh"]  # a single process accumulates view after view
    assert rel_err(r0["flat"], tot) < 1e-6
    # the route: forced, or chosen from the gathered counts (72 B per touched row against the dense route's bytes)
    t = float(r0["touched"][0]) + float(r1["touched"][0])
    want_rows = rows in (True, "sparse_rows") or (rows == "auto" and 72 * t <= 0.6 * (12 * world + 112))
    assert bool(r0["rows_route"][0]) == bool(r1["rows_route"][0]) == want_rows, (t, rows)
    if want_rows:  # packed rows are added view after view to zeros: the single process's sums, bit for bit
        assert np.array_equal(r0["flat"], tot)
    # rebuilt from 3 floats per view = the accumulated per-view SH gradients, bit for bit
    assert np.array_equal(r0["sh"], sh.reshape(r0["sh"].shape))


def test_sh_grad_compose_oracle_equals_backward_sh(oracle):
    """The restated composition against the backward's own dL_dsh, one view, degrees 0-3."""
    for D in (0, 1, 2, 3):
        case = make_case(1500, 96, 64, seed=7 + D, s0=0.06, sh_degree=3)
        case["D"] = D
        f = oracle_forward(oracle, case)
        g = oracle_backward(oracle, case, f, seed_gradient(64, 96, 3) * 64 * 96)
        vis = (f["radii"] > 0)[:, None]
        rgb = np.where(np.logical_and(vis, f["clamped"] == 0), g["dL_dcolors"].reshape(-1, 3), 0.0).astype(np.float32)
        sh = oracle.sh_grad_compose(case["sc"]["xyz"].numpy(), case["cam"].camera_center.numpy()[None], rgb[None], D, 16)
        assert np.array_equal(sh, g["dL_dsh"].reshape(sh.shape)), D


def test_bucket_layout_and_allocator():
    from gaussianeditor_amd.multiview import GradBucket

    b = GradBucket(8, 16, "cpu")
    assert [tuple(b.views[k].shape) for k in ("means3D", "sh", "opacities", "scales", "rotations", "means2D")] == \
        [(8, 3), (8, 16, 3), (8, 1), (8, 3), (8, 4), (8, 3)]
    b.flat.fill_(1.0)
    v = b.allocator("means2D", (8, 3), True)
    assert v is b.views["means2D"] and float(v.abs().sum()) == 0.0 and float(b.views["sh"].sum()) == 8 * 48
    assert b.allocator("sh", (8, 4, 3), False) is None          # shape mismatch -> private tensor
    assert b.allocator("colors_precomp", (8, 3), True) is None  # not a parameter gradient
    # the blend backward's accumulator table is the binding's workspace: asking for it only opens a backward
    assert b.allocator("acc_rows", (128,), False) is None
    # P % 4 != 0: every segment still starts on a 16-byte boundary (padding words between the segments), so the
    # backward's gradients always ARE the bucket's segments -- a bucket that handed out None here lost them silently
    for P_ in (1, 5, 6, 7, 1201):
        b2 = GradBucket(P_, 16, "cpu")
        for name, v in b2.views.items():
            assert v.data_ptr() % 16 == 0, (P_, name)
            assert b2.allocator(name, tuple(v.shape), False) is v
        b3 = GradBucket(P_, 16, "cpu", sh_exchange="rgb")
        assert all(v.data_ptr() % 16 == 0 for v in b3.views.values()) and b3.allocator("sh_rgb", (P_, 3), False) is b3.rgb
        # the two opt-in row masks: absent by default; persistent rows start out "may hold anything" and return there
        assert b3.row_valid is None and b3.row_state is None and b3.allocator("row_state", (P_,), False) is None
        b4 = GradBucket(P_, 16, "cpu", sparse_rows=True, persistent_rows=True)
        assert b4.row_valid.dtype == b4.row_state.dtype == torch.uint8 and int(b4.row_state.min()) == int(b4.row_valid.min()) == 1
        # the state is only handed out in a backward whose gradients were all answered with the bucket's own tensors
        assert b4.allocator("row_state", (P_,), False) is None
        b4.allocator("acc_rows", (16 * P_,), False)
        for nm in ("means2D", "opacities", "means3D", "scales", "rotations"):
            assert b4.allocator(nm, tuple(b4.views[nm].shape), False) is b4.views[nm]
        assert b4.allocator("row_state", (P_,), False) is None  # (the SH / colour gradient is missing)
        assert b4.allocator("sh", (P_, 16, 3), False) is b4.views["sh"]
        assert b4.allocator("row_state", (P_,), False) is b4.row_state and b4.allocator("row_state", (P_ + 1,), False) is None
        b4.allocator("acc_rows", (16 * P_,), False)  # the next backward starts from nothing again
        assert b4.allocator("row_state", (P_,), False) is None
        b4.row_state.zero_()
        b4.invalidate_rows()
        assert int(b4.row_state.min()) == 1


def test_grad_allocator_rides_on_the_autograd_node(oracle, monkeypatch):
    """The allocator belongs to ONE render (it is an attribute of that render's autograd node): the backward finds it
    whichever thread runs it -- autograd runs CUDA backwards on its own device thread, and the web UI renders from a
    second Python thread while a training thread steps -- and no other render sees it."""
    import threading

    import oracle_backend
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer, _C

    oracle_backend.install(monkeypatch)
    case = make_case(300, 48, 32, seed=2, s0=0.08)
    sc = case["sc"]
    calls = {"a": [], "b": []}

    def render():
        leaves = [sc[k].clone().requires_grad_(True) for k in ("xyz", "opacity", "features", "scaling", "rotation")]
        x, o, f, s_, r = leaves
        c, _, _ = GaussianRasterizer(settings(case, "cpu"))(x, torch.zeros_like(x, requires_grad=True), o, shs=f, scales=s_,
                                                            rotations=r)
        return c, leaves

    ca, la = render()
    cb, lb = render()
    _C.attach_grad_allocator(ca, lambda name, shape, zero: calls["a"].append(name))
    # the backward of render a on ANOTHER thread still finds a's allocator; render b (same thread as the attach) has none
    t = threading.Thread(target=lambda: ca.sum().backward())
    t.start()
    t.join()
    cb.sum().backward()
    assert "means3D" in calls["a"] and "sh" in calls["a"] and calls["b"] == []
    with pytest.raises(RuntimeError, match="not the output"):
        _C.attach_grad_allocator(torch.zeros(3, requires_grad=True) * 2, None)


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: what keeps the sharded loop equal to the reference's one-process loop besides the gradient exchange
# (SURVEY.md section 8(e)): the batch-loss scaling and replica-identical densification.
# ---------------------------------------------------------------------------------------------------------------------
def _toy_images(params, views):
    """A differentiable stand-in for `render` on CPU: (K,H,W,3) images from (P,3) parameters and per-view phases."""
    base = torch.tanh(params).sum(dim=0)  # (3,)
    return torch.stack([torch.sin(base * (1.0 + v)).view(1, 1, 3).expand(4, 5, 3) * (0.5 + 0.1 * v) for v in views], 0)


def _reference_batch_loss(params, anchor, views, gts):
    """The reference's loss over a batch in ONE process, GassuianEditorEdit.py:100-104 (L1 = mean over the stacked batch,
    perceptual stand-in = .sum() over the batch) + :133-145 (the anchor term, once per step)."""
    images = _toy_images(params, views)
    l1 = torch.nn.functional.l1_loss(images, gts)
    perceptual = ((images - gts) ** 2).mean(dim=(1, 2, 3)).sum()
    return 10.0 * l1 + 10.0 * perceptual + 5.0 * ((params - anchor) ** 2).mean()


def _worker_loss_scale(rank, world, port, out_dir, per_step):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussianeditor_amd.multiview import batch_loss_scale

        g = torch.Generator().manual_seed(3)
        params0 = torch.randn(50, 3, generator=g)
        anchor = params0 + 0.1 * torch.randn(50, 3, generator=g)
        gts = torch.rand(world, 4, 5, 3, generator=g)
        views = [float(v) for v in range(world)]
        p = params0.clone().requires_grad_(True)
        sc = batch_loss_scale(world, rank, per_step=per_step)
        img = _toy_images(p, [views[rank]])
        local = (sc["mean"] * 10.0 * torch.nn.functional.l1_loss(img, gts[rank:rank + 1])
                 + sc["sum"] * 10.0 * ((img - gts[rank:rank + 1]) ** 2).mean(dim=(1, 2, 3)).sum()
                 + sc["per_step"] * 5.0 * ((p - anchor) ** 2).mean())
        local.backward()
        grad, loss = p.grad.clone(), local.detach().clone().view(1)
        dist.all_reduce(grad)  # what the gradient exchange forms
        dist.all_reduce(loss)
        q = params0.clone().requires_grad_(True)
        ref = _reference_batch_loss(q, anchor, views, gts)
        ref.backward()
        np.savez(os.path.join(out_dir, f"ls_{per_step}_{rank}.npz"), grad=grad.numpy(), ref_grad=q.grad.numpy(),
                 loss=loss.numpy(), ref_loss=ref.detach().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("per_step", ["split", "rank0"])
@pytest.mark.parametrize("world", [2, 4])
def test_batch_loss_scale_reproduces_the_single_process_loss(tmp_path, world, per_step):
    """L1 is a batch MEAN, the perceptual term a batch SUM, the anchor loss once per step
    (threestudio/systems/GassuianEditorEdit.py:100-104, 133-145): with batch_loss_scale the summed local losses and the
    all-reduced gradients equal the reference's one-process batch loss and its gradients."""
    mp.spawn(_worker_loss_scale, args=(world, _free_port(), str(tmp_path), per_step), nprocs=world, join=True)
    for r in range(world):
        z = np.load(tmp_path / f"ls_{per_step}_{r}.npz")
        assert abs(float(z["loss"][0]) - float(z["ref_loss"])) <= 1e-6 * abs(float(z["ref_loss"]))
        assert np.abs(z["grad"] - z["ref_grad"]).max() <= 1e-6 * np.abs(z["ref_grad"]).max()
    from gaussianeditor_amd.multiview import batch_loss_scale

    assert batch_loss_scale(1) == {"mean": 1.0, "sum": 1.0, "per_step": 1.0}
    with pytest.raises(ValueError):
        batch_loss_scale(0)
    with pytest.raises(ValueError):
        batch_loss_scale(2, 2)


class _ToyModel:
    """The tensor side of GaussianModel.densify_and_split (gaussiansplatting/scene/gaussian_model.py:673-728) on CPU."""

    def __init__(self):
        g = torch.Generator().manual_seed(11)
        self.xyz = torch.randn(200, 3, generator=g)
        self.scaling = torch.rand(200, 3, generator=g) * 0.1
        self.grads = torch.rand(200, generator=g)

    def densify_and_split(self, N=2):
        sel = self.grads >= 0.7
        stds = self.scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3)), std=stds)  # :685-687: the process RNG
        new_xyz = samples + self.xyz[sel].repeat(N, 1)
        keep = ~sel
        self.xyz = torch.cat((self.xyz[keep], new_xyz))
        self.scaling = torch.cat((self.scaling[keep], stds / (0.8 * N)))
        self.grads = torch.zeros(self.xyz.shape[0])
        return int(sel.sum())


def _worker_densify(rank, world, port, out_dir, synchronized):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussianeditor_amd.multiview import densify_synchronized, replicas_identical

        torch.manual_seed(1000 + rank)  # launch.py:102-103: pl.seed_everything(cfg.seed + get_rank())
        m = _ToyModel()
        assert replicas_identical([m.xyz, m.scaling])
        before = torch.rand(1)  # the rank's own stream of random numbers ...
        torch.manual_seed(1000 + rank)
        _ = torch.rand(1)
        if synchronized:
            n = densify_synchronized(m.densify_and_split, replicated=lambda: [m.xyz, m.scaling, m.grads])
        else:
            n = m.densify_and_split()
        after = torch.rand(1)  # ... continues where it was: the synchronised step leaves the generators untouched
        torch.manual_seed(1000 + rank)
        _ = torch.rand(1)
        if synchronized and rank == 0:  # the source rank's stream advances by the one seed it draws
            _ = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
        expected_after = torch.rand(1)
        same = replicas_identical([m.xyz, m.scaling])
        np.savez(os.path.join(out_dir, f"dz_{int(synchronized)}_{rank}.npz"), xyz=m.xyz.numpy(), n=np.array([n]),
                 same=np.array([same]), rng_kept=np.array([bool(torch.equal(after, expected_after)) or not synchronized]),
                 before=before.numpy())
        if synchronized:  # a diverged replica is refused, not trained on
            m.xyz[0, 0] += float(rank)
            try:
                densify_synchronized(lambda: None, replicated=lambda: [m.xyz])
                raised = False
            except RuntimeError:
                raised = True
            np.savez(os.path.join(out_dir, f"dzr_{rank}.npz"), raised=np.array([raised]))
    finally:
        dist.destroy_process_group()


def test_densify_synchronized_keeps_replicas_bit_identical(tmp_path):
    """densify_and_split draws torch.normal from the process RNG (gaussian_model.py:685-687) and every rank is seeded
    differently (launch.py:102-103): unsynchronised replicas differ after one split, synchronised ones are bit-identical,
    and the per-rank RNG streams continue undisturbed."""
    world = 2
    mp.spawn(_worker_densify, args=(world, _free_port(), str(tmp_path), False), nprocs=world, join=True)
    a, b = (np.load(tmp_path / f"dz_0_{r}.npz") for r in range(world))
    assert a["n"][0] == b["n"][0] > 0 and not a["same"][0] and not np.array_equal(a["xyz"], b["xyz"])  # the problem
    mp.spawn(_worker_densify, args=(world, _free_port(), str(tmp_path), True), nprocs=world, join=True)
    a, b = (np.load(tmp_path / f"dz_1_{r}.npz") for r in range(world))
    assert a["same"][0] and b["same"][0] and np.array_equal(a["xyz"], b["xyz"]) and a["xyz"].shape[0] > 200
    assert a["rng_kept"][0] and b["rng_kept"][0] and not np.array_equal(a["before"], b["before"])
    assert all(np.load(tmp_path / f"dzr_{r}.npz")["raised"][0] for r in range(world))


def _worker_speculative_cap(rank, world, port, out_dir):
    """Three steps on ONE bucket: the second and third size their messages from the step before (no wait for the counts
    before pack / all-gather are enqueued); the third touches far more rows than the second, so its speculative messages
    overflow and are sent again."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pytest as _pt

    import oracle_backend
    from gaussianeditor_amd import multiview as mv

    mpatch = _pt.MonkeyPatch()
    oracle_backend.install(mpatch)
    try:
        Pn = 1500
        case = make_case(Pn, W, H, seed=5, s0=0.07, view=rank, nviews=world)
        sc = case["sc"]
        params = {k: sc[k] for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        bucket = mv.GradBucket(Pn, 16, "cpu")
        sends = []
        orig_pack = mv._C.view_message_pack
        mpatch.setattr(mv._C, "view_message_pack", lambda plan, g5, rgb, cam, cap, msg: (sends.append(int(cap)), orig_pack(plan, g5, rgb, cam, cap, msg))[1])
        Gfull = seed_gradient(H, W, 100 + rank) * H * W
        Gsmall = Gfull.clone()
        Gsmall[:, :, W // 6:] = 0.0  # only a strip of the image carries a gradient: few touched rows
        log = []
        for step, G in enumerate((Gsmall, Gsmall, Gfull)):
            n0 = len(sends)
            mv.multiview_step(settings(case, "cpu"), params, G, bucket, rows=True)
            log.append((len(sends) - n0, list(bucket.last_counts), int(bucket._cap_hint)))
        np.savez(os.path.join(out_dir, f"spec_{rank}.npz"), flat=_segments(bucket), sh=bucket.views["sh"].numpy(),
                 sends=np.array([e[0] for e in log]), counts=np.array([e[1] for e in log]), hints=np.array([e[2] for e in log]),
                 caps=np.array(sends))
    finally:
        mpatch.undo()
        dist.destroy_process_group()


def test_speculative_message_size_is_exact_when_it_overflows(oracle, tmp_path):
    world = 2
    mp.spawn(_worker_speculative_cap, args=(world, _free_port(), str(tmp_path), ), nprocs=world, join=True)
    z = [np.load(tmp_path / f"spec_{r}.npz") for r in range(world)]
    # step 0 knows no hint (one send, exact cap); step 1 speculates and fits (one send); step 2 speculates on step 1's small
    # counts, overflows and sends again
    assert list(z[0]["sends"]) == [1, 1, 2] and list(z[1]["sends"]) == [1, 1, 2]
    assert z[0]["counts"][2].max() > z[0]["hints"][1] >= z[0]["counts"][1].max()
    assert np.array_equal(z[0]["counts"], z[1]["counts"]) and np.array_equal(z[0]["hints"], z[1]["hints"])
    # the result of the overflowing step: the single-process sum of the two views' gradients, bit for bit on both replicas
    want_flat, want_sh = None, None
    for v in range(world):
        case = make_case(1500, W, H, seed=5, s0=0.07, view=v, nviews=world)
        f = oracle_forward(oracle, case)
        g = oracle_backward(oracle, case, f, seed_gradient(H, W, 100 + v) * H * W)
        segs = np.concatenate([g[k].reshape(-1) for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dopacity")])
        want_flat = segs if want_flat is None else want_flat + segs
        want_sh = g["dL_dsh"].reshape(1500, 16, 3) if want_sh is None else want_sh + g["dL_dsh"].reshape(1500, 16, 3)
    for r in range(world):
        assert np.array_equal(z[r]["flat"], want_flat.astype(np.float32)) and np.array_equal(z[r]["sh"], want_sh.astype(np.float32))


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: a batch of K = 8 views on N = 1 / 2 / 4 ranks (the reference loops over the whole batch in one
# process, threestudio/systems/GassuianEditor.py:165-207)
# ----------------------------------------------------------------------------------------------------------------------
KB = 8


def _worker_batch(rank, world, port, out_dir, steps, KB=KB):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import pytest as _pt

    import oracle_backend
    from gaussianeditor_amd.multiview import GradBucket, multiview_batch_step, views_of_rank

    mpatch = _pt.MonkeyPatch()
    oracle_backend.install(mpatch)
    try:
        mine = list(views_of_rank(KB, world, rank))
        cases = [make_case(P, W, H, seed=5, s0=0.05, view=v, nviews=KB) for v in mine]
        sc = cases[0]["sc"]
        params = {k: sc[k] for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        bucket = GradBucket(P, 16, "cpu", sh_exchange="rgb")
        for step in range(steps):  # (a second step runs on the speculated message size of the first)
            if step == 2:  # a third step whose speculation is far too small: it must notice and re-run in the exact form
                bucket._cap_hint = 16
            Gs = [seed_gradient(H, W, 100 + v + 1000 * step) * H * W for v in mine]
            colors, radii, depths, grads = multiview_batch_step([settings(c, "cpu") for c in cases], params, Gs, bucket)
        assert len(colors) == len(mine) and bucket.last_route == "rows" and len(bucket.last_counts) == KB
        assert bucket.last_exchange["speculated"] == (steps == 2)  # first step and overflowing step: exact; second: speculated
        assert bucket.last_exchange["views"] == KB and bucket.last_exchange["views_local"] == len(mine)
        np.savez(os.path.join(out_dir, f"batch_w{world}_r{rank}{'' if KB == 8 else f'_k{KB}'}.npz"), flat=_segments(bucket), sh=bucket.views["sh"].numpy(),
                 radii=radii.numpy(), counts=np.array(bucket.last_counts))
    finally:
        mpatch.undo()
        if world > 1:
            dist.destroy_process_group()


def test_views_of_rank_deals_contiguous_ascending_blocks():
    from gaussianeditor_amd.multiview import views_of_rank

    for n in (1, 2, 4, 8):
        dealt = [v for r in range(n) for v in views_of_rank(8, n, r)]
        assert dealt == list(range(8))
    with pytest.raises(ValueError):
        views_of_rank(8, 3, 0)
    with pytest.raises(ValueError):
        views_of_rank(2, 4, 0)


@pytest.mark.parametrize("steps", [1, 2, 3])
def test_batch_of_eight_views_on_1_2_4_ranks_is_bit_identical_to_the_single_process_loop(oracle, tmp_path, steps):
    out = {}
    for world in (1, 2, 4):
        if world == 1:
            _worker_batch(0, 1, 0, str(tmp_path), steps)
        else:
            mp.spawn(_worker_batch, args=(world, _free_port(), str(tmp_path), steps), nprocs=world, join=True)
        rs = [np.load(tmp_path / f"batch_w{world}_r{r}.npz") for r in range(world)]
        for r in rs[1:]:  # replicas agree bit for bit
            for k in ("flat", "sh", "radii", "counts"):
                assert np.array_equal(rs[0][k], r[k]), (world, k)
        out[world] = rs[0]
    for world in (2, 4):  # ... and with the single process that rendered all eight views itself
        for k in ("flat", "sh", "radii", "counts"):
            assert np.array_equal(out[1][k], out[world][k]), (world, k)
    # the single process's sums ARE the reference's loop: per-view gradients added one after the other, view 0 first
    step = steps - 1
    tot = sh = rad = None
    for v in range(KB):
        case = make_case(P, W, H, seed=5, s0=0.05, view=v, nviews=KB)
        f = oracle_forward(oracle, case)
        g = oracle_backward(oracle, case, f, seed_gradient(H, W, 100 + v + 1000 * step) * H * W)
        flat = np.concatenate([g[k].reshape(-1) for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dmeans2D",
                                                          "dL_dopacity")])
        tot = flat if tot is None else tot + flat
        sh = g["dL_dsh"] if sh is None else sh + g["dL_dsh"]
        rad = f["radii"] if rad is None else np.maximum(rad, f["radii"])
    assert np.array_equal(out[1]["flat"], tot)
    assert np.array_equal(out[1]["sh"], sh.reshape(out[1]["sh"].shape))
    assert np.array_equal(out[1]["radii"], rad)


def test_eight_rank_weak_step_matches_the_single_process_sums(oracle, tmp_path):
    """The weak-scaling path of BASELINE configs[3] at its real rank count (the driver's 8-GPU run is the first time hardware
    sees world_size 8): `multiview_step` with one view per rank on EIGHT gloo ranks -- the whole step as bench.py runs it,
    route chosen from the gathered counts -- leaves every replica with the same bits, equal to a single process that renders
    the eight views one after the other and sums."""
    world, P8, s0 = 8, 1203, 0.01  # (P % 4 != 0: segment padding; small splats: the touched-rows route)
    mp.spawn(_worker_rgb, args=(world, _free_port(), str(tmp_path), "auto", s0, P8), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"rgb_rank{r}.npz") for r in range(world)]
    for r in rs[1:]:
        for k in ("flat", "sh", "radii"):
            assert np.array_equal(rs[0][k], r[k]), k
    tot, sh, rad = None, None, None
    for v in range(world):
        case = make_case(P8, W, H, seed=5, s0=s0, view=v, nviews=world)
        f = oracle_forward(oracle, case)
        g = oracle_backward(oracle, case, f, seed_gradient(H, W, 100 + v) * H * W)
        flat = np.concatenate([g[k].reshape(-1) for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dopacity")])
        tot = flat if tot is None else tot + flat
        sh = g["dL_dsh"] if sh is None else sh + g["dL_dsh"]
        rad = f["radii"] if rad is None else np.maximum(rad, f["radii"])
    assert all(bool(r["rows_route"][0]) == bool(rs[0]["rows_route"][0]) for r in rs)
    assert rel_err(rs[0]["flat"], tot) < 1e-6 and np.array_equal(rs[0]["radii"], rad)
    if bool(rs[0]["rows_route"][0]):  # rows added view after view to zeros: the single process's sums, bit for bit
        assert np.array_equal(rs[0]["flat"], tot)
    assert np.array_equal(rs[0]["sh"], sh.reshape(rs[0]["sh"].shape))


def test_densify_synchronized_on_eight_ranks(tmp_path):
    """`densify_synchronized` at world_size 8: eight differently seeded replicas split identically, their own random streams
    continue undisturbed, and a diverged replica is refused on every rank."""
    world = 8
    mp.spawn(_worker_densify, args=(world, _free_port(), str(tmp_path), True), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"dz_1_{r}.npz") for r in range(world)]
    assert all(r["same"][0] and r["rng_kept"][0] for r in rs)
    assert all(np.array_equal(rs[0]["xyz"], r["xyz"]) for r in rs[1:]) and rs[0]["xyz"].shape[0] > 200
    assert len({float(r["before"][0]) for r in rs}) == world  # eight different generators went in
    assert all(np.load(tmp_path / f"dzr_{r}.npz")["raised"][0] for r in range(world))


def test_batch_of_sixteen_views_on_eight_ranks_is_bit_identical_to_the_single_process_loop(oracle, tmp_path):
    """`bench.py --gpus 8 --views 16` in numbers: the fixed batch of 16 views dealt to EIGHT gloo ranks, two per rank (each packed
    into its message before the next one's backward overwrites the bucket), one 8-way all-gather -- every replica the same bits,
    equal to one process that renders all sixteen, over two steps (the second on a speculated message size)."""
    K16, world, steps = 16, 8, 2
    _worker_batch(0, 1, 0, str(tmp_path), steps, K16)
    mp.spawn(_worker_batch, args=(world, _free_port(), str(tmp_path), steps, K16), nprocs=world, join=True)
    one = np.load(tmp_path / f"batch_w1_r0_k{K16}.npz")
    assert len(one["counts"]) == K16
    for r in range(world):
        z = np.load(tmp_path / f"batch_w{world}_r{r}_k{K16}.npz")
        for k in ("flat", "sh", "radii", "counts"):
            assert np.array_equal(one[k], z[k]), (r, k)
