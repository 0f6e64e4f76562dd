"""The multi-view gradient exchange on the RCCL backend ("nccl"), every route, checked against a single-process sum.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/rccl_exchange_check.py [P_gaussians]

Rank r renders view r of an N-view ring; for each route of gaussianeditor_amd.multiview
    direct-dense   (SH gradient inside the all-reduced bucket)
    direct-sparse  (touched-mask MAX all-reduce + packed union rows)
    rgb-dense      (colour-gradient all-gather + 56 B/Gaussian all-reduce + SH gradient rebuilt)
    rows           (touched-row messages all-gathered + one ordered accumulate kernel)
the exchanged gradients must equal the sum, over the N views, of the gradients a single process computes for them
(every rank renders all N views locally for that).  With N = 1 the collectives still run (force_exchange), so a
single-GPU box proves every collective call of the step against the real backend.  Exit status 0 = all routes agree.
"""
import math
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks need {world} GPUs, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianeditor_amd.multiview import GradBucket, multiview_step, render_view_grads
    from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene

    P = int(sys.argv[1]) if len(sys.argv) > 1 else 20001  # P % 4 != 0 on purpose: padded bucket segments
    W, H, M = 320, 192, 16
    sc = synth_scene(P, seed=3, s0=0.03)
    cams = ring_cameras(max(world, 2), W, H)
    params = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}

    def settings(v):
        cam = cams[v]
        return GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), sc["bg"].to(dev), 1.0,
                                             cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                             cam.camera_center.to(dev), False, False)

    def G(v):
        return (seed_gradient(H, W, 50 + v) * H * W).to(dev)

    # single-process reference: the views one after the other, gradients accumulated in view order
    names = ("means3D", "sh", "scales", "rotations", "means2D", "opacities")
    want, want_radii = None, None
    for v in range(world):
        _, radii, _, g = render_view_grads(settings(v), params["xyz"], params["opacity"], params["features"],
                                           params["scaling"], params["rotation"], G(v))
        want = {k: g[k].clone() for k in names} if want is None else {k: want[k] + g[k] for k in names}
        want_radii = radii.clone() if want_radii is None else torch.maximum(want_radii, radii)

    failures = []
    routes = (("direct", dict(sparse=False), "dense"), ("direct", dict(sparse=True), "sparse"),
              ("rgb", dict(rows=False), "dense"), ("rgb", dict(rows=True), "rows"))
    for exch, kw, expect in routes:
        bucket = GradBucket(P, M, dev, sh_exchange=exch)
        color, radii, depth, grads = multiview_step(settings(rank), params, G(rank), bucket, force_exchange=True, **kw)
        torch.cuda.synchronize(dev)
        route = bucket.last_route
        if route != expect and not (expect == "sparse" and route == "dense"):  # (a union above the threshold falls back)
            failures.append(f"{exch}/{kw}: route {route}, expected {expect}")
        for k in names:
            got, ref = bucket.views[k], want[k].reshape(bucket.views[k].shape)
            scale = float(ref.abs().max()) or 1.0
            err = float((got - ref).abs().max()) / scale
            # (the single-process sum comes from SEPARATE backward runs: the blend's float atomics re-associate from run
            #  to run, ~1e-7; the bit-for-bit properties of the ordered routes are asserted across the replicas below)
            tol = 2e-6
            if err > tol:
                failures.append(f"{exch}/{route}: {k} differs by {err:.2e} (tol {tol})")
            if world > 1 and (route == "rows" or (k == "sh" and exch == "rgb")):
                # ordered accumulation: every replica holds the same bits
                mine = got.detach().clone().contiguous()
                theirs = mine.clone()
                dist.broadcast(theirs, src=0)
                if not torch.equal(mine, theirs):
                    failures.append(f"{exch}/{route}: {k} is not bit-identical on rank {rank} and rank 0")
        if not torch.equal(radii, want_radii):
            failures.append(f"{exch}/{route}: batch-max radii differ")
        # every rank says what IT saw (first contact with an 8-GPU node: the lines must agree with each other and with
        # profiles/README.md, "what to run on the 8-GPU node")
        from gaussianeditor_amd.multiview import replicas_identical

        same = replicas_identical([bucket.views[k] for k in names])
        if not same and route in ("rows",):
            failures.append(f"{exch}/{route}: the replicas' gradients are not bit-identical")
        lx = bucket.last_exchange or {}
        print(f"[rccl_exchange_check rank {rank}] world_size seen by {dist.get_backend()}: {dist.get_world_size()}, device {dev}, "
              f"exchange {exch} -> route {route} (expected {expect}), sent {int(lx.get('bytes_sent', 0))} B received "
              f"{int(lx.get('bytes_received', 0))} B, replicas identical: {same}, "
              f"{'ok' if not failures else 'FAILURES: ' + '; '.join(failures)}", flush=True)
    # sparse + persistent rows (GradBucket(sparse_rows=True, persistent_rows=True)) over several steps on ONE bucket: the
    # exchange writes only the rows some view touched, the backward rewrites a zero row only if it does not hold zeros
    # already; after every step the valid rows hold the batch sums, the others are zero (SH rows: not written at all)
    bucket = GradBucket(P, M, dev, sh_exchange="rgb", sparse_rows=True, persistent_rows=True)
    for step in range(4):
        shift = step % max(world, 2)
        view = (rank + shift) % max(world, 2)
        ref = None
        for v in range(world):
            vv = (v + shift) % max(world, 2)
            _, _, _, g = render_view_grads(settings(vv), params["xyz"], params["opacity"], params["features"],
                                           params["scaling"], params["rotation"], G(vv))
            ref = {k: g[k].clone() for k in names} if ref is None else {k: ref[k] + g[k] for k in names}
        multiview_step(settings(view), params, G(view), bucket, force_exchange=True, rows=True)
        torch.cuda.synchronize(dev)
        valid = bucket.row_valid.bool()
        for k in names:
            got, want_k = bucket.views[k].reshape(P, -1), ref[k].reshape(P, -1)
            scale = float(want_k.abs().max()) or 1.0
            if float((got[valid] - want_k[valid]).abs().max()) / scale > 2e-6:
                failures.append(f"sparse+persistent step {step}: {k} valid rows differ")
            if bool(want_k[~valid].any()):
                failures.append(f"sparse+persistent step {step}: {k} has a non-zero row that is not marked valid")
            if k != "sh" and bool(got[~valid].any()):
                failures.append(f"sparse+persistent step {step}: {k} invalid rows are not zero")
        if not (0 < int(valid.sum()) < P):
            failures.append(f"sparse+persistent step {step}: {int(valid.sum())} valid rows of {P}")
    if rank == 0:
        print("[rccl_exchange_check] sparse + persistent rows over 4 steps: ok" if not failures else
              f"[rccl_exchange_check] failures so far: {failures}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if failures:
        print("\n".join(failures), file=sys.stderr)
        raise SystemExit(1)
    if rank == 0:
        print(f"rccl_exchange_check: world_size {world}, backend nccl, P {P}: all routes agree with the single-process sum")


if __name__ == "__main__":
    main()
