"""Round 4 GPU tests: boundary hardening of the C ABI, K views on N ranks, the bench line's extra blocks."""
import ctypes
import math
import os
import sys

import numpy as np
import pytest
import torch

from helpers import make_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _preprocess(L, case, flags, geom=None):
    from gaussianeditor_amd import _native

    sc, cam = case["sc"], case["cam"]
    P, W, H = sc["xyz"].shape[0], case["W"], case["H"]
    d = lambda t: t.to(DEV).contiguous()  # noqa: E731
    t = dict(xyz=d(sc["xyz"]), sca=d(sc["scaling"]), rot=d(sc["rotation"]), op=d(sc["opacity"]), sh=d(sc["features"]),
             view=d(cam.world_view_transform), proj=d(cam.full_proj_transform), cp=d(cam.camera_center))
    gb, _, ib = _native.scratch_sizes(P, 0, W, H)
    if geom is None:
        geom = torch.empty(gb, dtype=torch.uint8, device=DEV)
    radii = torch.empty(P, dtype=torch.int32, device=DEV)
    counts = (ctypes.c_int64 * 2)()
    s = torch.cuda.current_stream().cuda_stream
    p = lambda x: x.data_ptr()  # noqa: E731
    st = L.gsr_preprocess(s, P, 3, 16, p(t["xyz"]), p(t["sca"]), 1.0, p(t["rot"]), p(t["op"]), p(t["sh"]), None, None,
                          p(t["view"]), p(t["proj"]), p(t["cp"]), W, H, case["tfx"], case["tfy"], 0, 0, flags, p(radii),
                          p(geom), counts)
    assert st == 0
    return t, geom, radii


def test_backward_on_a_forward_only_state_is_refused():
    """gsr_preprocess(GSR_FLAG_FORWARD_ONLY) leaves out what only K8+K9 reads; a backward on that buffer must return
    GSR_ERR_BAD_ARGUMENT instead of reading uninitialised memory -- and be accepted again once a full preprocess has
    rewritten the buffer."""
    from gaussianeditor_amd import _native

    L = _native.lib()
    case = make_case(3000, 128, 96, seed=2)
    P, W, H = 3000, 128, 96
    t, geom, radii = _preprocess(L, case, 8)
    z = lambda *shape: torch.zeros(shape, device=DEV)  # noqa: E731
    g = dict(acc=z(P, 16), m2=z(P, 3), op=z(P, 1), m3=z(P, 3), cov=z(P, 6), sh=z(P, 16, 3), sc=z(P, 3), rot=z(P, 4))
    p = lambda x: x.data_ptr()  # noqa: E731
    s = torch.cuda.current_stream().cuda_stream

    def pbw(geom_):
        return L.gsr_preprocess_backward(s, P, 3, 16, W, H, p(t["xyz"]), p(t["sh"]), p(t["sca"]), 1.0, p(t["rot"]), None,
                                         p(t["view"]), p(t["proj"]), p(t["cp"]), case["tfx"], case["tfy"], p(radii), p(geom_),
                                         p(g["acc"]), p(g["m2"]), p(g["op"]), None, p(g["m3"]), p(g["cov"]), p(g["sh"]), p(g["sc"]),
                                         p(g["rot"]), 0)

    assert pbw(geom) == -1
    # a copy of the state elsewhere is (documented) not recognised, and another buffer is unaffected
    t2, geom2, radii2 = _preprocess(L, case, 0)
    assert pbw(geom2) == 0
    # the same buffer, preprocessed in full: accepted
    _preprocess(L, case, 0, geom=geom)
    assert pbw(geom) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g["sh"]).all())


def _bench(args, env_extra=None, launcher=None, timeout=900):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    # the contract: ONE JSON line on stdout and nothing else (RCCL's version banner, torch warnings, ... go to stderr)
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]
    return json.loads(lines[0])


def test_batch_step_on_the_gpu_equals_the_sequential_single_process_sums(monkeypatch):
    """multiview_batch_step with K = 3 local views (no communication): the batch-summed gradients are bit-identical to
    accumulating the three views' own gradients one after the other (the reference's loop, GassuianEditor.py:165-207), for
    the five small tensors AND the SH gradient rebuilt from the colour gradients; radii = the batch maximum.  (The views'
    gradients are captured as the step packs them: K7 accumulates with float atomics, so a second render of the same view
    differs in the last bits.)  Two steps: the second runs on the message size the first one speculated."""
    from gaussianeditor_amd import multiview as mv
    from gaussianeditor_amd.multiview import GradBucket, multiview_batch_step
    from helpers import seed_gradient, settings

    P, W, H, K = 20000, 256, 192, 3
    cases = [make_case(P, W, H, seed=4, s0=0.02, view=v, nviews=K) for v in range(K)]
    sc = cases[0]["sc"]
    d = lambda t: t.to(DEV).contiguous()  # noqa: E731
    params = {k: d(sc[k]) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
    bucket = GradBucket(P, 16, torch.device(DEV), sh_exchange="rgb")
    seen = []
    orig_pack = mv._C.view_message_pack

    def spy(plan, grads5, rgb, campos, cap, message):
        seen.append(([g.clone() for g in grads5], rgb.clone(), campos.clone()))
        return orig_pack(plan, grads5, rgb, campos, cap, message)

    monkeypatch.setattr(mv._C, "view_message_pack", spy)
    for step in range(2):
        seen.clear()
        Gs = [d(seed_gradient(H, W, 10 * step + v)) for v in range(K)]
        colors, radii, depths, grads = multiview_batch_step([settings(c, DEV) for c in cases], params, Gs, bucket)
        assert bucket.last_route == "rows" and len(bucket.last_counts) == K and bucket.last_exchange["bytes_sent"] == 0
        assert len(seen) == K and len(colors) == K
        tot = [t.clone() for t in seen[0][0]]
        sh = mv._C.sh_grad_compose(params["xyz"], seen[0][2].view(1, 3), seen[0][1].view(1, P, 3), 3, 16)
        for g5, rgb, campos in seen[1:]:
            tot = [a + b for a, b in zip(tot, g5)]
            sh = sh + mv._C.sh_grad_compose(params["xyz"], campos.view(1, 3), rgb.view(1, P, 3), 3, 16)
        torch.cuda.synchronize()
        for name, t in zip(mv._ROW_SEGS, tot):
            assert torch.equal(bucket.views[name], t.view_as(bucket.views[name])), (step, name)
        assert torch.equal(bucket.views["sh"], sh), step
        touched = [int((torch.cat([g.reshape(P, -1) for g in g5] + [rgb], 1) != 0).any(1).sum()) for g5, rgb, _ in seen]
        assert all(0 < t <= c <= P for t, c in zip(touched, bucket.last_counts))  # (the blend-level plan is a superset)
        # radii: the maximum over the views' own radii
        from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer
        rad = None
        with torch.no_grad():
            for c in cases:
                _, r, _ = GaussianRasterizer(settings(c, DEV))(params["xyz"], torch.zeros_like(params["xyz"]), params["opacity"],
                                                                shs=params["features"], scales=params["scaling"],
                                                                rotations=params["rotation"])
                rad = r.clone() if rad is None else torch.maximum(rad, r)
        assert torch.equal(radii, rad)


def test_default_bench_line_carries_extra_configs_and_both_roofline_fractions():
    """VERDICT r03 items 4 + 5: the default line (headline workload) reports the compulsory-byte fraction next to section
    8(d)'s, pixel-instances/s for the blend kernel that dominates, and an `extra_configs` block with C2 / C3 / C5."""
    line = _bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline"], timeout=1200)
    rf = line["roofline"]
    assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_8d", "frac_compulsory", "frac_counter",
                       "valu_issue_frac"}
    # frac = the contract's figure (section 8(d) bytes / time / peak); the compulsory-byte model never exceeds it
    assert 0 < rf["frac_compulsory"] <= rf["frac"] + 1e-12 and rf["frac"] == rf["frac_8d"] <= 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    if rf["kernel"].startswith("blend"):
        assert rf["pixel_instances_per_s"] > 1e9
    if rf["traffic"] is not None:  # (counters are quoted only while their source hash matches this build)
        # round 6: the issue fraction is priced at the MEASURED rate of K7's own group body (2.79 cycles per VALU instruction and
        # SIMD), the 4-cycle figure of rounds 3-5 stays next to it
        assert 0 < rf["frac_counter"] <= 1 and 0 < rf["valu_issue_frac"] < rf["valu_issue_frac_4cycle"] <= 1
        assert 0 < line["hbm_fraction_train_iter_counter"] <= 1
    if rf["kernel"] == "blend_backward":  # K7's critical path from its own per-item cycle counts
        assert 0.3 < rf["critical_item_frac"] <= 1.0 and 0.3 < rf["mean_workgroup_frac"] <= 1.0 and rf["k7_items"] > 1000
    assert 0 < line["hbm_fraction_train_iter"] <= 1
    ex = line["extra_configs"]
    assert ex["C2_synth6M_1080p_forward"]["visible"] == 6_000_000 and 0.3 < ex["C2_synth6M_1080p_forward"]["forward_ms"] < 5
    assert set(ex["C2_synth6M_1080p_forward"]["stage_ms"]) == {"preprocess", "bin", "blend_forward"}
    assert 0 < ex["C2_synth6M_1080p_forward"]["roofline"]["frac_compulsory"] <= ex["C2_synth6M_1080p_forward"]["roofline"]["frac"] <= 1
    v2 = ex["synth_v2_1M_1080p"]
    assert v2["visible"] > 600_000 and 0.3 < v2["train_ms_per_step"] < 10 and set(v2["stage_ms"]) == set(line["stage_ms"])
    v8 = ex["views8_one_gpu"]
    assert v8["pipelined"]["min"] <= v8["pipelined"]["view_iters_per_s_median"] <= v8["pipelined"]["max"] and v8["serial"]["min"] > 100
    # (functional only: which form is faster, and by how much, is a measurement -- the line reports it as
    # `pipelined_over_serial`, bench.py; a wall-clock inequality here would fail on a shared or throttled GPU for no defect)
    assert v8["pipelined"]["min"] > 100 and v8["pipelined_over_serial"] > 0
    pr = ex["persistent_rows_1M_1080p"]
    # (functional bounds: which of two timings is smaller is a measurement the line reports, not something a test on a shared
    #  box can assert -- one host hiccup in a 30-step window is 2 ms per step; bench.py reports medians of three windows)
    assert 0.3 < pr["train_ms_per_step"] < 2.0 * line["ms_per_step"]
    assert 0.2 < ex["C3_edit_loop_512_1M"]["ms_per_step"] < 10 and 0.05 < ex["C5_apply_weights_12views_512_1M"]["ms_per_view"] < 5
    # round 6: the unmodified double render is served by view reuse (the second render: the blend kernel alone); stage times and
    # a roofline for the two 512 x 512 entries; the headline through render() + autograd
    c3 = ex["C3_edit_loop_512_1M"]
    assert c3["view_reuse_hits"] >= 30 and c3["ms_per_step"] < 1.5 * ex["C3_edit_loop_512_1M_no_reuse"]["ms_per_step"]
    assert set(c3["stage_ms"]) == set(line["stage_ms"]) and 0 < c3["roofline"]["frac_compulsory"] <= 1
    assert set(ex["C5_apply_weights_12views_512_1M"]["stage_ms"]) == {"preprocess", "bin", "trace_weights"}
    assert ex["C3_edit_loop_512_1M_reference_model"]["reuse"]["compare_launches"] >= 30
    l2 = ex["headline_via_render_l2"]
    assert 0.3 < l2["ms_per_step"] < 10
    assert ex["seconds"] < 300


def test_bench_stdout_is_one_json_line_even_when_rccl_is_initialised():
    """RCCL prints a version banner to the process's stdout at exit (C stdio): with a process group up (--force-exchange: the
    one-rank RCCL exchange) the line must still be the only thing on stdout; the pre-warm steps are reported."""
    line = _bench(["--gaussians", "50000", "--width", "640", "--height", "368", "--steps", "3", "--warmup", "1", "--prewarm", "7",
                   "--no-cpu-baseline", "--force-exchange"])
    assert line["multi_gpu"]["backend"] == "nccl" and line["prewarm_steps"] == 7 and line["steps"] == 3
    assert line["step_ms_gpu"]["first"] > 0 and line["step_ms_gpu"]["max"] >= line["step_ms_gpu"]["median"]


def test_bench_fixed_batch_of_views_on_one_gpu_and_on_two_ranks_sharing_it():
    """`bench.py --views K`: the fixed batch dealt to the ranks (strong scaling).  K = 4 on one rank, and K = 4 on two ranks
    that share the one GPU over gloo (K > N: two views per rank, one all-gather per step) -- as the driver launches it."""
    import socket

    small = ["--gaussians", "50000", "--width", "640", "--height", "368", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    one = _bench(["--views", "4"] + small)
    assert one["scaling"] == "strong" and one["config"]["views_per_step"] == 4 and one["config"]["views_per_rank"] == 4
    assert one["config"]["grad_exchange_route"] == "rows" and len(one["config"]["grad_exchange_rows_per_view"]) == 4
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    two = _bench(["--gpus", "2", "--views", "4"] + small, env_extra={"GSR_BENCH_SHARED_GPU": "1"},
                 launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                           "127.0.0.1", "--master-port", str(port)])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["views_per_rank"] == 2
    assert two["config"]["grad_exchange_rows_per_view"] == one["config"]["grad_exchange_rows_per_view"]  # the same four views
    mg = two["multi_gpu"]
    assert mg["world_size"] == 2 and mg["route"] == "rows" and len(mg["per_rank"]["step_ms_gpu"]) == 2
    assert all(b > 0 for b in mg["per_rank"]["bytes_sent_per_step"]) and all(t > 0 for t in mg["per_rank"]["exchange_ms_gpu"])
    assert mg["per_rank"]["bytes_received_per_step"][0] == 2 * mg["per_rank"]["bytes_sent_per_step"][0]


def test_backward_list_segments_match_the_oracle_when_forced_on_small_scenes():
    """The backward walks a deep tile's list as independent segments that start from the forward's checkpoints (round 4).  On
    the deep-tile and the 6 M workloads of the three-way parity test it is on by default (stride 512; views with short lists
    -- the headline view -- run without: gsr_capi.hip checkpoint_chunks); here a checkpoint every 64 list positions and no work
    threshold cut nearly every tile of the SMALL parity scenes into segments: forward / backward / apply_weights parity against the oracle in a fresh process with those knobs."""
    import subprocess

    # (strides below 4 chunks are clamped outside GSR_CK_DEBUG=1, round 5: they are inside the per-row parity bar)
    # (round 6: sixteen checkpoint slots per tile with the fine stride, eight with the coarse one -- GSR_CK_SLOTS: both here)
    for slots in ("16", "8", "5"):
        env = dict(os.environ, GSR_CK_CHUNKS="1", GSR_CK_DEBUG="1", GSR_BWD_SEG="1", GSR_CK_SLOTS=slots)
        p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k",
                            "backward_vs_oracle or backward_precomp or forward_all_stages or edge_geometries or backward_twice"],
                           capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
        assert p.returncode == 0, (slots, (p.stdout + p.stderr)[-3000:])
        assert " passed" in p.stdout
    # The three-way test's PER-ROW bar (a Gaussian's own gradient within 1e-3 of itself for all but 0.1 % of the rows) is
    # where a segment's start state shows: the colour behind a checkpoint is the difference of two binary32 accumulators of the
    # forward, so a Gaussian that only ever appears behind transmittance ~1e-3 gets its gradient to ~1e-4 instead of ~1e-6.
    # Measured with every tile cut (GSR_BWD_SEG=1): stride 64 / 128 leave 12-20 rows of dL_dopacity outside that bar where
    # 5-11 are allowed (every tensor-max bar holds); stride 256 and the default 512 pass all six cases incl. 1 M, 6 M and the
    # deep-tile scene.
    env = dict(os.environ, GSR_CK_CHUNKS="4", GSR_BWD_SEG="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_round2.py"), "-k",
                        "three_way_parity"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
