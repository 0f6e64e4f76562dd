#!/usr/bin/env python3
"""bench.py -- headline benchmark of the rasterizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--views V]
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or started plainly:
     without WORLD_SIZE in the environment `--gpus N` re-executes itself under torch.distributed.run with N ranks on
     127.0.0.1, and refuses to run if fewer than N GPUs are visible -- it never reports n_gpus = 1 for --gpus N)

Workload (BASELINE.json metric, config C4): synth-v1 scene, 1,000,000 Gaussians, SH degree 3
(M = 16), 1920x1080, ring-v1 cameras.  Default: rank r renders view r (one view per GPU, weak scaling).
`--views V` (V a multiple of N) fixes the batch instead: the V views of BASELINE configs[3] are dealt to the ranks in
contiguous blocks, every rank renders V / N of them and packs one touched-rows message per view, ONE all-gather per step,
the sums formed in ascending view order (gaussianeditor_amd/multiview.py: multiview_batch_step) -- strong scaling of the
reference's own batch loop (threestudio/systems/GassuianEditor.py:165-207), bit-identical to it for every N.
A step = forward + backward of this rank's view(s) through the drop-in L1 API + the gradient
exchange (SUM over the views, MAX over radii); inputs are resident in HBM.

One JSON line is printed by rank 0: the train rate is `value`; the forward-only rate (renders/s, Mpixels/s), the
per-stage GPU times, the roofline block of the dominant stage (section 8(d) bytes, the compulsory-byte model, counter
traffic and VALU issue share), the other BASELINE configurations (`extra_configs`: 6 M forward, the 512^2 edit loop,
tracing) and the CPU baselines ride along in the same object; N > 1 lines add `multi_gpu` (per-rank step times, exchange
time and bytes, route, RCCL version).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the host driver of these boxes only supports dmabuf IPC: without this RCCL's peer access between ranks fails with
# `hipIpcGetMemHandle: invalid argument` (exported in the image already; kept for a launcher that starts from a clean environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MAX_CLOCK_HZ = 2.4e9   # MI355X_MICROARCH.md: max engine clock 2400 MHz (the issue ceiling is priced at it: DVFS runs these kernels
#                        at ~2.1-2.3 GHz, so the fraction understates the share of the cycles actually clocked)
STAGES = ("preprocess", "bin", "blend_forward", "blend_backward", "preprocess_backward")
# Issue ceiling of the blend kernels' instruction mix, MEASURED (round 6; rounds 3-5 priced a VALU instruction at 4 cycles per
# SIMD, the review of round 5 at 2.1-2.4 from dependent-chain microbenchmarks -- both wrong for this mix): K7's own 4-entry
# group body as hipcc emits it (299 VALU + 15 SALU + 24 LDS instructions), replayed without memory, barriers or list walk at
# 4 waves per SIMD on every SIMD of the chip, takes 834.6 cycles per group and SIMD (tools/microbench/k7_group_replay.py,
# profiles/r06_a_k7_replay.md) = 2.79 cycles per VALU instruction with the scalar / LDS instructions that accompany it.
ISSUE_CYCLES_PER_VALU = 834.6 / 299.0


def algorithmic_bytes(P, V, R, N, T, M):
    """compulsory HBM bytes per stage for one view, as the survey wrote them (the contract figure:
    per-INSTANCE gather terms for the blend stages)."""
    return {
        "preprocess": (48 + 12 * M) * P + 67 * V,            # K1+K2: params in, radii + survivor state out
        "bin": 36 * R + 16 * T,                              # K3 emit 12 + K4 sort 24 per instance, K5 ranges
        "blend_forward": 44 * R + 24 * N,                    # K6: list 4 + gather 40 per instance; 24 B per pixel out
        "blend_backward": 20 * N + 76 * R,                   # K7: pixel inputs 20; per instance 4 + 36 + 36
        "preprocess_backward": (103 + 12 * M) * V + (56 + 12 * M) * P,  # K8+K9
    }


def compulsory_bytes(P, V, R, N, T, M):
    """The same model with the blend stages' per-instance terms put right: a Gaussian's 48-byte gather
    record has to cross the HBM interface once per VIEW, not once per (tile, Gaussian) instance, and its reduced gradient
    (its accumulator row: nine floats that matter of a 64-byte line) is written once per visible Gaussian -- an instance
    costs only its 4-byte list entry.  With section 8(d)'s figures a deep-tile scene (R = 49 x V) reported a fraction above
    1 of the HBM peak.  (The 44 is kept from rounds 3-5 -- 11 floats in four arrays -- so that the fraction stays comparable
    across rounds; the row the kernels move since round 6 is 64 bytes.)"""
    b = algorithmic_bytes(P, V, R, N, T, M)
    b["blend_forward"] = 4 * R + 48 * V + 24 * N
    b["blend_backward"] = 20 * N + 4 * R + (48 + 44) * V
    return b


def self_launch(n: int) -> None:
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run` with N ranks (one per GPU)."""
    import socket

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible; refusing to run a smaller job under that label")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def percentile(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    i = q * (len(xs) - 1)
    lo, hi = int(math.floor(i)), int(math.ceil(i))
    return xs[lo] + (xs[hi] - xs[lo]) * (i - lo)


def csrc_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: what profiles/traffic_latest.json is stamped with."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "gaussianeditor_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def workload_key(P, W, H, s0, scene="v1") -> str:
    """Key of a workload in profiles/traffic_latest.json (tools/collect_counters.py writes the same)."""
    if scene == "v2":
        return f"synth-v2:{int(P)}:{int(W)}x{int(H)}"
    return f"synth-v1:{int(P)}:{int(W)}x{int(H)}:s0={float(s0):g}"


def stage_times(dev, params, rs, G, D, flags, iters, backward=True, info=None):
    """Per-stage GPU time of one view through the individual C-ABI calls (HIP events on the launch stream), plus what the
    byte models need: num_rendered R, visible V, and the pixel-instances of the view (sum over tiles of pixels x list length:
    SURVEY.md section 8(d) "flops (secondary)")."""
    import ctypes

    from gaussianeditor_amd import _native

    L = _native.lib()
    s = torch.cuda.current_stream(dev)
    sp = s.cuda_stream
    P, M = int(params["xyz"].shape[0]), int(params["features"].shape[1])
    H, W = int(rs.image_height), int(rs.image_width)
    tfx, tfy = float(rs.tanfovx), float(rs.tanfovy)
    names = STAGES if backward else STAGES[:3]
    acc = {k: 0.0 for k in names}
    p = lambda t: t.data_ptr()  # noqa: E731
    op_flat = params["opacity"].contiguous()
    R = V = pix_inst = 0
    # the blend backward's accumulator table, one 64-byte row per Gaussian (include/gsr.h: GSR_ACC_*), kept across the
    # iterations as the binding keeps it across backwards: zero on entry, left zero again by K8+K9 (GSR_FLAG_ACC_SELF_CLEAN);
    # dL_dmeans2D / dL_dopacity leave through K8+K9
    acc_rows = torch.zeros(P * _native.ACC_ROW, device=dev) if backward else None
    for it in range(iters + 2):
        gb, _, ib = _native.scratch_sizes(P, 0, W, H)
        geom = torch.empty(gb, dtype=torch.uint8, device=dev)
        img = torch.empty(ib, dtype=torch.uint8, device=dev)
        radii_t = torch.empty(P, dtype=torch.int32, device=dev)
        color = torch.empty((3, H, W), device=dev)
        depth = torch.empty((1, H, W), device=dev)
        if backward:
            d_m2, d_op = torch.empty(P * 3, device=dev), torch.empty(P, device=dev)
            d_m3, d_cov = torch.empty(P * 3, device=dev), torch.empty(P * 6, device=dev)
            d_sh, d_sc, d_rot = torch.empty(P * M * 3, device=dev), torch.empty(P * 3, device=dev), torch.empty(P * 4, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        Rc = (ctypes.c_int64 * 2)()
        ev[0].record(s)
        _native.check("pre", L.gsr_preprocess(sp, P, D, M, p(params["xyz"]), p(params["scaling"]), 1.0, p(params["rotation"]),
                                              p(op_flat), p(params["features"]), None, None, p(rs.viewmatrix), p(rs.projmatrix),
                                              p(rs.campos), W, H, tfx, tfy, 0, 0, flags | (0 if backward else 8), p(radii_t),
                                              p(geom), Rc))
        ev[1].record(s)
        R, Gi = int(Rc[0]), int(Rc[1])
        _, bb, _ = _native.scratch_sizes(P, R, W, H, Gi)
        binning = torch.empty(bb, dtype=torch.uint8, device=dev)
        _native.check("bin", L.gsr_bin(sp, P, R, Gi, W, H, p(geom), p(binning), p(img)))
        ev[2].record(s)
        _native.check("fwd", L.gsr_blend_forward(sp, P, R, W, H, p(rs.bg), p(geom), p(binning), p(img), p(color), p(depth), flags))
        ev[3].record(s)
        if backward:
            _native.check("bwd", L.gsr_blend_backward(sp, P, R, W, H, p(rs.bg), p(geom), p(binning), p(img), p(G), p(acc_rows),
                                                      None, flags))
            ev[4].record(s)
            _native.check("pbw", L.gsr_preprocess_backward(sp, P, D, M, W, H, p(params["xyz"]), p(params["features"]),
                                                           p(params["scaling"]), 1.0, p(params["rotation"]), None, p(rs.viewmatrix),
                                                           p(rs.projmatrix), p(rs.campos), tfx, tfy, p(radii_t), p(geom),
                                                           p(acc_rows), p(d_m2), p(d_op), None, p(d_m3), None, p(d_sh), p(d_sc),
                                                           p(d_rot), 32))  # (no dL_dcolors / dL_dcov3D: SHs, scales + rotations)
            ev[5].record(s)
        torch.cuda.synchronize(dev)
        if it >= 2:
            for i, k in enumerate(names):
                acc[k] += ev[i].elapsed_time(ev[i + 1])
        if it == iters + 1 and backward and info is not None:
            # K7's critical path: the longest single work item / the span of the launch, from the kernel's own per-item
            # cycle counts (gsr_debug_blend_backward_profile; a 4-wave workgroup per tile, 1.6 items per workgroup on the
            # headline view -- a launch cannot end before its longest item does)
            nrec = ctypes.c_int64(0)
            L.gsr_debug_blend_backward_profile(sp, P, R, W, H, p(rs.bg), p(geom), p(binning), p(img), p(G), p(acc_rows), 1, 0,
                                               ctypes.byref(nrec))
            nwg, Tt = int(nrec.value), ((W + 15) // 16) * ((H + 15) // 16)
            rec = torch.zeros((nwg + Tt, 8), dtype=torch.int64, device=dev)
            acc_rows.zero_()
            _native.check("k7 profile", L.gsr_debug_blend_backward_profile(sp, P, R, W, H, p(rs.bg), p(geom), p(binning), p(img),
                                                                            p(G), p(acc_rows), p(rec), nwg + Tt, ctypes.byref(nrec)))
            torch.cuda.synchronize(dev)
            acc_rows.zero_()  # (the profile launch runs K7 alone: nothing cleans the table behind it)
            wg = rec[:nwg]
            livewg = wg[:, 1] > 0
            if bool(livewg.any()):
                # (the XCDs' cycle counters are not aligned with each other: a workgroup's own end - start is, the longest
                #  workgroup stands for the launch -- the persistent workgroups all start within the first microsecond)
                durs = (wg[livewg, 1] - wg[livewg, 0]).to(torch.float64)
                span = int(durs.max().item())
                items = rec[nwg:].reshape(-1, 4)[:, 0]
                longest = int(items.max().item())
                mean_wg = float(durs.mean().item())
                info["critical_item_frac"] = longest / max(span, 1)
                info["mean_workgroup_frac"] = mean_wg / max(span, 1)
                info["k7_items"] = int((items > 0).sum().item())
                info["k7_workgroups"] = int(livewg.sum().item())
        if it == iters + 1:
            V = int((radii_t > 0).sum().item())
            T = ((W + 15) // 16) * ((H + 15) // 16)
            rng = torch.empty((T, 2), dtype=torch.int32, device=dev)
            _native.check("export_image", L.gsr_debug_export_image(sp, W, H, p(img), p(rng), None, None))
            lens = (rng[:, 1] - rng[:, 0]).to(torch.int64)
            gx = (W + 15) // 16
            t = torch.arange(T, device=dev)
            pw = torch.clamp(W - (t % gx) * 16, max=16)
            ph = torch.clamp(H - (t // gx) * 16, max=16)
            pix_inst = int((lens * pw * ph).sum().item())
    return {k: acc[k] / iters for k in names}, R, V, pix_inst


def trace_stage_times(dev, params, rs, image_weights, flags, iters):
    """The three C-ABI calls of one GaussianRasterizer.apply_weights view (preprocessing without colours, binning, K12), HIP
    events between them -> ({stage: ms}, R)."""
    import ctypes

    from gaussianeditor_amd import _native

    L = _native.lib()
    s = torch.cuda.current_stream(dev)
    sp = s.cuda_stream
    P, H, W, C = int(params["xyz"].shape[0]), int(rs.image_height), int(rs.image_width), int(image_weights.shape[0])
    p = lambda t: t.data_ptr()  # noqa: E731
    op_flat = params["opacity"].contiguous()
    names = ("preprocess", "bin", "trace_weights")
    acc, R = {k: 0.0 for k in names}, 0
    for it in range(iters + 2):
        gb, _, ib = _native.scratch_sizes(P, 0, W, H)
        geom = torch.empty(gb, dtype=torch.uint8, device=dev)
        img = torch.empty(ib, dtype=torch.uint8, device=dev)
        radii_t = torch.empty(P, dtype=torch.int32, device=dev)
        w = torch.zeros((P, C), device=dev)
        cnt = torch.zeros((P,), dtype=torch.int32, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        Rc = (ctypes.c_int64 * 2)()
        ev[0].record(s)
        _native.check("pre", L.gsr_preprocess(sp, P, 0, 0, p(params["xyz"]), p(params["scaling"]), 1.0, p(params["rotation"]),
                                              p(op_flat), None, None, None, p(rs.viewmatrix), p(rs.projmatrix), None, W, H,
                                              float(rs.tanfovx), float(rs.tanfovy), 0, 1, flags | 8, p(radii_t), p(geom), Rc))
        ev[1].record(s)
        R, Gi = int(Rc[0]), int(Rc[1])
        _, bb, _ = _native.scratch_sizes(P, R, W, H, Gi)
        binning = torch.empty(bb, dtype=torch.uint8, device=dev)
        _native.check("bin", L.gsr_bin(sp, P, R, Gi, W, H, p(geom), p(binning), p(img)))
        ev[2].record(s)
        _native.check("trace", L.gsr_trace_weights(sp, P, R, W, H, C, p(geom), p(binning), p(img), p(image_weights), p(w), p(cnt), flags))
        ev[3].record(s)
        torch.cuda.synchronize(dev)
        if it >= 2:
            for i, k in enumerate(names):
                acc[k] += ev[i].elapsed_time(ev[i + 1])
    return {k: acc[k] / iters for k in names}, R


def load_counters(key):
    """profiles/traffic_latest.json: per-launch HBM bytes and VALU wave-instructions per stage, measured with rocprofv3 PMC
    passes on this workload (tools/collect_counters.py) and stamped with the hash of the kernel sources; stale (or absent)
    -> (None, reason)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    except Exception:
        return None, "profiles/traffic_latest.json is missing"
    if tj.get("csrc_sha16") != csrc_sha16():
        return None, (f"stale: profiles/traffic_latest.json was measured on kernel sources {tj.get('csrc_sha16')}, "
                      f"this build is {csrc_sha16()}")
    w = tj.get("workloads", {}).get(key)
    if w is None:
        return None, f"profiles/traffic_latest.json holds no counters for {key}"
    return w, tj.get("source")


def roofline_block(stage_ms, ab, cb, dominant, counters, source, dev, pix_inst, info=None):
    """The line's `roofline` object for the stage that takes the most time."""
    t = stage_ms[dominant] * 1e-3
    ach = ab[dominant] / t / 1e9
    # `achieved` / `frac`: the contract's figure -- SURVEY.md section 8(d)'s algorithmic bytes of the stage / its time, against
    # 8 TB/s (round 4 printed the compulsory-byte model under these names and the survey's under *_8d; the review asked for the
    # names to mean what the contract says).  Section 8(d) counts a 40-byte gather and a 36-byte gradient per (tile, Gaussian)
    # INSTANCE, which the L2 serves: on deep-tile scenes the figure exceeds 1.  `frac_compulsory` prices a record once per
    # visible Gaussian instead, `frac_counter` is what the memory controllers counted.
    out = {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "model": "achieved = SURVEY.md section 8(d) algorithmic bytes of the dominant stage / its HIP-event time (bench.py "
                    "algorithmic_bytes); frac = achieved / peak.  frac_compulsory: the same with the blend stages' per-instance "
                    "gather / gradient terms once per visible Gaussian (bench.py compulsory_bytes)",
           "frac": ach / HBM_PEAK_GBS,
           "achieved_8d": ach, "frac_8d": ach / HBM_PEAK_GBS,
           "achieved_compulsory": cb[dominant] / t / 1e9, "frac_compulsory": cb[dominant] / t / 1e9 / HBM_PEAK_GBS,
           "traffic": None, "frac_counter": None, "valu_issue_frac": None, "traffic_source": source}
    if counters is not None:
        tb = counters.get("per_launch_bytes", {}).get(dominant)
        vi = counters.get("valu_wave_insts", {}).get(dominant)
        if tb is not None:
            out["traffic"] = tb
            out["frac_counter"] = tb / t / 1e9 / HBM_PEAK_GBS
        if vi is not None:
            simds = torch.cuda.get_device_properties(dev).multi_processor_count * 4
            # time the SIMDs would need for these instructions at the measured issue rate of the blend kernels' own mix
            # (ISSUE_CYCLES_PER_VALU) / the stage's time.  (Rounds 3-5: 4 cycles per instruction -- kept as *_4cycle.)
            out["valu_issue_frac"] = vi * ISSUE_CYCLES_PER_VALU / (simds * t * MAX_CLOCK_HZ)
            out["valu_issue_frac_4cycle"] = vi / (simds * t * MAX_CLOCK_HZ / 4.0)
            out["valu_issue_model"] = (f"{ISSUE_CYCLES_PER_VALU:.2f} cycles per VALU wave-instruction and SIMD: K7's group body "
                                       "replayed at 4 waves per SIMD (tools/microbench/k7_group_replay.py)")
            out["valu_wave_insts"] = vi
    if dominant in ("blend_forward", "blend_backward"):
        out["limiter"] = ("instruction issue / per-item latency, not HBM: see valu_issue_frac and pixel_instances_per_s "
                          "(DESIGN.md section 3.1)")
        out["pixel_instances_per_s"] = pix_inst / t
    if info and dominant == "blend_backward":
        out.update({k: info[k] for k in ("critical_item_frac", "mean_workgroup_frac", "k7_items", "k7_workgroups") if k in info})
    return out


def extra_configs(dev, flags, budget_s=60.0):
    """The other BASELINE.json configurations with the substitutes SURVEY.md section 8(d) prescribes, bounded to about a
    minute (rank 0, N = 1 only): C2 = 6 M Gaussians forward at 1080p (stage times + roofline), C3 = the 512^2 edit loop
    (SH render + override_color render + backward per step; and with the fused second image), C5 = tracing ms / view.
    The same measurements as tools/bench_configs.py."""
    from types import SimpleNamespace

    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianeditor_amd.gaussian_renderer import camera2rasterizer, render
    from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene

    t_begin = time.perf_counter()
    out = {"note": "substitutes for assets that are not available offline (bicycle.ply, the bear scene, diffusion weights); "
                   "tools/bench_configs.py measures the same"}
    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)

    class PC:
        def __init__(self, sc, grad=False):
            self.t = {k: v.to(dev).requires_grad_(grad) for k, v in sc.items() if isinstance(v, torch.Tensor) and k != "bg"}
            self.active_sh_degree, self.max_sh_degree = 3, 3

        get_xyz = property(lambda s: s.t["xyz"])
        get_opacity = property(lambda s: s.t["opacity"])
        get_scaling = property(lambda s: s.t["scaling"])
        get_rotation = property(lambda s: s.t["rotation"])
        get_features = property(lambda s: s.t["features"])

    def timed(fn, steps, warmup):
        # (the cyclic collector off around the timed steps, as around the headline's: one generation-2 pass is 50-90 ms -- inside
        #  a 30-step window it turned a 0.86 ms step into 2.06 ms once, profiles/r06_z_bench.json)
        import gc

        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        was = gc.isenabled()
        gc.disable()
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / steps
        finally:
            if was:
                gc.enable()

    def timed_median(fn, steps, warmup, windows=3):
        # (the median of a few short windows: a single host hiccup -- an allocator round trip, a scheduler preemption -- is tens of
        #  milliseconds, i.e. 2 ms per step in a 30-step window; seen twice in round 6 on the first C3 entry, 2.4 and 2.7 ms
        #  where every other run gave 0.64-0.70)
        runs = sorted(timed(fn, steps, warmup if i == 0 else 2) for i in range(windows))
        return runs[len(runs) // 2]

    # ---- C3 / C5 on the headline scene at the editor's 512 x 512
    sc = synth_scene(1_000_000, seed=0, s0=0.01)
    bg = sc["bg"].to(dev)
    pc = PC(sc, grad=True)
    cam = ring_cameras(8, 512, 512)[0].to(dev)
    G = seed_gradient(512, 512, 0).to(dev)
    mask = (torch.rand(1_000_000, 1, device=dev) > 0.5).float().repeat(1, 3)

    def edit_step():
        a = render(cam, pc, pipe, bg)
        with torch.no_grad():
            render(cam, pc, pipe, bg, override_color=mask)  # the semantic pass (GassuianEditor.py:183-191)
        (a["render"] * G).sum().backward()
        for v in pc.t.values():
            v.grad = None

    def edit_step_fused():  # both images from ONE preprocessing / sort: render(..., semantic_color=mask)
        a = render(cam, pc, pipe, bg, semantic_color=mask)
        (a["render"] * G).sum().backward()
        for v in pc.t.values():
            v.grad = None

    import gaussianeditor_amd as _pkg
    from gaussianeditor_amd.diff_gaussian_rasterization import _reuse

    # the reference's UNMODIFIED call pattern (two render() calls per view): since round 6 the second one is served by the
    # blend kernel alone when the rasterizer can prove it is the view it rendered last (view reuse, _reuse.py)
    hits0 = _reuse.stats["hits"]
    t = timed_median(edit_step, 30, 10)
    out["C3_edit_loop_512_1M"] = {"ms_per_step": 1e3 * t, "what": "SH render + override_color render + backward, 1 M Gaussians, "
                                  "two unmodified render() calls per step (view reuse on: the default)",
                                  "view_reuse_hits": _reuse.stats["hits"] - hits0}
    try:
        _pkg.set_view_reuse(False)
        t = timed_median(edit_step, 30, 5)
    finally:
        _pkg.set_view_reuse(True)
    out["C3_edit_loop_512_1M_no_reuse"] = {"ms_per_step": 1e3 * t, "what": "the same with GSR_VIEW_REUSE=0: two full renders (rounds 1-5)"}
    t = timed_median(edit_step_fused, 30, 5)
    out["C3_edit_loop_512_1M_fused_semantic"] = {"ms_per_step": 1e3 * t}
    # stage times + roofline of ONE 512 x 512 view of this loop (the editor's resolution: the K1 -> K6 chain and K8+K9 are most
    # of the step here, the blend kernels run as SPLIT items / list segments)
    rs512 = GaussianRasterizationSettings(512, 512, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0, cam.world_view_transform,
                                          cam.full_proj_transform, 3, cam.camera_center, False, False)
    p512 = {k: pc.t[k].detach() for k in ("xyz", "opacity", "features", "scaling", "rotation")}
    info512 = {}
    st5, R5, V5, pix5 = stage_times(dev, p512, rs512, G, 3, flags, 10, info=info512)
    N5, T5 = 512 * 512, 32 * 32
    ab5, cb5 = algorithmic_bytes(1_000_000, V5, R5, N5, T5, 16), compulsory_bytes(1_000_000, V5, R5, N5, T5, 16)
    dom5 = max(st5, key=st5.get)
    out["C3_edit_loop_512_1M"].update({"stage_ms": st5, "num_rendered": R5, "visible": V5,
                                       "roofline": roofline_block(st5, ab5, cb5, dom5, None, "no PMC pass for this workload", dev,
                                                                  pix5, info512)})

    # ... and with the reference's GaussianModel in front of it: parameters behind activations, so opacity / scaling / rotation
    # are FRESH tensors on every render() (scene/gaussian_model.py:222-258) and reuse has to compare them on the device
    class PCAct:
        def __init__(self, sc):
            self.p = {"xyz": sc["xyz"].to(dev).requires_grad_(True),
                      "opacity": torch.logit(sc["opacity"].clamp(1e-4, 1 - 1e-4)).to(dev).requires_grad_(True),
                      "scaling": torch.log(sc["scaling"]).to(dev).requires_grad_(True),
                      "rotation": sc["rotation"].to(dev).requires_grad_(True),
                      "features": sc["features"].to(dev).requires_grad_(True)}
            self.active_sh_degree, self.max_sh_degree = 3, 3

        get_xyz = property(lambda s: s.p["xyz"])
        get_opacity = property(lambda s: torch.sigmoid(s.p["opacity"]))
        get_scaling = property(lambda s: torch.exp(s.p["scaling"]))
        get_rotation = property(lambda s: torch.nn.functional.normalize(s.p["rotation"]))
        get_features = property(lambda s: s.p["features"])

    pca = PCAct(sc)

    def edit_step_model():
        a = render(cam, pca, pipe, bg)
        semantic = render(cam, pca, pipe, bg, override_color=mask)["render"]  # (with autograd, as GassuianEditor.py:183-191)
        (a["render"] * G).sum().backward()
        for v in pca.p.values():
            v.grad = None
        return semantic

    res = {}
    for name, on in (("reuse", True), ("no_reuse", False)):
        try:
            _pkg.set_view_reuse(on)
            c0 = _reuse.stats["compares"]
            res[name] = {"ms_per_step": 1e3 * timed_median(edit_step_model, 30, 5), "compare_launches": _reuse.stats["compares"] - c0}
        finally:
            _pkg.set_view_reuse(True)
    out["C3_edit_loop_512_1M_reference_model"] = dict(res, what="the same loop over a GaussianModel-like object (activations "
                                                      "recomputed per render, autograd on in both renders)")
    del pca
    cams = [c.to(dev) for c in ring_cameras(12, 512, 512)]
    masks = [(torch.rand(1, 512, 512, device=dev) > 0.5).float() for _ in cams]
    zero_bg = torch.zeros(3, device=dev)

    def trace_all():
        w = torch.zeros(1_000_000, 1, device=dev)
        cnt = torch.zeros(1_000_000, 1, dtype=torch.int32, device=dev)
        with torch.no_grad():
            for c, m in zip(cams, masks):
                camera2rasterizer(c, zero_bg).apply_weights(pc.get_xyz, None, pc.get_opacity, None, w, pc.get_scaling,
                                                            pc.get_rotation, None, cnt, m)

    t = timed(trace_all, 5, 1)
    out["C5_apply_weights_12views_512_1M"] = {"ms_per_view": 1e3 * t / len(cams), "ms_total": 1e3 * t}
    rs_tr = GaussianRasterizationSettings(512, 512, math.tan(cams[0].FoVx / 2), math.tan(cams[0].FoVy / 2), zero_bg, 1.0,
                                          cams[0].world_view_transform, cams[0].full_proj_transform, 0, cams[0].camera_center,
                                          False, False)
    st_tr, R_tr = trace_stage_times(dev, p512, rs_tr, masks[0], flags, 10)
    # (K12 has no byte model in SURVEY.md section 8(d) -- "VALU / atomics" --: the fraction below is its compulsory traffic, the
    #  4-byte list entries + one 32-byte record pair per visible Gaussian + 4 B per pixel, against the stage's time)
    out["C5_apply_weights_12views_512_1M"].update({
        "stage_ms": st_tr, "num_rendered": R_tr,
        "roofline": {"bound": "hbm", "kernel": "trace_weights", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "achieved": (4 * R_tr + 32 * V5 + 4 * 512 * 512) / (st_tr["trace_weights"] * 1e-3) / 1e9,
                     "frac": (4 * R_tr + 32 * V5 + 4 * 512 * 512) / (st_tr["trace_weights"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "limiter": "instruction issue and memory-side atomics (one per (quadrant, Gaussian) and "
                                                  "channel), not HBM"}})
    del pc, mask
    torch.cuda.empty_cache()
    # ---- C2: 6 M Gaussians, forward only, 1080p
    if time.perf_counter() - t_begin < budget_s:
        P6, W, H = 6_000_000, 1920, 1080
        sc = synth_scene(P6, seed=0, s0=0.01)
        cam = ring_cameras(8, W, H)[0]
        params = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), sc["bg"].to(dev), 1.0,
                                           cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                           cam.camera_center.to(dev), False, False)
        st, R, V, pix = stage_times(dev, params, rs, None, 3, flags, 10, backward=False)
        N, T, M = W * H, ((W + 15) // 16) * ((H + 15) // 16), 16
        ab, cb = algorithmic_bytes(P6, V, R, N, T, M), compulsory_bytes(P6, V, R, N, T, M)
        dom = max(st, key=st.get)
        counters, src = load_counters(workload_key(P6, W, H, 0.01))
        fwd_ms = sum(st.values())
        out["C2_synth6M_1080p_forward"] = {
            "forward_ms": fwd_ms, "renders_per_s": 1e3 / fwd_ms, "mpixels_per_s": N / fwd_ms / 1e3, "stage_ms": st,
            "num_rendered": R, "visible": V,
            "what": "sum of the three forward stages through the C ABI (HIP events), forward-only preprocess",
            "hbm_fraction_forward": sum(cb[k] for k in st) / (fwd_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
            "roofline": roofline_block(st, ab, cb, dom, counters, src, dev, pix)}
        del params, sc
        torch.cuda.empty_cache()
    # ---- synth-v2 at the headline's size: a scene that looks like a trained capture to the rasterizer --
    # every tile non-empty, ~70 % of the visible Gaussians receive a gradient -- train iteration + forward, stage times, roofline
    if time.perf_counter() - t_begin < budget_s + 30.0:
        from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer
        from gaussianeditor_amd.multiview import GradBucket, multiview_batch_step, multiview_step
        from gaussianeditor_amd.synth import synth_scene_v2

        P2, W, H = 1_000_000, 1920, 1080
        sc = synth_scene_v2(P2, seed=0)
        ring = ring_cameras(8, W, H)
        params = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}

        def rs_of(v):
            c = ring[v % 8]
            return GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), sc["bg"].to(dev), 1.0,
                                                 c.world_view_transform.to(dev), c.full_proj_transform.to(dev), 3,
                                                 c.camera_center.to(dev), False, False)

        rs = rs_of(0)
        G2 = seed_gradient(H, W, 0).to(dev)
        bucket = GradBucket(P2, 16, dev, sh_exchange="auto")
        t_train = timed(lambda: multiview_step(rs, params, G2, bucket, rows="auto"), 100, 30)
        rast, m2d = GaussianRasterizer(rs), torch.zeros_like(params["xyz"])

        def fwd():
            with torch.no_grad():
                rast(params["xyz"], m2d, params["opacity"], shs=params["features"], scales=params["scaling"], rotations=params["rotation"])

        t_fwd = timed(fwd, 100, 10)
        v2_info = {}
        st, R, V, pix = stage_times(dev, params, rs, G2, 3, flags, 10, info=v2_info)
        N, T, M = W * H, ((W + 15) // 16) * ((H + 15) // 16), 16
        ab, cb = algorithmic_bytes(P2, V, R, N, T, M), compulsory_bytes(P2, V, R, N, T, M)
        dom = max(st, key=st.get)
        counters, src = load_counters(workload_key(P2, W, H, 0.0, "v2"))
        out["synth_v2_1M_1080p"] = {
            "what": "synth-v2 (gaussianeditor_amd/synth.py: disks on a dome, a ground, three shells and a wall; bimodal opacity), "
                    "1 M Gaussians, 1920x1080, ring view 0: one train iteration (forward + backward through the L1 API) and one "
                    "forward, wall clock over 100 iterations each; stage times from HIP events around the C-ABI calls",
            "train_ms_per_step": 1e3 * t_train, "train_iters_per_s": 1.0 / t_train, "forward_ms": 1e3 * t_fwd,
            "forward_mpixels_per_s": N / t_fwd / 1e6, "stage_ms": st, "num_rendered": R, "visible": V,
            "roofline": roofline_block(st, ab, cb, dom, counters, src, dev, pix, v2_info)}
        del bucket
        # ---- the headline iteration through the drop-in route itself: render() (L2) -> GaussianRasterizer (L1 autograd.Function)
        # -> loss.backward(), i.e. gaussian_renderer/__init__.py:45-150 as an unmodified caller uses it.  The headline `value`
        # calls the L0 entry points directly (same native calls): this entry says what L1 / L2 + the autograd engine add.
        sch = synth_scene(1_000_000, seed=0, s0=0.01)
        pch = PC(sch, grad=True)
        camh = ring[0].to(dev)
        bgh = sch["bg"].to(dev)
        Gh = seed_gradient(H, W, 0).to(dev)

        def l2_step():
            a = render(camh, pch, pipe, bgh)
            (a["render"] * Gh).sum().backward()
            for v in pch.t.values():
                v.grad = None

        # (three windows of 50 steps, the median reported: one 100-step window once came out at 1.12 ms on a box whose other
        #  three runs gave 0.60 -- a single host hiccup is 0.5 ms per step in a window this short; profiles/r06_z_bench.json)
        l2_runs = sorted(timed(l2_step, 50, 30 if i == 0 else 5) for i in range(3))
        t_l2 = l2_runs[1]
        out["headline_via_render_l2"] = {"ms_per_step": 1e3 * t_l2, "iters_per_s": 1.0 / t_l2,
                                         "ms_per_step_min": 1e3 * l2_runs[0], "ms_per_step_max": 1e3 * l2_runs[2],
                                         "what": "synth-v1 1 M Gaussians, 1920x1080, ring view 0: render() + (image * G).sum()."
                                                 "backward() per step (two extra elementwise kernels over the image for the "
                                                 "loss), wall clock, median of three windows of 50 steps"}
        del pch
        # ---- the fixed 8-view batch of configs[3] on ONE GPU (multiview_batch_step: two-stream view pipelining, the blend kernels
        # at 2 waves per SIMD by the library's own choice), headline scene; 3 repetitions each way -> median and range
        sc1 = synth_scene(1_000_000, seed=0, s0=0.01)
        p1 = {k: sc1[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        rs8 = [GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), sc1["bg"].to(dev), 1.0,
                                             c.world_view_transform.to(dev), c.full_proj_transform.to(dev), 3,
                                             c.camera_center.to(dev), False, False) for c in ring]
        import gaussianeditor_amd.multiview as mv

        res = {}
        pipe_default = mv._VIEW_PIPELINE
        try:
            for name, pipe_on in (("pipelined", True), ("serial", False)):
                mv._VIEW_PIPELINE = pipe_on
                b8 = GradBucket(1_000_000, 16, dev, sh_exchange="rgb")
                runs = []
                for _ in range(3):
                    t8 = timed(lambda: multiview_batch_step(rs8, p1, [G2] * 8, b8), 12, 4)
                    runs.append(8.0 / t8)
                runs.sort()
                res[name] = {"view_iters_per_s_median": runs[1], "min": runs[0], "max": runs[2], "ms_per_view": 1e3 / runs[1]}
                del b8
        finally:
            mv._VIEW_PIPELINE = pipe_default
        res["pipelined_over_serial"] = res["pipelined"]["view_iters_per_s_median"] / res["serial"]["view_iters_per_s_median"]
        out["views8_one_gpu"] = dict(res, what="the 8 ring views of BASELINE configs[3] as ONE batch on one GPU (bench.py --views 8): "
                                               "forward + backward of every view, touched-rows messages, one accumulate; three "
                                               "repetitions of 12 steps each way")
        # ---- the headline iteration with persistent gradient rows (opt-in, GradBucket(persistent_rows=True)): the backward
        # rewrites a zero gradient row only if it does not hold zeros already.  NOT the headline: that writes every row.
        bp = GradBucket(1_000_000, 16, dev, sh_exchange="auto", persistent_rows=True)
        t_p = timed(lambda: multiview_step(rs8[0], p1, G2, bp, rows="auto"), 100, 30)
        out["persistent_rows_1M_1080p"] = {
            "what": "the headline workload (synth-v1, 1 M Gaussians, 1920x1080, ring view 0) with GradBucket(persistent_rows=True): "
                    "same gradients, zero rows that already hold zeros are not rewritten; wall clock over 100 iterations",
            "train_ms_per_step": 1e3 * t_p, "train_iters_per_s": 1.0 / t_p}
        del bp, params, p1
        torch.cuda.empty_cache()
    out["seconds"] = time.perf_counter() - t_begin
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--keep-gc", action="store_true",
                    help="leave Python's cyclic garbage collector on during the timed steps (default: collected once and switched "
                         "off around them, as training loops do: a generation-2 pass stalls the launch thread for milliseconds)")
    ap.add_argument("--prewarm", type=int, default=300,
                    help="untimed train steps in FRONT of the --warmup steps (same count on every rank): a GPU that comes out of idle "
                         "needs tens of milliseconds of load to reach its clocks, the contract's warmup may be a handful of steps")
    ap.add_argument("--views", type=int, default=0,
                    help="views per step of the WHOLE job (default 0 = one per GPU, weak scaling).  A multiple of --gpus: every "
                         "rank renders views/gpus views per step and the batch is fixed as N grows (strong scaling of "
                         "BASELINE configs[3]'s 8-view batch: --views 8)")
    ap.add_argument("--no-view-pipeline", action="store_true",
                    help="with --views: render a rank's views one after the other on one stream (A/B of the two-stream pipelining)")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--s0", type=float, default=0.01, help="synth-v1 median scale (0.01 = headline; 0.03-0.05 = deep tiles)")
    ap.add_argument("--scene", choices=("v1", "v2"), default="v1",
                    help="v1 = synth-v1, the uniform cube of SURVEY.md section 8(d) (the headline); v2 = synth-v2, surfaces of "
                         "disk-like Gaussians with bimodal opacity seen from inside (every tile non-empty, most visible "
                         "Gaussians receive a gradient: gaussianeditor_amd/synth.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the `extra_configs` block (6 M forward, edit loop, tracing)")
    ap.add_argument("--train-only", action="store_true",
                    help="profiling runs (tools/gpu_counters.sh): only the warm-up and the timed train iterations, nothing else "
                         "is launched -- every kernel launch of the process belongs to a train iteration")
    ap.add_argument("--cpu-iters", type=int, default=15, help="oracle train iterations timed for cpu_baseline")
    ap.add_argument("--ply", type=str, default=None,
                    help="a 3DGS point cloud in the reference's save_ply layout (gaussiansplatting/scene/gaussian_model.py:"
                         "410-445), e.g. Mip-NeRF360 bicycle/point_cloud/iteration_30000/point_cloud.ply = BASELINE configs[1]; "
                         "replaces the synthetic scene (cameras stay ring-v1 around the scene's median, see --ply-fit)")
    ap.add_argument("--ply-fit", type=int, default=1,
                    help="1 (default): translate the loaded scene's median to the origin and scale it (positions and scales) so "
                         "that 90 %% of the Gaussians lie within the unit ball the ring-v1 cameras look at; 0: as stored")
    ap.add_argument("--persistent-grads", action="store_true",
                    help="GradBucket(persistent_rows=True): the backward rewrites a zero gradient row only if it does not hold zeros "
                         "already (development runs; the default writes every row of every gradient every iteration)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="development: on ONE GPU, run the multi-rank gradient exchange anyway (an RCCL group of one rank, "
                         "touched-rows route, every collective issued): what a rank's step costs locally before a byte "
                         "crosses xGMI.  The line is marked; it is not the benchmark.")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # The contract: rank 0 prints ONE JSON line.  RCCL writes a version banner to the process's stdout through C stdio (it
    # appears when the process exits, i.e. BEHIND anything Python printed), so the descriptor the line goes to is set aside
    # and everything else any library writes to stdout -- on every rank -- is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    views = args.views if args.views > 0 else world
    if views % world != 0:
        raise SystemExit(f"--views {views} must be a multiple of --gpus {world}")
    batch_mode = args.views > 0  # the fixed batch of --views views, dealt to the ranks (multiview_batch_step)
    if batch_mode and (args.force_exchange or args.persistent_grads):
        raise SystemExit("--views does not combine with --force-exchange / --persistent-grads")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the rasterizer has no CPU fallback)")
    # GSR_BENCH_SHARED_GPU=1 (testing only): run all ranks on GPU 0 with the gloo backend, to exercise the multi-rank
    # code path on a box with a single GPU.  The numbers of such a run mean nothing.
    shared = os.environ.get("GSR_BENCH_SHARED_GPU", "0") == "1"
    if shared:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    elif args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)

    # Several views per rank: multiview_batch_step pipelines successive views on two streams (view v + 1's forward under
    # view v's backward) and tells the rasterizer itself to run the persistent blend kernels at 2 waves per SIMD for those
    # views (GSR_FLAG_SHARED_SIMDS; until round 4 this file set GSR_BLEND_WAVES_PER_SIMD=2 for the process).
    if args.no_view_pipeline:
        os.environ["GSR_VIEW_PIPELINE"] = "0"
    import gaussianeditor_amd
    from gaussianeditor_amd import _native

    # GSR_TILE_BOUNDS=alpha (opt-in): bin by the alpha >= 1/255 box; default = the reference's rule
    gaussianeditor_amd.set_tile_bounds(os.environ.get("GSR_TILE_BOUNDS", "reference"))
    # GSR_FAST_EXP=1 (opt-in): hardware 2^x in the blend loops instead of the specified polynomial
    gaussianeditor_amd.set_fast_exp(os.environ.get("GSR_FAST_EXP", "0") == "1")
    from gaussianeditor_amd import options
    flags = options.current_flags()
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussianeditor_amd.multiview import GradBucket, multiview_batch_step, multiview_step, views_of_rank
    from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene, synth_scene_v2

    P, W, H = args.gaussians, args.width, args.height
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    if args.ply is not None:
        from gaussianeditor_amd import scene_ply

        raw = scene_ply.load_gaussians_ply(args.ply)
        sc = scene_ply.activated(raw)
        if args.ply_fit:
            c = sc["xyz"].median(dim=0).values
            r = torch.quantile((sc["xyz"] - c).norm(dim=1)[:8_000_000], 0.9).clamp_min(1e-12)
            sc["xyz"] = ((sc["xyz"] - c) / r).contiguous()
            sc["scaling"] = (sc["scaling"] / r).contiguous()
        sc["bg"] = torch.zeros(3)
        P = int(sc["xyz"].shape[0])
        ply_degree = int(raw["max_sh_degree"])
    else:
        sc = synth_scene_v2(P, seed=0) if args.scene == "v2" else synth_scene(P, seed=0, s0=args.s0, sh_degree=3)
        ply_degree = 3
    M = sc["features"].shape[1]
    ring = ring_cameras(8, W, H)
    params = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
    G = seed_gradient(H, W, 0).to(dev)

    def settings_of(view):
        cam_ = ring[view % 8]
        return GaussianRasterizationSettings(H, W, math.tan(cam_.FoVx / 2), math.tan(cam_.FoVy / 2), sc["bg"].to(dev), 1.0,
                                             cam_.world_view_transform.to(dev), cam_.full_proj_transform.to(dev), ply_degree,
                                             cam_.camera_center.to(dev), False, False)

    my_views = list(views_of_rank(views, world, rank))
    cam = ring[my_views[0] % 8]
    tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    rs = settings_of(my_views[0])
    rs_list = [settings_of(v) for v in my_views]
    bucket = GradBucket(P, M, dev, sh_exchange="rgb" if (args.force_exchange or batch_mode) else "auto",
                        persistent_rows=args.persistent_grads)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds: float) -> float:
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---------------- train step: fwd + bwd + exchange ----------------
    route = {"last": "local"}
    # GSR_BENCH_ROWS=0|1 (testing): force the dense / the touched-rows form of the exchange instead of choosing by bytes
    rows_env = os.environ.get("GSR_BENCH_ROWS")
    rows_mode = "auto" if rows_env is None else (rows_env == "1")
    marks_log = []

    def train_step(record=False, marks=None):
        if record and marks is None:
            marks = {}
        if batch_mode:
            # the fixed batch: this rank's block of views, one message per view, one all-gather (multiview_batch_step)
            _, radii, _, _ = multiview_batch_step(rs_list, params, [G] * len(rs_list), bucket, marks=marks)
        else:
            # forward, radii MAX all-reduce started, backward, gradient exchange (gaussianeditor_amd/multiview.py)
            _, radii, _, _ = multiview_step(rs, params, G, bucket, rows=True if args.force_exchange else rows_mode,
                                            force_exchange=args.force_exchange, marks=marks)
        route["last"] = bucket.last_route
        if record:
            marks_log.append(marks)
        return radii

    exchange = bucket.sh_exchange
    # (a failing exchange fails the benchmark: it must never silently measure another workload)
    for _ in range(max(args.prewarm, 0)):  # (untimed, in front of the contract's warmup: clock ramp)
        train_step()
    sync_all()
    for _ in range(max(args.warmup, 1) if world > 1 else args.warmup):
        train_step()
    sync_all()
    if dist.is_initialized():
        # RCCL's banner sits in the C library's stdout buffer until the process exits; push it out NOW (to stderr, see above),
        # so that it also precedes the JSON line for a reader that merges the two streams
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # (no libc handle: the banner then leaves at exit, still on stderr)
            pass
    # Timed region: exactly `steps` steps between barrier + synchronize on both sides (the contract's number).  An event
    # per step on the launch stream additionally gives the distribution of the GPU-side step time (median / p10 / p90,
    # SURVEY.md section 8(d)) without adding any synchronisation.
    cur = torch.cuda.current_stream(dev)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # (the steps' own marks -- local_done / step_done, multiview._mark -- are created here as well: an event created inside the
    #  timed region can stall the launch thread for milliseconds when the runtime has to grow its pool)
    step_marks = [{k: torch.cuda.Event(enable_timing=True) for k in ("local_done", "step_done")} for _ in range(args.steps)]
    import gc

    if not args.keep_gc:
        gc.disable()  # (no collect() here: tens of milliseconds of host time in front of the timed region let the GPU fall idle)
    t0 = time.perf_counter()
    step_ev[0].record(cur)
    for i in range(args.steps):
        train_step(record=True, marks=step_marks[i])
        step_ev[i + 1].record(cur)
    sync_all()
    train_s = max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    local_ms = [step_ev[i].elapsed_time(m["local_done"]) for i, m in enumerate(marks_log) if m and "local_done" in m]
    exch_ms = [m["local_done"].elapsed_time(m["step_done"]) for m in marks_log if m and "step_done" in m]

    # ---------------- multi-GPU diagnostics: every rank's view of the step ----------------
    multi_gpu = None
    if world > 1 or args.force_exchange:
        lx = bucket.last_exchange if route["last"] == "rows" else None
        mine = torch.tensor([percentile(step_ms, 0.5) or 0.0, percentile(local_ms, 0.5) or 0.0, percentile(exch_ms, 0.5) or 0.0,
                             float((lx or {}).get("bytes_sent", 0)), float((lx or {}).get("bytes_received", 0))],
                            dtype=torch.float64, device=dev)
        if world > 1:
            allr = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        per_rank = [[float(x) for x in r.tolist()] for r in allr]
        # do the replicas hold the same bits?  (the exchanged gradients of the last step; one MAX + one MIN all-reduce of
        # an exact fingerprint, multiview.replicas_identical) -- a mismatch makes the run FAIL behind its line
        from gaussianeditor_amd.multiview import replicas_identical

        exchanged = [v for k, v in sorted(bucket.views.items()) if v is not None]
        identical = bool(replicas_identical(exchanged)) if world > 1 else True
        print(f"[bench rank {rank}] world_size seen by the {dist.get_backend() if dist.is_initialized() else 'no'} backend: "
              f"{dist.get_world_size() if dist.is_initialized() else 1}, device {dev}, route {route['last']}, "
              f"views {list(my_views)}, step {percentile(step_ms, 0.5) or 0.0:.3f} ms (local {percentile(local_ms, 0.5) or 0.0:.3f} + "
              f"exchange {percentile(exch_ms, 0.5) or 0.0:.3f}), sent {int((lx or {}).get('bytes_sent', 0))} B, received "
              f"{int((lx or {}).get('bytes_received', 0))} B, replicas identical: {identical}", file=sys.stderr, flush=True)
        try:
            nccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as ex:  # (a build without the binding)
            nccl_version = f"unavailable ({type(ex).__name__})"
        multi_gpu = {
            "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": dist.get_backend() if dist.is_initialized() else None,
            "rccl_version": nccl_version,
            "shared_gpu_test_run": shared,
            "views_per_step": views, "views_per_rank": len(my_views),
            "route": route["last"],
            "replicas_identical": identical,
            "per_rank": {"step_ms_gpu": [r[0] for r in per_rank],
                         "local_ms_gpu": [r[1] for r in per_rank],
                         "exchange_ms_gpu": [r[2] for r in per_rank],
                         "bytes_sent_per_step": [int(r[3]) for r in per_rank],
                         "bytes_received_per_step": [int(r[4]) for r in per_rank]},
            "note": "medians over the timed steps of HIP events on each rank's launch stream: step = local (forward + backward "
                    "of the rank's views, incl. packing) + exchange (collectives + the accumulate kernel); bytes are the "
                    "touched-rows messages of the last step (0 on the dense route: its all-reduce moves the whole bucket)",
        }

    fwd_s, fwd_ms = None, []
    stage_ms, ab, cb, R, V, pix_inst, k7_info = {}, {}, {}, 0, 0, 0, {}
    if not args.train_only:
        # ---------------- forward only ----------------
        rast = GaussianRasterizer(rs)
        m2d = torch.zeros_like(params["xyz"])

        def fwd_step():
            with torch.no_grad():
                return rast(params["xyz"], m2d, params["opacity"], shs=params["features"], scales=params["scaling"],
                            rotations=params["rotation"])

        for _ in range(max(2, args.warmup // 2)):
            fwd_step()
        sync_all()
        fwd_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        fwd_ev[0].record(cur)
        for i in range(args.steps):
            fwd_step()
            fwd_ev[i + 1].record(cur)
        sync_all()
        fwd_s = max_over_ranks(time.perf_counter() - t0)
        fwd_ms = [fwd_ev[i].elapsed_time(fwd_ev[i + 1]) for i in range(args.steps)]

        # ---------------- per-stage GPU time (HIP events on the launch stream), rank-local ----------------
        k7_info = {}
        stage_ms, R, V, pix_inst = stage_times(dev, params, rs, G, ply_degree, flags, max(5, min(args.steps, 20)), info=k7_info)
        ab, cb = algorithmic_bytes(P, V, R, N, T, M), compulsory_bytes(P, V, R, N, T, M)

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N == 1) ----------------
    # Both legs are BOUNDED: the C++/OpenMP oracle runs one iteration, then as many more as fit ~12 s (at most
    # --cpu-iters); the PyTorch-CPU restatement runs in a subprocess (its own thread pool -- sharing the process with the
    # oracle's OpenMP runtime on a 256-thread host oversubscribes it into a crawl) with a hard timeout, blends every 8th
    # non-empty tile and scales the blend time up (anchor, measured once in full: profiles/r04_a_torch_cpu_full.md).
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.train_only:
        from oracle import cpu as O

        cam0 = cam
        G_cpu = seed_gradient(H, W, 0)

        def oracle_iter():
            f_ = O.forward(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], None, None,
                           cam0.world_view_transform, cam0.full_proj_transform, cam0.camera_center, sc["bg"], W, H, tfx, tfy,
                           1.0, ply_degree)
            O.backward(f_, G_cpu, sc["xyz"], sc["scaling"], sc["rotation"], sc["features"], None, None,
                       cam0.world_view_transform, cam0.full_proj_transform, cam0.camera_center, sc["bg"], W, H, tfx, tfy,
                       1.0, ply_degree)
            return f_

        t0 = time.perf_counter()
        f = oracle_iter()
        first_s = time.perf_counter() - t0
        more = max(0, min(args.cpu_iters - 1, int(12.0 / max(first_s, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(more):
            oracle_iter()
        cpu_s = (time.perf_counter() - t0) if more else first_s
        n_it = more if more else 1
        cpu = {"value": n_it / cpu_s, "unit": "train iters/s", "cores": O.num_threads(), "kind": "port",
               "sample": f"{n_it} train iterations (fwd+bwd, one {W}x{H} view of the same {P}-Gaussian scene) "
                         f"in {cpu_s:.1f} s with OpenMP over {O.num_threads()} threads (after one warm-up iteration of "
                         f"{first_s:.1f} s); host has {os.cpu_count()} logical cores",
               "pixel_instances_per_view": int(f["pixel_instances"])}
        # second leg: the PyTorch-CPU restatement of the forward render, same view
        import subprocess

        # (32 threads at most: the restatement is torch's stable argsort, gathers and per-tile cumprod, whose intra-op
        #  parallelism saturates there -- measured in full on the 256-thread host: 8.0 s per render with 32 threads, 55.4 s
        #  with 128, profiles/r04_a_torch_cpu_full.md; the oracle leg above uses every core)
        threads = max(1, min(32, os.cpu_count() or 1))
        stride = 8
        try:
            if args.ply is not None:
                raise RuntimeError("the PyTorch-CPU leg regenerates the synth-v1 scene; not run for --ply")
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
            pr = subprocess.run([sys.executable, "-m", "oracle.torch_cpu", str(P), str(W), str(H), str(args.s0), str(rank % 8), "8",
                                 str(stride), str(threads)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
            tj = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
            cpu["torch_cpu"] = {"value": 1.0 / tj["seconds_per_render"], "unit": "forward renders/s", "cores": tj["threads"],
                                "kind": "port",
                                "sample": f"oracle/torch_cpu.py (vectorised float32 torch ops, stable argsort, per-tile cumprod "
                                          f"blending) on the same view: preprocess + sort of all {tj['num_rendered']} instances "
                                          f"{tj['prepare_s']:.2f} s, blending of {tj['tiles_blended']} of {tj['tiles_nonempty']} "
                                          f"non-empty tiles (every {stride}th) {tj['blend_s_sampled']:.2f} s, scaled to all tiles; "
                                          f"{threads} threads",
                                "anchor": "measured in full once (all 1 696 non-empty tiles of this view): 8.02 s per render = "
                                          "0.125 renders/s with 32 threads, 55.4 s with 128 threads on the 256-thread EPYC 9575F "
                                          "host (profiles/r04_a_torch_cpu_full.md)"}
        except Exception as ex:  # the second leg must never cost the bench line
            cpu["torch_cpu"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}

    rows_per_view = bucket.last_counts if route["last"] == "rows" else None
    extra = None
    # (only next to the headline workload: a bench line of another workload -- tests, --ply, deep tiles -- stays short)
    is_headline = (args.ply is None and args.scene == "v1" and (P, W, H) == (1_000_000, 1920, 1080) and args.s0 == 0.01
                   and not batch_mode)
    if rank == 0 and world == 1 and is_headline and not (args.no_extra_configs or args.train_only or args.force_exchange):
        del bucket
        torch.cuda.empty_cache()
        extra = extra_configs(dev, flags)

    if rank == 0:
        iters_per_s = views * args.steps / train_s
        out = {
            "metric": "train iters/sec (fwd+bwd of one 1080p view per GPU + grad all-reduce), 1M Gaussians @1080p",
            "value": iters_per_s,
            "unit": "view-iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * train_s / args.steps,
            "step_ms_gpu": {"median": percentile(step_ms, 0.5), "p10": percentile(step_ms, 0.1), "p90": percentile(step_ms, 0.9),
                            "first": step_ms[0] if step_ms else None, "max": max(step_ms) if step_ms else None,
                            "note": "HIP events around every timed step on the launch stream of rank 0"},
            "prewarm_steps": max(args.prewarm, 0),
            "higher_is_better": True,
            "scaling": "strong" if batch_mode else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if args.ply is None else "file (scene) + synthetic cameras / pixel gradient",
            "config": {"workload": ((f"synth-v1 {P} Gaussians SH3 (M=16, s0={args.s0})" if args.scene == "v1" else
                                     f"synth-v2 {P} Gaussians SH3 (M=16): disks on surfaces, bimodal opacity, every tile non-empty")
                                    + f", {W}x{H}, ring-v1 8 views, "
                                    + (f"a fixed batch of {views} views per step dealt to the ranks" if batch_mode else "one view per GPU")
                                    + " (BASELINE.json configs[3] shape; configs[1] bicycle.ply is not available offline)")
                       if args.ply is None else
                       (f"{os.path.basename(args.ply)}: {P} Gaussians SH{ply_degree} (M={M}) loaded from the reference's save_ply "
                        f"layout{' (fitted to the unit ball)' if args.ply_fit else ''}, {W}x{H}, ring-v1 8 views "
                        "(BASELINE.json configs[1] shape when the file is bicycle/point_cloud.ply)"),
                       "gaussians": P, "width": W, "height": H, "views_per_step": views, "views_per_rank": len(my_views),
                       "parallelism": f"dp{world}-views",
                       "view_pipeline": bool(batch_mode and len(my_views) > 1 and not args.no_view_pipeline),
                       "blend_waves_per_simd": (int(os.environ["GSR_BLEND_WAVES_PER_SIMD"]) if "GSR_BLEND_WAVES_PER_SIMD" in os.environ
                                                else (2 if (batch_mode and len(my_views) > 1 and not args.no_view_pipeline) else 4)),
                       "tile_bounds": gaussianeditor_amd.get_tile_bounds(), "fast_exp": gaussianeditor_amd.get_fast_exp(),
                       "synth_s0": args.s0,
                       "grad_exchange": exchange + (" (forced on one rank: development run)" if args.force_exchange else ""),
                       "grad_exchange_route": route["last"],
                       "grad_rows": "persistent: zero rows rewritten only when they do not hold zeros already (--persistent-grads)"
                       if args.persistent_grads else "every row of every gradient written every iteration",
                       "grad_exchange_rows_per_view": rows_per_view,
                       "num_rendered": R, "visible": V, "pixel_instances": pix_inst,
                       "sort_key_bits": int(_native.lib().gsr_sort_key_bits(W, H))},
        }
        if multi_gpu is not None:
            out["multi_gpu"] = multi_gpu
        if not args.train_only:
            renders_per_s = world * args.steps / fwd_s
            dominant = max(stage_ms, key=stage_ms.get)
            counters, src = (load_counters(workload_key(P, W, H, args.s0, args.scene)) if args.ply is None
                             else (None, "not a synthetic workload"))
            out.update({
                "forward_ms_gpu": {"median": percentile(fwd_ms, 0.5), "p10": percentile(fwd_ms, 0.1), "p90": percentile(fwd_ms, 0.9)},
                "forward_renders_per_s": renders_per_s,
                "forward_mpixels_per_s": renders_per_s * N / 1e6,
                "forward_ms": 1e3 * fwd_s / args.steps,
                "stage_ms": stage_ms,
                "stage_algorithmic_bytes": ab,
                "stage_compulsory_bytes": cb,
                # the whole iteration against 8 TB/s, the honest figures first: what the memory controllers counted (PMC passes,
                # when they belong to this build), then the compulsory-byte model; section 8(d)'s per-instance model last
                "hbm_fraction_train_iter_counter": ((len(my_views) * sum(counters["per_launch_bytes"].values()) / (train_s / args.steps))
                                                    / (HBM_PEAK_GBS * 1e9)) if counters and counters.get("per_launch_bytes") else None,
                "hbm_fraction_train_iter": (len(my_views) * sum(cb.values()) / (train_s / args.steps)) / (HBM_PEAK_GBS * 1e9),
                "hbm_fraction_train_iter_8d": (len(my_views) * sum(ab.values()) / (train_s / args.steps)) / (HBM_PEAK_GBS * 1e9),
                "roofline": roofline_block(stage_ms, ab, cb, dominant, counters, src, dev, pix_inst, k7_info),
                "cpu_baseline": cpu,
            })
            if extra is not None:
                out["extra_configs"] = extra
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()
    if multi_gpu is not None and not multi_gpu["replicas_identical"]:
        raise SystemExit("bench.py: the ranks do NOT hold bit-identical gradients after the exchange (see the per-rank lines)")


if __name__ == "__main__":
    main()
