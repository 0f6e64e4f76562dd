"""-m gpu, round 6: view reuse (the second render() of a view runs the blend kernel alone -- VERDICT r05 next 3), the
compare primitive behind it (gsr_arrays_equal), and the device checks of every tensor argument of the binding (next 7)."""
import math

import pytest
import torch

from helpers import make_case, settings

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _PC:
    """The part of the reference's GaussianModel that render() reads (scene/gaussian_model.py:222-258): parameters
    behind activations, so get_opacity / get_scaling / get_rotation are FRESH tensors on every call."""

    def __init__(self, sc, dev):
        self._xyz = torch.nn.Parameter(sc["xyz"].to(dev))
        self._opacity = torch.nn.Parameter(torch.logit(sc["opacity"].clamp(1e-4, 1 - 1e-4)).to(dev))
        self._scaling = torch.nn.Parameter(torch.log(sc["scaling"]).to(dev))
        self._rotation = torch.nn.Parameter(sc["rotation"].to(dev))
        self._features = torch.nn.Parameter(sc["features"].to(dev))
        self.active_sh_degree = self.max_sh_degree = 3

    get_xyz = property(lambda s: s._xyz)
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_features = property(lambda s: s._features)


class _Pipe:
    compute_cov3D_python = False
    convert_SHs_python = False


def _two_renders(pc, cam, bg, mask):
    """threestudio/systems/GassuianEditor.py:166-191, unmodified call pattern."""
    from gaussianeditor_amd.gaussian_renderer import render

    a = render(cam, pc, _Pipe, bg)
    b = render(cam, pc, _Pipe, bg, override_color=mask)
    return a, b


@pytest.fixture
def reuse():
    import gaussianeditor_amd
    from gaussianeditor_amd.diff_gaussian_rasterization import _reuse

    was = gaussianeditor_amd.get_view_reuse()
    gaussianeditor_amd.set_view_reuse(True)
    _reuse.forget()
    for k in _reuse.stats:
        _reuse.stats[k] = 0
    yield _reuse
    _reuse.forget()
    gaussianeditor_amd.set_view_reuse(was)


@pytest.mark.parametrize("P,W,H,s0", [(20000, 512, 512, 0.03), (3000, 250, 131, 0.08)])
def test_second_render_of_a_view_is_served_by_the_blend_kernel_alone(reuse, P, W, H, s0):
    """The reference's double render through the unmodified render(): with view reuse the second image, its radii and its
    depth are bit-identical to two full renders; the first image is untouched; gradients of the first render too."""
    import gaussianeditor_amd

    case = make_case(P, W, H, seed=5, s0=s0)
    cam, bg = case["cam"], case["bg"].to(DEV)
    for a in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, a, getattr(cam, a).to(DEV))
    pc = _PC(case["sc"], DEV)
    mask = (torch.rand(P, 1, generator=torch.Generator().manual_seed(1)) > 0.6).float().repeat(1, 3).to(DEV)
    G = torch.rand(3, H, W, generator=torch.Generator().manual_seed(2)).to(DEV)

    def run():
        for p in (pc._xyz, pc._opacity, pc._scaling, pc._rotation, pc._features):
            p.grad = None
        a, b = _two_renders(pc, cam, bg, mask)
        (a["render"] * G).sum().backward()
        torch.cuda.synchronize()
        return a, b, [p.grad.clone() for p in (pc._xyz, pc._opacity, pc._scaling, pc._rotation, pc._features)]

    a1, b1, g1 = run()
    assert reuse.stats["hits"] == 1 and reuse.stats["compares"] == 1  # opacity / scaling / rotation: fresh tensors, compared
    gaussianeditor_amd.set_view_reuse(False)
    a0, b0, g0 = run()
    assert reuse.stats["hits"] == 1
    for k in ("render", "radii", "depth_3dgs"):
        assert torch.equal(a1[k], a0[k]), k
        assert torch.equal(b1[k], b0[k]), k
    assert torch.equal(b1["visibility_filter"], b0["visibility_filter"])
    for x, y in zip(g1, g0):  # (the backward's float atomics: run-to-run rounding only)
        assert torch.allclose(x, y, rtol=0, atol=2e-5 * float(y.abs().max()))


def test_view_reuse_misses_when_anything_changed(reuse):
    """An in-place parameter update (optimizer step), another camera, another image size, another scale modifier or a
    densified model between the two calls: the second render runs in full and equals a render without reuse."""
    import gaussianeditor_amd
    from gaussianeditor_amd.gaussian_renderer import render

    P, W, H = 8000, 256, 192
    case = make_case(P, W, H, seed=7, s0=0.05)
    cam, bg = case["cam"], case["bg"].to(DEV)
    cam2 = make_case(P, W, H, seed=7, s0=0.05, view=2)["cam"]
    for c in (cam, cam2):
        for a in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(c, a, getattr(c, a).to(DEV))
    pc = _PC(case["sc"], DEV)
    mask = torch.rand(P, 3, generator=torch.Generator().manual_seed(3)).to(DEV)

    def full(cam_, **kw):
        gaussianeditor_amd.set_view_reuse(False)
        try:
            return render(cam_, pc, _Pipe, bg, override_color=mask, **kw)["render"].detach().clone()
        finally:
            gaussianeditor_amd.set_view_reuse(True)

    # 1. in-place update of a parameter between the renders (what optimizer.step() does)
    render(cam, pc, _Pipe, bg)
    with torch.no_grad():
        pc._opacity.add_(0.25)
    h = reuse.stats["hits"]
    out = render(cam, pc, _Pipe, bg, override_color=mask)["render"]
    assert reuse.stats["hits"] == h and torch.equal(out, full(cam))
    # 2. ... of the positions (the SAME tensor object on both calls: only its version counter tells)
    render(cam, pc, _Pipe, bg)
    with torch.no_grad():
        pc._xyz.mul_(1.01)
    out = render(cam, pc, _Pipe, bg, override_color=mask)["render"]
    assert reuse.stats["hits"] == h and torch.equal(out, full(cam))
    # 3. another camera; 4. another scale modifier
    render(cam, pc, _Pipe, bg)
    out = render(cam2, pc, _Pipe, bg, override_color=mask)["render"]
    assert reuse.stats["hits"] == h and torch.equal(out, full(cam2))
    render(cam, pc, _Pipe, bg)
    out = render(cam, pc, _Pipe, bg, scaling_modifier=0.5, override_color=mask)["render"]
    assert reuse.stats["hits"] == h and torch.equal(out, full(cam, scaling_modifier=0.5))
    # 5. the same values in a camera built anew (other tensor objects): compared by content -> hit
    import copy

    cam3 = copy.copy(cam)
    cam3.world_view_transform = cam.world_view_transform.clone()
    cam3.full_proj_transform = cam.full_proj_transform.clone()
    render(cam, pc, _Pipe, bg)
    out = render(cam3, pc, _Pipe, bg, override_color=mask)["render"]
    assert reuse.stats["hits"] == h + 1 and torch.equal(out, full(cam))
    # 6. under no_grad (the web UI's frames, webui.py:693-713): activations are fresh leaves -- compared, hit
    with torch.no_grad():
        render(cam, pc, _Pipe, bg)
        out = render(cam, pc, _Pipe, bg, override_color=mask)["render"]
    assert reuse.stats["hits"] == h + 2 and torch.equal(out, full(cam))


def test_backward_through_a_reused_render_runs_the_skipped_forward(reuse):
    """Nobody in GaussianEditor differentiates through the semantic image -- but a caller may: the gradients of a served
    render equal those of a full colour-override render (run-to-run rounding of the float atomics)."""
    import gaussianeditor_amd
    from gaussianeditor_amd.gaussian_renderer import render

    P, W, H = 6000, 160, 128
    case = make_case(P, W, H, seed=9, s0=0.06)
    cam, bg = case["cam"], case["bg"].to(DEV)
    for a in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, a, getattr(cam, a).to(DEV))
    pc = _PC(case["sc"], DEV)
    col = torch.rand(P, 3, generator=torch.Generator().manual_seed(4)).to(DEV).requires_grad_(True)
    G = torch.rand(3, H, W, generator=torch.Generator().manual_seed(5)).to(DEV)
    params = (pc._xyz, pc._opacity, pc._scaling, pc._rotation, col)

    def grads(on):
        gaussianeditor_amd.set_view_reuse(on)
        for p in params:
            p.grad = None
        render(cam, pc, _Pipe, bg)
        out = render(cam, pc, _Pipe, bg, override_color=col)
        (out["render"] * G).sum().backward()
        return [p.grad.clone() for p in params] + [out["viewspace_points"].grad.clone()]

    h = reuse.stats["hits"]
    g1 = grads(True)
    assert reuse.stats["hits"] == h + 1
    g0 = grads(False)
    for x, y in zip(g1, g0):
        assert float(y.abs().max()) > 0 and torch.allclose(x, y, rtol=0, atol=2e-5 * float(y.abs().max()))


def test_arrays_equal():
    from gaussianeditor_amd.diff_gaussian_rasterization import _C

    g = torch.Generator().manual_seed(0)
    for n in (1, 3, 4, 1000, 4099, 3_000_001):
        a = torch.rand(n, generator=g).to(DEV)
        b = a.clone()
        assert _C.arrays_equal([(a, b)])
        b[-1] += 1.0
        assert not _C.arrays_equal([(a, b)])
        b = a.clone()
        b[n // 2] = float("nan")
        assert not _C.arrays_equal([(a, b)])
    # several pairs, one of them unaligned to 16 bytes; NaN equals NaN bit for bit; identical pointers are not read
    base = torch.rand(70001, generator=g).to(DEV)
    base[5] = float("nan")
    u, v = base[1:], base[1:].clone()
    w = torch.randint(0, 1 << 30, (977, 3), generator=g, dtype=torch.int32).to(DEV)
    assert _C.arrays_equal([(u, v), (w, w.clone()), (base, base)])
    v[-3] = 0.0
    assert not _C.arrays_equal([(w, w.clone()), (u, v)])
    assert _C.arrays_equal([])
    assert not _C.arrays_equal([(base[:10], base[:11])])  # shapes differ: unequal without a launch


def test_every_tensor_argument_is_device_checked():
    """rasterize_points.cu:46-48 checks `means3D` only; a `bg` or a matrix left on the host would hand a foreign pointer
    to a kernel.  Here every argument is checked against the device of `means3D`, by name."""
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    case = make_case(500, 64, 64, seed=1)
    sc, rs = case["sc"], settings(case, DEV)
    x, o, f = sc["xyz"].to(DEV), sc["opacity"].to(DEV), sc["features"].to(DEV)
    s, r = sc["scaling"].to(DEV), sc["rotation"].to(DEV)

    def call(rs_=rs, **kw):
        a = dict(means3D=x, means2D=torch.zeros_like(x), opacities=o, shs=f, scales=s, rotations=r)
        a.update(kw)
        return GaussianRasterizer(rs_)(**a)

    call()
    for field, name in (("bg", "bg"), ("viewmatrix", "viewmatrix"), ("projmatrix", "projmatrix"), ("campos", "campos")):
        bad = rs._replace(**{field: getattr(rs, field).cpu()})
        with pytest.raises(RuntimeError, match=f"`{name}` is on cpu"):
            call(bad)
    for kw, name in ((dict(opacities=o.cpu()), "opacity"), (dict(shs=f.cpu()), "sh"), (dict(scales=s.cpu()), "scales"),
                     (dict(rotations=r.cpu()), "rotations")):
        with pytest.raises(RuntimeError, match=f"`{name}` is on cpu"):
            call(**kw)
    with pytest.raises(RuntimeError, match="expected float32 for `opacity`"):
        call(opacities=o.double())
    # backward: a gradient image on the host
    xg = x.clone().requires_grad_(True)
    color, _, _ = GaussianRasterizer(rs)(means3D=xg, means2D=torch.zeros_like(x), opacities=o, shs=f, scales=s, rotations=r)
    from gaussianeditor_amd.diff_gaussian_rasterization import _C
    node = color.grad_fn
    with pytest.raises(RuntimeError, match="`dL_dout_color` is on cpu"):
        _C.rasterize_gaussians_backward(rs.bg, x, torch.zeros(500, dtype=torch.int32, device=DEV), torch.empty(0, device=DEV), s, r,
                                        1.0, torch.empty(0, device=DEV), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                                        torch.zeros(3, 64, 64), f, 3, rs.campos, node.saved_tensors[7], node.num_rendered,
                                        node.saved_tensors[8], node.saved_tensors[9], False)
    # apply_weights: the mask image and the counters
    w = torch.zeros(500, 1, device=DEV)
    cnt = torch.zeros(500, 1, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="`image_weights` is on cpu"):
        GaussianRasterizer(rs).apply_weights(x, None, o, None, w, s, r, None, cnt, torch.ones(1, 64, 64))
    with pytest.raises(RuntimeError, match="`cnt` is on cpu"):
        GaussianRasterizer(rs).apply_weights(x, None, o, None, w, s, r, None, cnt.cpu(), torch.ones(1, 64, 64, device=DEV))
    if torch.cuda.device_count() > 1:  # a cross-device argument (only where two GPUs are visible)
        with pytest.raises(RuntimeError, match="`sh` is on cuda:1"):
            call(shs=f.to("cuda:1"))


# ---- full-size parity beyond ring view 0 and beyond 1080p ---------------------------------------------------
@pytest.mark.parametrize("view", range(1, 8))
def test_three_way_parity_headline_every_ring_view(oracle, view):
    """BASELINE configs[3] is EIGHT views of the 1 M-Gaussian scene at 1920x1080; rounds 2-5 pinned view 0 only.  The other
    seven, three ways: the reference's own kernels (contraction-free gfx950 build) == oracle == product -- integers and lists
    bit for bit, images / depth / every gradient within 1e-5 incl. the per-row bar (test_gpu_round2.three_way)."""
    from helpers import seed_gradient
    from test_gpu_round2 import three_way

    case = make_case(1_000_000, 1920, 1080, seed=0, s0=0.01, view=view, nviews=8, bg=(0.0, 0.0, 0.0))
    three_way(oracle, case, seed_gradient(1080, 1920, view), 3)


def test_three_way_parity_edit_loop_workload_1M_at_512(oracle):
    """The exact workload of `extra_configs.C3_edit_loop_512_1M` / `C5_apply_weights...`: 1 M Gaussians seen through the
    editor's 512 x 512 image (K12 on the same workload: tests/test_gpu_reference.py::test_apply_weights_vs_reference)
    -- ~2 250 list entries per tile, so the forward's SPLIT items (a quadrant cut into 2 or 4 items:
    1 024 tiles for 4 096 persistent waves) and its checkpoints / the backward's list segments are live TOGETHER, which no
    smaller case exercises at this density."""
    from helpers import seed_gradient
    from test_gpu_round2 import three_way

    case = make_case(1_000_000, 512, 512, seed=0, s0=0.01, view=0, nviews=8, bg=(0.0, 0.0, 0.0))
    three_way(oracle, case, seed_gradient(512, 512, 0), 3)


def test_accumulator_table_kept_across_backwards_is_left_zero():
    """GSR_FLAG_ACC_SELF_CLEAN: the binding keeps the blend backward's accumulator table between backwards (per device and
    stream), K8+K9 puts every row K7 touched back to zero, and no clear runs in front of the next backward.  Over a sequence
    of different views (Gaussians go from touched to untouched and back) every gradient equals the one a freshly cleared
    table gives, and the kept table is all zero after every backward."""
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer, _C

    P, W, H = 20000, 256, 192
    sc = make_case(P, W, H, seed=3, s0=0.04)["sc"]
    leaves = {k: sc[k].to(DEV).requires_grad_(True) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
    G = torch.rand(3, H, W, generator=torch.Generator().manual_seed(8)).to(DEV)

    def grads(view):
        case = make_case(P, W, H, seed=3, s0=0.04, view=view, nviews=5)
        for t in leaves.values():
            t.grad = None
        m2 = torch.zeros_like(leaves["xyz"], requires_grad=True)
        color, _, _ = GaussianRasterizer(settings(case, DEV))(leaves["xyz"], m2, leaves["opacity"], shs=leaves["features"],
                                                               scales=leaves["scaling"], rotations=leaves["rotation"])
        (color * G).sum().backward()
        torch.cuda.synchronize()
        return [t.grad.clone() for t in leaves.values()] + [m2.grad.clone()]

    was = _C._ACC_PERSIST
    try:
        _C._ACC_PERSIST = True
        _C._ACC_TABLES.clear()
        kept = []
        for view in (0, 3, 1, 3, 4, 0):
            kept.append(grads(view))
            tables = list(_C._ACC_TABLES.values())
            assert len(tables) == 1 and tables[0].numel() == 16 * P and float(tables[0].abs().max()) == 0.0
        first = tables[0].data_ptr()
        _C._ACC_PERSIST = False
        for view, k in zip((0, 3, 1, 3, 4, 0), kept):
            fresh = grads(view)
            for a, b in zip(k, fresh):  # (the backward's float atomics: run-to-run rounding only)
                assert float(b.abs().max()) > 0 and torch.allclose(a, b, rtol=0, atol=2e-5 * float(b.abs().max()))
        assert list(_C._ACC_TABLES.values())[0].data_ptr() == first  # (untouched while the option is off)
    finally:
        _C._ACC_PERSIST = was
        _C._ACC_TABLES.clear()


def test_blend_backward_row_mask_equals_the_nonzero_rows_of_its_table():
    """gsr_blend_backward's `touched` mask (what the multi-GPU exchange plans its messages from, without a pass over the
    64 P bytes of the table): cleared by the call, 1 exactly for the Gaussians whose accumulator row is not all zero --
    on a small image, on SPLIT / half-tile items, and when nothing is rendered."""
    import ctypes

    from gaussianeditor_amd import _native
    from gaussianeditor_amd.diff_gaussian_rasterization import _C
    from helpers import seed_gradient

    L = _native.lib()
    for P, W, H, s0, seed in ((20011, 320, 200, 0.02, 6), (5000, 640, 360, 0.4, 21), (977, 33, 17, 0.3, 8)):
        case = make_case(P, W, H, seed=seed, s0=s0)
        sc, rs = case["sc"], settings(case, DEV)
        d = lambda t: t.to(DEV)  # noqa: E731
        e = torch.empty(0, device=DEV)
        R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
            rs.bg, d(sc["xyz"]), e, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), 1.0, e, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, H, W, d(sc["features"]), 3, rs.campos, False, False)
        G = d(seed_gradient(H, W, seed) * (H * W))
        acc = torch.full((P, 16), float("nan"), device=DEV)
        touched = torch.full((((P + 15) // 16) * 16,), 7, dtype=torch.uint8, device=DEV)
        sp = torch.cuda.current_stream().cuda_stream
        p = lambda t: t.data_ptr()  # noqa: E731
        _native.check("k7", L.gsr_blend_backward(sp, P, R, W, H, p(rs.bg), p(geom), p(binning), p(img), p(G), p(acc), p(touched), 4))
        torch.cuda.synchronize()
        rows = (acc != 0).any(dim=1)
        assert bool(torch.isfinite(acc).all()) and 0 < int(rows.sum()) < P
        assert torch.equal(touched[:P] != 0, rows) and int(touched[:P].max()) == 1
        # ... and the plan built from it counts exactly those rows
        plan, count = _C.view_message_plan_blend(touched[:P])
        torch.cuda.synchronize()
        assert int(count.item()) == int(rows.sum())
    # nothing rendered: the mask is cleared all the same
    touched.fill_(9)
    _native.check("k7", L.gsr_blend_backward(sp, P, 0, W, H, p(rs.bg), p(geom), None, p(img), p(G), p(acc), p(touched), 4))
    torch.cuda.synchronize()
    assert not bool(touched[:P].any())


def test_forward_in_two_halves_equals_the_whole_forward():
    """gsr_preprocess_begin / _end (ABI 6; _C.rasterize_gaussians_begin / _finish): the same kernels as gsr_preprocess, so
    every output is bit-identical to rasterize_gaussians() -- also with two views in flight on one thread, finished in the
    other order, and with the halves of the views on two streams as the pipelined view batch issues them.  A pending
    forward is finished once, on the stream it began on."""
    from gaussianeditor_amd.diff_gaussian_rasterization import _C

    P, W, H = 30000, 320, 200
    cases = [make_case(P, W, H, seed=5, s0=0.03, view=v, nviews=3) for v in range(3)]
    sc = cases[0]["sc"]
    d = lambda t: t.to(DEV).contiguous()  # noqa: E731
    xyz, op, sh, scl, rot = (d(sc[k]) for k in ("xyz", "opacity", "features", "scaling", "rotation"))
    absent = xyz.new_empty(0)

    def call(fn, case):
        rs = settings(case, DEV)
        return fn(rs.bg, xyz, absent, op, scl, rot, rs.scale_modifier, absent, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                  rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)

    whole = [call(_C.rasterize_gaussians, c) for c in cases]

    def same(got, want):
        assert got[0] == want[0] > 0  # num_rendered
        for a, b in zip(got[1:4], want[1:4]):  # color, depth, radii
            assert torch.equal(a, b)

    # two in flight on the current stream, finished in the other order
    pa, pb = call(_C.rasterize_gaussians_begin, cases[0]), call(_C.rasterize_gaussians_begin, cases[1])
    fb, fa = _C.rasterize_gaussians_finish(pb), _C.rasterize_gaussians_finish(pa)
    same(fa, whole[0])
    same(fb, whole[1])
    with pytest.raises(RuntimeError, match="finished already"):
        _C.rasterize_gaussians_finish(pa)
    # the state a split forward leaves serves the backward like the whole forward's
    G = torch.rand(3, H, W, generator=torch.Generator().manual_seed(2)).to(DEV)
    rs = settings(cases[0], DEV)

    def backward(f):
        R, color, depth, radii, geom, binning, img = f
        return _C.rasterize_gaussians_backward(rs.bg, xyz, radii, absent, scl, rot, rs.scale_modifier, absent, rs.viewmatrix,
                                               rs.projmatrix, rs.tanfovx, rs.tanfovy, G, sh, rs.sh_degree, rs.campos, geom, R,
                                               binning, img, False)

    for a, b in zip(backward(fa), backward(whole[0])):
        if a is not None and a.numel():  # (float atomics: run-to-run rounding only)
            assert torch.allclose(a, b, rtol=0, atol=2e-5 * float(b.abs().max()) + 1e-30)
    # two streams, view v + 1 begun before view v is finished (the pipelined batch's order)
    torch.cuda.synchronize()
    S = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    with torch.cuda.stream(S[0]):
        begun = call(_C.rasterize_gaussians_begin, cases[0])
    got = []
    for v in range(3):
        nxt = None
        if v + 1 < 3:
            with torch.cuda.stream(S[(v + 1) % 2]):
                nxt = call(_C.rasterize_gaussians_begin, cases[v + 1])
        if v == 1:
            with pytest.raises(RuntimeError, match="must run on the stream"):
                _C.rasterize_gaussians_finish(begun)  # (the current stream is not the one it began on)
            assert begun.ticket is not None  # still pending
        with torch.cuda.stream(S[v % 2]):
            got.append(_C.rasterize_gaussians_finish(begun))
        begun = nxt
    torch.cuda.synchronize()
    for g, w in zip(got, whole):
        same(g, w)


@pytest.mark.parametrize("C,W,H,s0,opaque", [(1, 250, 131, 0.04, False), (3, 96, 80, 0.3, True), (2, 200, 120, 0.15, True),
                                             (1, 64, 48, 0.5, True)])
def test_apply_weights_grouped_loop_edges(oracle, C, W, H, s0, opaque):
    """K12's inner loop walks four survivors per iteration on selects (round 6).  What the parity cases with 0 / 1 masks do
    not reach: real-valued mask values (sums within the re-association of the atomics), scenes of large opaque splats in
    which every pixel of a quadrant saturates in the middle of a chunk (the loop breaks between two groups and flushes only
    what it wrote), survivor counts that are no multiple of four (padding), ragged images, and a NaN in the mask (it reaches
    exactly the Gaussians blended at that pixel).  `cnt` is an integer: identical to the oracle's."""
    import numpy as np

    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    P = 4000
    case = make_case(P, W, H, seed=21, s0=s0)
    sc, cam = case["sc"], case["cam"]
    if opaque:
        sc["opacity"] = torch.full_like(sc["opacity"], 0.99)
    gen = torch.Generator().manual_seed(5)
    mask = torch.rand(C, H, W, generator=gen)
    mask[0, H // 2, W // 3] = float("nan")
    w_ref = np.zeros((P, C), np.float32)
    c_ref = np.zeros((P,), np.int32)
    oracle.apply_weights(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], None, cam.world_view_transform,
                         cam.full_proj_transform, cam.camera_center, W, H, case["tfx"], case["tfy"], mask, w_ref, c_ref)
    w = torch.zeros((P, C), device=DEV)
    cnt = torch.zeros((P, 1), dtype=torch.int32, device=DEV)
    GaussianRasterizer(settings(case, DEV, D=0)).apply_weights(sc["xyz"].to(DEV), None, sc["opacity"].to(DEV), None, w,
                                                             sc["scaling"].to(DEV), sc["rotation"].to(DEV), None, cnt,
                                                             mask.to(DEV))
    torch.cuda.synchronize()
    got_w, got_c = w.cpu().numpy(), cnt.cpu().numpy().reshape(-1)
    assert c_ref.sum() > 0 and np.array_equal(got_c, c_ref)
    assert np.array_equal(np.isnan(got_w), np.isnan(w_ref)) and np.isnan(w_ref).any() and not np.isnan(w_ref).all()
    ok = ~np.isnan(w_ref)
    # (sums of up to ~10^3 real values per Gaussian, added sequentially by the oracle and as trees + atomics here)
    assert np.abs(got_w[ok] - w_ref[ok]).max() <= 2e-5 * max(1.0, float(np.abs(w_ref[ok]).max()))
    if opaque:  # the scene does what it is here for: most Gaussians in the frustum are hidden behind saturated pixels
        assert (c_ref > 0).sum() < 0.5 * P
