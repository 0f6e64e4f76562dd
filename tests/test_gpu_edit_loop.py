"""-m gpu: the editing LOOP composed end to end (missing 2 / next 1c) -- BASELINE configs[2] (edit-n2n) and
configs[4] (del-ctn) are loops, not kernels: per step two renders of every camera of the batch (SH image, then the
semantic mask splatted with override_color), an image loss, backward, the gradient mask, Adam; every k steps
densify-and-prune; before the first step apply_weights over a ring of views turns 2-D masks into the per-Gaussian mask.

Two complete stacks run the same loop from the same scene and are compared step by step:

  product    gaussianeditor_amd.gaussian_renderer.render (HIP rasterizer, fused semantic image) + FusedMaskedAdam with
             the row mask + densify.clone_rows / cat_tensors_to_optimizer / prune_optimizer (gsr_append_rows,
             gsr_compact_*) + GaussianRasterizer.apply_weights
  reference  the reference's own rasterizer sources compiled for gfx950 (oracle/_ref, contraction-free build), forward
             and backward, behind a torch.autograd.Function + torch.optim.Adam(eps=1e-15) with the reference's gradient
             hooks + the reference's densification surgery restated with torch.cat / boolean indexing

The control flow both stacks share is the reference's, restated with its line numbers:
  threestudio/systems/GassuianEditor.py:95-137 (mask from apply_weights), :155-224 (forward over the batch), :251-281
  (on_before_optimizer_step: view-space gradient sum, max radii, densification stats, densify_and_prune);
  gaussiansplatting/scene/gaussian_model.py:336-380 (training_setup), :553-671 (optimizer surgery), :673-728
  (densify_and_split), :730-766 (densify_and_clone), :768-809 (densify_and_prune), :811-815 (add_densification_stats),
  :841-856 (apply_grad_mask); gaussiansplatting/utils/general_utils.py:78-99 (build_rotation).
The diffusion guidance is replaced by fixed synthetic target images (there are no weights offline).
"""
import math

import numpy as np
import pytest
import torch

from helpers import make_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W = H = 512
STEPS, DENSIFY_EVERY, BATCH, TRACE_VIEWS = 20, 5, 2, 12
# densification constants: the editor's (configs/edit-n2n.yaml, GassuianEditor.py:275-281) except max_grad (5 there: with
# these synthetic targets nothing would ever densify), max_screen_size (5 px there: would prune most of this scene) and
# min_opacity (0.005 there; 0.02 here so that the 2 % of the scene made faint below is pruned whatever five Adam steps do)
MAX_GRAD, MAX_DENSIFY_PERCENT, MIN_OPACITY, EXTENT, MAX_SCREEN, PERCENT_DENSE = 1e-12, 0.01, 0.02, 1.0, 20, 0.01
LR = dict(xyz=0.00016 * EXTENT, f_dc=0.0025, f_rest=0.0025 / 20.0, opacity=0.05, scaling=0.005, rotation=0.001)
LAMBDA_L1 = 10.0


def _ref_lib():
    from helpers import require_ref

    return require_ref("nofma")


class _RefRasterize(torch.autograd.Function):
    """The reference's rasterizer (oracle/_ref) as an autograd op: diff_gaussian_rasterization/__init__.py:50-225."""

    @staticmethod
    def forward(ctx, means3D, means2D, opacity, shs, colors, scales, rotations, cam, bg, D):
        R = _ref_lib().Reference("nofma", DEV)
        use_sh = colors is None
        out = R.forward(means3D, scales, rotations, opacity, shs if use_sh else None, None if use_sh else colors, None,
                        cam.world_view_transform, cam.full_proj_transform, cam.camera_center, bg, W, H,
                        math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), 1.0, D)
        ctx.R, ctx.use_sh, ctx.M = R, use_sh, (shs.shape[1] if use_sh else 0)
        ctx.mark_non_differentiable(out["radii"])
        return out["color"], out["radii"]

    @staticmethod
    def backward(ctx, dL_dcolor, _):
        g = ctx.R.backward(dL_dcolor.contiguous())
        P = g["dL_dmeans3D"].shape[0]
        return (g["dL_dmeans3D"], g["dL_dmeans2D"], g["dL_dopacity"].view(P, 1),
                g["dL_dsh"].view(P, ctx.M, 3) if ctx.use_sh else None, None if ctx.use_sh else g["dL_dcolors"],
                g["dL_dscales"], g["dL_drotations"], None, None, None)


def _build_rotation(r):  # general_utils.py:78-99
    q = r / torch.sqrt((r * r).sum(dim=1))[:, None]
    a, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - a * z)
    R[:, 0, 2] = 2 * (x * z + a * y)
    R[:, 1, 0] = 2 * (x * y + a * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - a * x)
    R[:, 2, 0] = 2 * (x * z - a * y)
    R[:, 2, 1] = 2 * (y * z + a * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
MASKED = ("xyz", "f_dc", "f_rest", "opacity", "scaling")  # apply_grad_mask, gaussian_model.py:841-856 (not the rotation)


class Model:
    """The part of GaussianModel the loop touches, on either stack."""

    def __init__(self, sc, stack):
        self.stack = stack
        d = lambda t: t.to(DEV).clone().contiguous()  # noqa: E731
        f = d(sc["features"])
        self.p = dict(xyz=d(sc["xyz"]), f_dc=f[:, :1].contiguous(), f_rest=f[:, 1:].contiguous(),
                      opacity=torch.logit(d(sc["opacity"])), scaling=torch.log(d(sc["scaling"])), rotation=d(sc["rotation"]))
        self.p = {k: torch.nn.Parameter(v.requires_grad_(True)) for k, v in self.p.items()}
        self.active_sh_degree = self.max_sh_degree = 3
        P = self.P
        self.xyz_gradient_accum = torch.zeros((P, 1), device=DEV)
        self.denom = torch.zeros((P, 1), device=DEV)
        self.max_radii2D = torch.zeros((P,), device=DEV)
        self.mask = torch.ones(P, dtype=torch.bool, device=DEV)
        product = stack.startswith("product")
        groups = [dict(params=[self.p[k]], lr=LR[k], name=k, **({"masked": k in MASKED} if product else {}))
                  for k in NAMES]  # training_setup, gaussian_model.py:336-380
        if product:
            from gaussianeditor_amd.optim import FusedMaskedAdam

            self.optimizer = FusedMaskedAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.hooks = []
        self.oa = None
        if stack == "product-arena":  # every per-Gaussian tensor a view of ONE buffer (gaussianeditor_amd/arena.py)
            from gaussianeditor_amd.arena import OptimizerArena

            self.oa = OptimizerArena(self.optimizer, extra=dict(xyz_gradient_accum=self.xyz_gradient_accum, denom=self.denom,
                                                                max_radii2D=self.max_radii2D, mask=self.mask))
            self._from_arena()

    def _from_arena(self):
        self.p = dict(self.oa.params())
        e = self.oa.extra
        self.xyz_gradient_accum, self.denom, self.max_radii2D, self.mask = e["xyz_gradient_accum"], e["denom"], e["max_radii2D"], e["mask"]

    P = property(lambda s: int(s.p["xyz"].shape[0]))
    get_xyz = property(lambda s: s.p["xyz"])
    get_opacity = property(lambda s: torch.sigmoid(s.p["opacity"]))
    get_scaling = property(lambda s: torch.exp(s.p["scaling"]))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s.p["rotation"]))
    get_features = property(lambda s: torch.cat((s.p["f_dc"], s.p["f_rest"]), dim=1))

    def apply_grad_mask(self, mask):  # gaussian_model.py:837-856
        if self.oa is not None and mask.data_ptr() != self.oa.extra["mask"].data_ptr():
            self.oa.extra["mask"].copy_(mask)  # the mask lives in the arena, too
            mask = self.oa.extra["mask"]
        self.mask = mask
        for h in self.hooks:
            h.remove()
        self.hooks = []
        if self.stack.startswith("product"):
            self.optimizer.set_row_mask(mask)  # the mask is applied inside the fused step
            return
        for k in MASKED:
            t = self.p[k]
            self.hooks.append(t.register_hook(
                lambda grad: grad * (self.mask[:, None] if grad.ndim == 2 else self.mask[:, None, None])))

    # --- tensor surgery: the ONLY part that differs between the stacks besides the rasterizer and the optimizer
    def _cat(self, ext):  # cat_tensors_to_optimizer, gaussian_model.py:609-641
        if self.stack == "product":
            from gaussianeditor_amd.densify import cat_tensors_to_optimizer

            new = cat_tensors_to_optimizer(self.optimizer, ext)
        else:
            new = {}
            for group in self.optimizer.param_groups:
                e = ext[group["name"]]
                st = self.optimizer.state.get(group["params"][0], None)
                if st is not None:
                    st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(e)), dim=0)
                    st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(e)), dim=0)
                    del self.optimizer.state[group["params"][0]]
                    group["params"][0] = torch.nn.Parameter(torch.cat((group["params"][0], e), dim=0).requires_grad_(True))
                    self.optimizer.state[group["params"][0]] = st
                else:
                    group["params"][0] = torch.nn.Parameter(torch.cat((group["params"][0], e), dim=0).requires_grad_(True))
                new[group["name"]] = group["params"][0]
        self.p = dict(new)

    def _prune(self, remove):  # prune_points + _prune_optimizer, gaussian_model.py:568-607
        keep = ~remove
        if self.oa is not None:
            self.oa.prune(keep)
            self._from_arena()
        elif self.stack == "product":
            from gaussianeditor_amd.densify import compact_rows, prune_optimizer

            self.p = dict(prune_optimizer(self.optimizer, keep))
            self.xyz_gradient_accum, self.denom, self.max_radii2D, self.mask = compact_rows(
                [self.xyz_gradient_accum, self.denom, self.max_radii2D, self.mask], keep)
        else:
            new = {}
            for group in self.optimizer.param_groups:
                st = self.optimizer.state.get(group["params"][0], None)
                if st is not None:
                    st["exp_avg"] = st["exp_avg"][keep]
                    st["exp_avg_sq"] = st["exp_avg_sq"][keep]
                    del self.optimizer.state[group["params"][0]]
                    group["params"][0] = torch.nn.Parameter(group["params"][0][keep].requires_grad_(True))
                    self.optimizer.state[group["params"][0]] = st
                else:
                    group["params"][0] = torch.nn.Parameter(group["params"][0][keep].requires_grad_(True))
                new[group["name"]] = group["params"][0]
            self.p = new
            self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
            self.denom = self.denom[keep]
            self.max_radii2D = self.max_radii2D[keep]
            self.mask = self.mask[keep]

    def _postfix(self, ext, new_mask):  # densification_postfix, gaussian_model.py:643-671 (+ the mask's cat, :751 / :705-706)
        if self.oa is not None:
            self.oa.append(ext, extra=dict(mask=new_mask))  # new rows behind the live ones; the statistics get zero rows ...
            self._from_arena()
            self.xyz_gradient_accum.zero_()                 # ... and are reset as a whole, as the reference does
            self.denom.zero_()
            self.max_radii2D.zero_()
            return
        self._cat(ext)
        self.xyz_gradient_accum = torch.zeros((self.P, 1), device=DEV)
        self.denom = torch.zeros((self.P, 1), device=DEV)
        self.max_radii2D = torch.zeros((self.P,), device=DEV)
        self.mask = torch.cat([self.mask, new_mask], dim=0)

    @torch.no_grad()
    def densify_and_prune(self, seed):  # gaussian_model.py:768-809
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        grads[~self.mask] = 0.0
        valid_percent = len(grads.nonzero()) * MAX_DENSIFY_PERCENT / grads.shape[0]
        threshold = torch.quantile(grads, 1 - valid_percent)
        grads[grads < threshold] = 0.0
        # densify_and_clone, :730-766
        sel = (torch.norm(grads, dim=-1) >= MAX_GRAD) & (self.get_scaling.max(dim=1).values <= PERCENT_DENSE * EXTENT)
        n_clone = int(sel.sum())
        self._postfix({k: self.p[k][sel] for k in NAMES}, self.mask[sel])
        # densify_and_split, :673-728 (N = 2)
        n_init = self.P
        padded = torch.zeros((n_init,), device=DEV)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= MAX_GRAD) & (self.get_scaling.max(dim=1).values > PERCENT_DENSE * EXTENT)
        n_split = int(sel.sum())
        stds = self.get_scaling[sel].repeat(2, 1)
        torch.manual_seed(seed)  # the reference draws from the process RNG; both stacks draw the same numbers
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=DEV), std=stds)
        rots = _build_rotation(self.p["rotation"][sel]).repeat(2, 1, 1)
        ext = dict(xyz=torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(2, 1),
                   scaling=torch.log(self.get_scaling[sel].repeat(2, 1) / (0.8 * 2)), rotation=self.p["rotation"][sel].repeat(2, 1),
                   f_dc=self.p["f_dc"][sel].repeat(2, 1, 1), f_rest=self.p["f_rest"][sel].repeat(2, 1, 1),
                   opacity=self.p["opacity"][sel].repeat(2, 1))
        self._postfix(ext, torch.cat([self.mask[sel]] * 2, dim=0))
        self._prune(torch.cat((sel, torch.zeros(2 * n_split, device=DEV, dtype=torch.bool))))
        # prune, :787-797
        prune = (self.get_opacity < MIN_OPACITY).squeeze()
        prune = prune | (self.max_radii2D > MAX_SCREEN) | (self.get_scaling.max(dim=1).values > 0.1 * EXTENT)
        prune = prune & self.mask
        n_prune = int(prune.sum())
        self._prune(prune)
        self.apply_grad_mask(self.mask)  # remove_grad_mask + apply_grad_mask, :803-804
        return n_clone, n_split, n_prune


def _cam_to(cam):
    return cam.to(DEV)


def _render(model, cam, bg, override_color=None, semantic_color=None):
    """render(), gaussiansplatting/gaussian_renderer/__init__.py:45-150, on the stack of `model`."""
    if model.stack.startswith("product"):
        from types import SimpleNamespace

        from gaussianeditor_amd.gaussian_renderer import render

        pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
        return render(cam, model, pipe, bg, override_color=override_color, semantic_color=semantic_color)
    xyz = model.get_xyz
    ssp = torch.zeros_like(xyz, requires_grad=True) + 0
    if ssp.requires_grad:  # (not under no_grad: the semantic render)
        ssp.retain_grad()
    shs = model.get_features if override_color is None else None
    color, radii = _RefRasterize.apply(xyz, ssp, model.get_opacity, shs, override_color, model.get_scaling,
                                       model.get_rotation, cam, bg, model.active_sh_degree)
    return dict(render=color, viewspace_points=ssp, visibility_filter=radii > 0, radii=radii)


def _trace_mask(model, cams, masks2d):
    """GassuianEditor.py:95-137: weights / (cnt + 1e-7) > mask_thres after apply_weights over the views."""
    P = model.P
    weights = torch.zeros((P, 1), device=DEV)
    cnt = torch.zeros((P, 1), dtype=torch.int32, device=DEV)
    with torch.no_grad():
        for cam, m in zip(cams, masks2d):
            if model.stack.startswith("product"):
                from gaussianeditor_amd.gaussian_renderer import camera2rasterizer

                camera2rasterizer(cam, torch.zeros(3, device=DEV)).apply_weights(
                    model.get_xyz, None, model.get_opacity, None, weights, model.get_scaling, model.get_rotation, None, cnt, m)
            else:
                R = _ref_lib().Reference("nofma", DEV)
                c1 = cnt.view(-1)
                R.apply_weights(model.get_xyz, model.get_scaling, model.get_rotation, model.get_opacity,
                                cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H,
                                math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), m, weights, c1)
    ratio = weights / (cnt + 1e-7)
    return (ratio > 0.5)[:, 0], weights, cnt


def _run(stack, sc, cams, trace_cams, masks2d, targets, record):
    torch.manual_seed(0)
    model = Model(sc, stack)
    bg = torch.zeros(3, device=DEV)
    sel, weights, cnt = _trace_mask(model, trace_cams, masks2d)
    record["trace"] = (sel.cpu(), weights.cpu(), cnt.cpu())
    model.apply_grad_mask(sel)
    record["P"], record["densify"], record["loss"], record["semantic"] = [model.P], [], [], []
    for step in range(STEPS):
        batch = [(step * BATCH + i) % len(cams) for i in range(BATCH)]
        vsp, radii, loss = [], None, 0.0
        for ci in batch:  # GassuianEditor.forward, :155-224
            cam = cams[ci]
            sem_color = model.mask[..., None].float().repeat(1, 3)
            if stack.startswith("product"):  # one pass: the SH image + the semantic image from the same preprocessing / lists
                out = _render(model, cam, bg, semantic_color=sem_color)
                semantic = out["semantic"]
            else:
                out = _render(model, cam, bg)
                with torch.no_grad():
                    semantic = _render(model, cam, bg, override_color=sem_color)["render"]
            vsp.append(out["viewspace_points"])
            radii = out["radii"] if radii is None else torch.max(radii, out["radii"])
            if step == 0:
                record["semantic"].append((torch.norm(semantic.detach(), dim=0) > 0.8).cpu())
            loss = loss + torch.nn.functional.l1_loss(out["render"], targets[ci]) / BATCH
        loss = LAMBDA_L1 * loss
        record["loss"].append(float(loss.detach()))
        model.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        with torch.no_grad():  # on_before_optimizer_step, :251-281
            vis = radii > 0
            g = torch.zeros_like(vsp[0])
            for v in vsp:
                g = g + v.grad
            model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis].float())
            model.xyz_gradient_accum[vis] += torch.norm(g[vis, :2], dim=-1, keepdim=True)  # add_densification_stats, :811-815
            model.denom[vis] += 1
            if step > 0 and step % DENSIFY_EVERY == 0:
                record["densify"].append(model.densify_and_prune(1234 + step))
        if not (step > 0 and step % DENSIFY_EVERY == 0):
            model.optimizer.step()  # (after a densification the new parameters carry no gradient: as in the reference, the
            #                          step of that iteration acts on nothing)
        record["P"].append(model.P)
    torch.cuda.synchronize()
    record["params"] = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in model.p.items()}
    record["mask"] = model.mask.cpu()
    record["arena_allocations"] = None if model.oa is None else model.oa.arena.allocations
    return record


def test_edit_loop_product_vs_reference_stack():
    _ref_lib()
    from gaussianeditor_amd.synth import ring_cameras

    case = make_case(20000, W, H, seed=4, s0=0.02)
    sc = case["sc"]
    sc["opacity"][::50] = 0.006  # faint (but above 1/255: they are blended, hence traced into the mask): the opacity prune (:787)
    cams = [_cam_to(c) for c in ring_cameras(8, W, H)]
    trace_cams = [_cam_to(c) for c in ring_cameras(TRACE_VIEWS, W, H)]
    g = torch.Generator().manual_seed(7)
    # 2-D masks: a disc in the middle of every view (0 / 1, 1 channel: apply_weights.cu C = 1); about half of the scene ends up selected
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    masks2d = [(((xx - W / 2 - 20 * math.cos(i)) ** 2 + (yy - H / 2) ** 2) < (0.2 * W) ** 2).float()[None].to(DEV)
               for i in range(TRACE_VIEWS)]
    # synthetic "edited" targets: a smooth colour field (no guidance network offline)
    targets = []
    for i in range(len(cams)):
        t = torch.stack([0.5 + 0.5 * torch.sin(xx / (23.0 + i) + c) * torch.cos(yy / (31.0 - i) + 2 * c) for c in range(3)])
        targets.append((0.15 + 0.7 * t.float() * torch.rand(1, generator=g).item()).to(DEV))
    a = _run("reference", sc, cams, trace_cams, masks2d, targets, {})
    b = _run("product", sc, cams, trace_cams, masks2d, targets, {})
    c = _run("product-arena", sc, cams, trace_cams, masks2d, targets, {})
    # --- the arena (prune = compaction into the other half, densify = rows appended in place, Parameters re-pointed) changes
    # WHERE the tensors live, nothing else: the same counts, masks and losses as with fresh tensors, and parameters equal to
    # the backward's run-to-run spread (its float atomics re-associate); ONE buffer for the whole run
    assert c["P"] == b["P"] and c["densify"] == b["densify"] and torch.equal(c["mask"], b["mask"])
    assert c["arena_allocations"] == 1
    for k in NAMES:
        assert np.abs(c["params"][k] - b["params"][k]).max() <= 1e-5 * max(np.abs(b["params"][k]).max(), 1e-30), k

    # --- the per-Gaussian mask traced from the 2-D masks.  cnt is an integer per Gaussian; it may differ only where a pixel's
    # alpha / transmittance sits within a rounding of a threshold under libm's exp (reference) vs gsr_expf (product) -- the
    # flips tests/test_gpu_reference.py::test_apply_weights_vs_reference counts pixel by pixel: a handful of Gaussians, by a
    # few counts each over the 12 views.  The SELECTED mask (ratio > 0.5) must be identical: the whole loop hangs on it.
    dc = (a["trace"][2].long() - b["trace"][2].long()).abs().view(-1)
    print(f"apply_weights over {TRACE_VIEWS} views: cnt differs on {int((dc != 0).sum())} of {dc.numel()} Gaussians "
          f"(max {int(dc.max())}); mask: {int(a['trace'][0].sum())} selected")
    assert int((dc != 0).sum()) <= 1e-3 * dc.numel() + 2 and int(dc.max()) <= 4 * TRACE_VIEWS
    assert torch.equal(a["trace"][0], b["trace"][0]), "apply_weights: selected mask differs"
    dw = (a["trace"][1] - b["trace"][1]).abs().view(-1)
    assert float(dw[dc == 0].max()) <= 1e-4 * max(1.0, float(a["trace"][1].abs().max()))
    # --- the thresholded semantic maps of the first step's views
    for sa, sb in zip(a["semantic"], b["semantic"]):
        assert int((sa != sb).sum()) <= 2  # a pixel whose |colour| sits within rounding of 0.8
    # --- P at every step, and what every densification did
    print("P per step:", b["P"])
    print("densify (clone, split, prune):", b["densify"])
    assert a["P"] == b["P"]
    assert a["densify"] == b["densify"] and len(b["densify"]) == (STEPS - 1) // DENSIFY_EVERY
    assert sum(d[0] + d[1] for d in b["densify"]) > 0 and sum(d[2] for d in b["densify"]) > 0, "the loop must densify AND prune"
    assert torch.equal(a["mask"], b["mask"])
    # --- losses and the parameter trajectories after STEPS steps
    la, lb = np.array(a["loss"]), np.array(b["loss"])
    print("loss first / last:", la[0], la[-1], "max rel diff", float(np.abs(la - lb).max() / np.abs(la).max()))
    assert np.abs(la - lb).max() <= 1e-5 * np.abs(la).max()
    for k in NAMES:
        x, y = a["params"][k], b["params"][k]
        rel = float(np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30))
        mx = float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-30))
        print(f"  {k}: rel-L2 {rel:.2e}  max/|max| {mx:.2e}")
        # Adam with eps = 1e-15 normalises every gradient component: an element whose gradient is a cancelling sum at the
        # rounding level moves by +-lr per step whatever its size, so the bar is on the tensor (rel-L2) and a looser one on
        # the worst element
        assert rel <= 1e-4, (k, rel)
        assert mx <= 5e-3, (k, mx)
