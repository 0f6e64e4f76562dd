"""TEST INFRASTRUCTURE: a CPU stand-in for `diff_gaussian_rasterization._C` built on the oracle.

It exists so that the host-side logic of the drop-in package (the L1 autograd wrapper, the
`render()` mirror, the gradient bucket, the multi-view all-reduce) can be exercised on a machine
without a GPU, and it is BASELINE config 1 ("10k random Gaussians, 256x256, CPU").  It is installed
by monkeypatching inside tests only; the product never imports it.
"""
import itertools

import numpy as np
import torch

from gaussianeditor_amd.diff_gaussian_rasterization import _C as real_C
from oracle import cpu as O

_registry = {}
_ids = itertools.count(1)


def _opt(t):
    return None if t is None or t.numel() == 0 else t.detach().cpu()


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, flags=None):
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    H, W = int(image_height), int(image_width)
    f = O.forward(means3D, _opt(scales), _opt(rotations), opacity, _opt(sh), _opt(colors), _opt(cov3D_precomp),
                  viewmatrix, projmatrix, campos, background, W, H, tan_fovx, tan_fovy, scale_modifier, degree, prefiltered)
    key = next(_ids)
    _registry[key] = f
    # inputs of the view, for rasterize_gaussians_aux (defined as: the image a second full render would produce)
    f["_view_args"] = (means3D, _opt(scales), _opt(rotations), opacity, _opt(cov3D_precomp), viewmatrix, projmatrix, campos,
                       W, H, tan_fovx, tan_fovy, scale_modifier, degree, prefiltered)
    tag = torch.tensor([key], dtype=torch.int64).view(torch.uint8).clone()
    z = torch.zeros(0, dtype=torch.uint8)
    return (int(f["num_rendered"]), torch.from_numpy(f["color"]), torch.from_numpy(f["depth"]),
            torch.from_numpy(f["radii"]), tag, z, z)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug, flags=None, grad_allocator=None):
    f = _registry[int(geomBuffer.view(torch.int64)[0])]
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    g = O.backward(f, dL_dout_color, means3D, _opt(scales), _opt(rotations), _opt(sh), _opt(colors), _opt(cov3D_precomp),
                   viewmatrix, projmatrix, campos, background, W, H, tan_fovx, tan_fovy, scale_modifier, degree)
    P = means3D.size(0)
    M = sh.size(1) if sh.size(0) != 0 else 0
    if grad_allocator is not None:
        grad_allocator("acc_rows", (16 * P,), False)  # (the binding's first request of a backward)
    if M != 0 and grad_allocator is not None and grad_allocator("sh_rgb", (P, 3), False) is not None:
        # the notification between K7 and K8+K9: K7's row mask (uint8 (P,): the Gaussians whose accumulator rows -- dL_dmeans2D,
        # dL_dopacity, dL_dconic, dL_dcolors; include/gsr.h GSR_ACC_* -- it adds to)
        rows = np.concatenate([np.ascontiguousarray(g[k]).reshape(P, -1) for k in ("dL_dmeans2D", "dL_dopacity", "dL_dconic", "dL_dcolors")],
                              axis=1)
        grad_allocator("after_blend_backward", torch.from_numpy((rows != 0).any(axis=1).astype(np.uint8)), False)
    out = []
    for name, key, shape in (("means2D", "dL_dmeans2D", (P, 3)), ("colors_precomp", "dL_dcolors", (P, 3)),
                             ("opacities", "dL_dopacity", (P, 1)), ("means3D", "dL_dmeans3D", (P, 3)),
                             ("cov3Ds_precomp", "dL_dcov3D", (P, 6)), ("sh", "dL_dsh", (P, M, 3)),
                             ("scales", "dL_dscales", (P, 3)), ("rotations", "dL_drotations", (P, 4))):
        if name == "sh" and M != 0 and grad_allocator is not None:
            rgb = grad_allocator("sh_rgb", (P, 3), False)
            if rgb is not None:  # "rgb" exchange mode: the clamp-masked colour gradient instead of dL_dsh
                vis = (f["radii"] > 0)[:, None]
                masked = np.where(np.logical_and(vis, f["clamped"] == 0), g["dL_dcolors"].reshape(P, 3), 0.0)
                rgb.copy_(torch.from_numpy(np.ascontiguousarray(masked, dtype=np.float32)))
                out.append(None)
                continue
        t = real_C._alloc(grad_allocator, name, shape, False, means3D.device)  # honours the gradient-bucket allocator
        t.copy_(torch.from_numpy(np.ascontiguousarray(g[key])).reshape(shape))
        out.append(t)
    return tuple(out)


def sh_grad_compose(means3D, campos_all, rgb_all, degree, M):
    return torch.from_numpy(O.sh_grad_compose(means3D.detach().cpu().numpy(), campos_all.cpu().numpy(), rgb_all.cpu().numpy(),
                                              int(degree), int(M)))


# --- view messages of the touched-rows exchange (include/gsr.h), same word layout as the library's -----------------------
def view_message_words(P, cap):
    return 4 + (int(P) + 1023) // 1024 + 18 * int(cap)


def view_message_plan(grads5, rgb, readback=True):
    P = int(rgb.size(0))
    rows = torch.cat([t.reshape(P, -1) for t in grads5] + [rgb.reshape(P, 3)], dim=1)
    idx = (rows != 0).any(dim=1).nonzero().view(-1)
    return idx, (int(idx.numel()) if readback else torch.tensor([idx.numel()], dtype=torch.int64))


def view_message_plan_blend(touched):
    idx = (touched != 0).nonzero().view(-1)
    return idx, torch.tensor([idx.numel()], dtype=torch.int64)


def view_message_pack(plan, grads5, rgb, campos, cap, message):
    P, n, nb = int(rgb.size(0)), int(plan.numel()), (int(rgb.size(0)) + 1023) // 1024
    w = message.view(torch.int32)
    message[0:3] = campos.reshape(3)
    w[3] = n  # the true count, even when only the first `cap` rows fit (the sender then sends the message again)
    w[4:4 + nb] = torch.searchsorted(plan, torch.arange(nb, dtype=plan.dtype) * 1024).to(torch.int32)
    off = 4 + nb
    m = min(n, int(cap))
    w[off:off + m] = plan[:m].to(torch.int32)
    off += cap
    for t, k in zip(list(grads5) + [rgb], (3, 3, 4, 3, 1, 3)):
        message[off:off + k * cap].view(cap, k)[:m] = t.reshape(P, k)[plan[:m]]
        off += k * cap


def view_messages_accumulate(messages, P, cap, degree, M, means3D, dense, row_valid=None):
    nb = (int(P) + 1023) // 1024
    stale = [None if d is None else d.clone() for d in dense] if row_valid is not None else None
    for d in dense:
        if d is not None:
            d.zero_()
    if row_valid is not None:
        row_valid.zero_()
        for v in range(messages.size(0)):
            msg = messages[v]
            n = int(msg.view(torch.int32)[3])
            row_valid[msg.view(torch.int32)[4 + nb:4 + nb + n].to(torch.int64)] = 1
    for v in range(messages.size(0)):  # ascending view order
        msg = messages[v]
        n = int(msg.view(torch.int32)[3])
        if n == 0:
            continue
        off = 4 + nb
        i = msg.view(torch.int32)[off:off + n].to(torch.int64)
        off += cap
        rows = []
        for k in (3, 3, 4, 3, 1, 3):
            rows.append(msg[off:off + k * cap].view(cap, k)[:n])
            off += k * cap
        for k in range(5):
            dense[k].reshape(int(P), -1)[i] += rows[k]
        if dense[5] is not None:
            # one view's SH gradient of the packed rows, rebuilt exactly as gsr_sh_grad_compose does for N = 1
            t = O.sh_grad_compose(means3D.detach()[i].numpy(), msg[0:3].reshape(1, 3).numpy(),
                                  rows[5].reshape(1, n, 3).numpy(), int(degree), int(M))
            dense[5][i] += torch.from_numpy(t)
    if row_valid is not None:  # rows no view sent keep what they held (the kernel does not write them)
        inv = ~row_valid.bool()
        for d, o in zip(dense, stale):
            if d is not None:
                d[inv] = o[inv]


def rasterize_gaussians_aux(background, colors, num_rendered, geomBuffer, binningBuffer, imgBuffer, image_height,
                            image_width, debug=False, flags=None):
    f = _registry[int(geomBuffer.view(torch.int64)[0])]
    (means3D, scales, rotations, opacity, cov3D, viewmatrix, projmatrix, campos, W, H, tfx, tfy, smod, degree,
     prefiltered) = f["_view_args"]
    g = O.forward(means3D, scales, rotations, opacity, None, colors.detach().cpu(), cov3D, viewmatrix, projmatrix, campos,
                  background, W, H, tfx, tfy, smod, degree, prefiltered)
    return torch.from_numpy(g["color"])


def mark_visible(means3D, viewmatrix, projmatrix):
    return torch.from_numpy(O.mark_visible(means3D, viewmatrix, projmatrix))


def apply_weights(background, means3D, weights, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                  projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                  image_weights, cnt, debug, flags=None):
    w = np.ascontiguousarray(weights.numpy())
    c = np.ascontiguousarray(cnt.numpy().reshape(-1))
    O.apply_weights(means3D, _opt(scales), _opt(rotations), opacity, _opt(cov3D_precomp), viewmatrix, projmatrix, campos,
                    int(image_width), int(image_height), tan_fovx, tan_fovy, image_weights, w, c, scale_modifier)
    weights.copy_(torch.from_numpy(w))
    cnt.copy_(torch.from_numpy(c).reshape(cnt.shape))


def install(monkeypatch):
    """Route the drop-in's native calls to the oracle for the duration of one test."""
    import gaussianeditor_amd.diff_gaussian_rasterization as dgr

    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "rasterize_gaussians_aux", "sh_grad_compose",
                 "view_message_words", "view_message_plan", "view_message_plan_blend", "view_message_pack", "view_messages_accumulate", "mark_visible", "apply_weights"):
        monkeypatch.setattr(dgr._C, name, globals()[name])
