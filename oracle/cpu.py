"""ctypes front-end of the CPU oracle (oracle/gsr_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- see the header of gsr_oracle.cpp.  Nothing under
gaussianeditor_amd/ imports this module.

All functions take/return numpy arrays (float32 / int32 / uint32 / uint64 / uint8)
and expose *every* intermediate of the reference pipeline
(DGR/cuda_rasterizer/rasterizer_impl.cu:179-341) so the HIP path can be compared
stage by stage.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsr_oracle.so")
_lib = None

c_f = ctypes.c_float
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (a few seconds)."""
    src = os.path.join(_HERE, "gsr_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgsr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.gsro_expf.restype = c_f
        _lib.gsro_expf.argtypes = [c_f]
        _lib.gsro_scan.restype = c_i64
        _lib.gsro_blend_forward.restype = c_i64
        _lib.gsro_sort_bits.restype = ctypes.c_uint32
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return c_vp(0)
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return c_vp(a.ctypes.data)


def _f32(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if a.size else None


def num_threads() -> int:
    return int(lib().gsro_num_threads())


def expf(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    f = lib().gsro_expf
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = f(float(flat_in[i]))
    return out


def sort_bits(W: int, H: int) -> int:
    return int(lib().gsro_sort_bits(c_i(W), c_i(H)))


def preprocess(means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp, viewmatrix, projmatrix,
               campos, W, H, tanfovx, tanfovy, scale_modifier=1.0, sh_degree=0, prefiltered=False) -> Dict[str, np.ndarray]:
    means3D = _f32(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    scales, rotations, opacities = _f32(scales), _f32(rotations), _f32(opacities)
    shs, cov3D_precomp, colors_precomp = _f32(shs), _f32(cov3D_precomp), _f32(colors_precomp)
    viewmatrix, projmatrix, campos = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    M = 0 if shs is None else shs.shape[1]
    out = dict(
        radii=np.zeros(P, np.int32),
        means2D=np.zeros((P, 2), np.float32),
        depths=np.zeros(P, np.float32),
        cov3D=np.zeros((P, 6), np.float32),
        rgb=np.zeros((P, 3), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32),
        tiles_touched=np.zeros(P, np.uint32),
        clamped=np.zeros((P, 3), np.uint8),
    )
    if P == 0:
        out["status"] = 0
        return out
    st = lib().gsro_preprocess(
        c_i(P), c_i(sh_degree), c_i(M), _p(means3D), _p(scales), c_f(scale_modifier), _p(rotations), _p(opacities),
        _p(shs), _p(cov3D_precomp), _p(colors_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), c_i(W), c_i(H),
        c_f(tanfovx), c_f(tanfovy), c_i(int(prefiltered)), _p(out["radii"]), _p(out["means2D"]), _p(out["depths"]),
        _p(out["cov3D"]), _p(out["rgb"]), _p(out["conic_opacity"]), _p(out["tiles_touched"]), _p(out["clamped"]))
    out["status"] = int(st)
    if cov3D_precomp is not None:
        out["cov3D"] = cov3D_precomp.reshape(P, 6)
    return out


def bin_tiles(geom: Dict[str, np.ndarray], W: int, H: int) -> Dict[str, np.ndarray]:
    P = geom["radii"].shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    offsets = np.zeros(P, np.uint32)
    R = int(lib().gsro_scan(c_i(P), _p(geom["tiles_touched"]), _p(offsets))) if P else 0
    out = dict(
        point_offsets=offsets,
        num_rendered=R,
        keys_unsorted=np.zeros(R, np.uint64),
        values_unsorted=np.zeros(R, np.uint32),
        keys=np.zeros(R, np.uint64),
        point_list=np.zeros(R, np.uint32),
        ranges=np.zeros((gx * gy, 2), np.uint32),
    )
    if P:
        lib().gsro_bin(c_i(P), c_i64(R), c_i(W), c_i(H), _p(geom["means2D"]), _p(geom["depths"]), _p(offsets),
                       _p(geom["radii"]), _p(out["keys_unsorted"]), _p(out["values_unsorted"]), _p(out["keys"]),
                       _p(out["point_list"]), _p(out["ranges"]))
    return out


def blend_forward(geom, binning, colors, bg, W, H) -> Dict[str, np.ndarray]:
    colors, bg = _f32(colors), _f32(bg)
    out = dict(
        final_T=np.zeros(H * W, np.float32),
        n_contrib=np.zeros(H * W, np.uint32),
        color=np.zeros((3, H, W), np.float32),
        depth=np.zeros((1, H, W), np.float32),
    )
    ev = lib().gsro_blend_forward(c_i(W), c_i(H), _p(binning["ranges"]), _p(binning["point_list"]), _p(geom["means2D"]),
                                  _p(colors), _p(geom["depths"]), _p(geom["conic_opacity"]), _p(bg), _p(out["final_T"]),
                                  _p(out["n_contrib"]), _p(out["color"]), _p(out["depth"]))
    out["pixel_instances"] = int(ev)
    return out


def forward(means3D, scales, rotations, opacities, shs, colors_precomp, cov3D_precomp, viewmatrix, projmatrix, campos,
            bg, W, H, tanfovx, tanfovy, scale_modifier=1.0, sh_degree=0, prefiltered=False) -> Dict[str, np.ndarray]:
    """Whole forward (Rasterizer::forward, rasterizer_impl.cu:179-285) with all intermediates."""
    geom = preprocess(means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp, viewmatrix,
                      projmatrix, campos, W, H, tanfovx, tanfovy, scale_modifier, sh_degree, prefiltered)
    binning = bin_tiles(geom, W, H)
    cp = _f32(colors_precomp)
    colors = cp if cp is not None else geom["rgb"]
    img = blend_forward(geom, binning, colors, bg, W, H)
    res = {}
    res.update(geom)
    res.update(binning)
    res.update(img)
    res["colors_used"] = colors
    return res


def backward(fwd: Dict[str, np.ndarray], dL_dpix, means3D, scales, rotations, shs, colors_precomp, cov3D_precomp,
             viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, scale_modifier=1.0, sh_degree=0):
    """Rasterizer::backward (rasterizer_impl.cu:289-341): K7 then K8+K9.
    Returns the 8 tensors of _C.rasterize_gaussians_backward plus dL_dconic."""
    means3D = _f32(means3D)
    P = means3D.shape[0]
    scales, rotations, shs = _f32(scales), _f32(rotations), _f32(shs)
    cp, c3 = _f32(colors_precomp), _f32(cov3D_precomp)
    viewmatrix, projmatrix, campos, bg = _f32(viewmatrix), _f32(projmatrix), _f32(campos), _f32(bg)
    dL_dpix = _f32(dL_dpix)
    M = 0 if shs is None else shs.shape[1]
    colors = cp if cp is not None else fwd["rgb"]
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32),
        dL_dconic=np.zeros((P, 4), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32),
        dL_dcolors=np.zeros((P, 3), np.float32),
        dL_dmeans3D=np.zeros((P, 3), np.float32),
        dL_dcov3D=np.zeros((P, 6), np.float32),
        dL_dsh=np.zeros((P, M, 3), np.float32),
        dL_dscales=np.zeros((P, 3), np.float32),
        dL_drotations=np.zeros((P, 4), np.float32),
    )
    lib().gsro_blend_backward(c_i(P), c_i(W), c_i(H), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(bg),
                              _p(fwd["means2D"]), _p(fwd["conic_opacity"]), _p(colors), _p(fwd["final_T"]),
                              _p(fwd["n_contrib"]), _p(dL_dpix), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]),
                              _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    cov3D = c3 if c3 is not None else fwd["cov3D"]
    lib().gsro_preprocess_backward(
        c_i(P), c_i(sh_degree), c_i(M), _p(means3D), _p(fwd["radii"]), _p(shs), _p(fwd["clamped"]), _p(scales),
        _p(rotations), c_f(scale_modifier), _p(np.ascontiguousarray(cov3D)), _p(viewmatrix), _p(projmatrix), c_i(W),
        c_i(H), c_f(tanfovx), c_f(tanfovy), _p(campos), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dcolors"]),
        _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    means3D = _f32(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    out = np.zeros(P, np.uint8)
    if P:
        lib().gsro_mark_visible(c_i(P), _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(out))
    return out.astype(bool)


def apply_weights(means3D, scales, rotations, opacities, cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tanfovx,
                  tanfovy, image_weights, weights: np.ndarray, cnt: np.ndarray, scale_modifier=1.0):
    """Rasterizer::apply_weights (rasterizer_impl.cu:343-447). `weights` (P,C) float32 and
    `cnt` (P[,1]) int32 are accumulated in place."""
    image_weights = _f32(image_weights)
    C = image_weights.shape[0]
    P = weights.shape[0]
    dummy_colors = np.zeros((P, 3), np.float32)  # the reference passes `weights` in the colour slot: SH is skipped
    geom = preprocess(means3D, scales, rotations, opacities, None, cov3D_precomp, dummy_colors, viewmatrix, projmatrix,
                      campos, W, H, tanfovx, tanfovy, scale_modifier, 0, False)
    binning = bin_tiles(geom, W, H)
    assert weights.dtype == np.float32 and cnt.dtype == np.int32
    st = lib().gsro_trace_weights(c_i(W), c_i(H), c_i(C), _p(binning["ranges"]), _p(binning["point_list"]),
                                  _p(geom["means2D"]), _p(geom["conic_opacity"]), _p(image_weights), _p(weights), _p(cnt))
    if st != 0:
        raise ValueError(f"Unsupported number of channels: {C}")
    return geom, binning


def knn_mean_dist2(points) -> np.ndarray:
    """simple_knn.distCUDA2 restated (brute force, exact): (P,3) -> (P,) float32."""
    pts = _f32(points)
    P = 0 if pts is None else pts.shape[0]
    out = np.zeros(P, np.float32)
    if P:
        lib().gsro_knn_mean_dist2(c_i(P), _p(pts), _p(out))
    return out


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, row_mask=None, masked=False, anchor=None,
              anchor_scale=0.0, row_weight=None):
    """One Adam step on float32 arrays, in place on (p, m, v); rows = p.shape[0]."""
    import ctypes as C

    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    g = _f32(g)
    n = p.size
    row_len = max(1, n // max(1, p.shape[0]))
    mask = None if row_mask is None else np.ascontiguousarray(row_mask, dtype=np.uint8)
    anc = None if anchor is None else _f32(anchor)
    rw = None if row_weight is None else _f32(row_weight)
    fn = lib().gsro_adam_step
    fn.restype = None
    fn.argtypes = [C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                   C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_longlong]
    fn(n, row_len, p.ctypes.data, g.ctypes.data, m.ctypes.data, v.ctypes.data, None if anc is None else anc.ctypes.data,
       float(anchor_scale), None if rw is None else rw.ctypes.data, None if mask is None else mask.ctypes.data,
       int(bool(masked)), float(lr), float(beta1), float(beta2), float(eps), int(step))


def sh_grad_compose(means3D, campos_all, rgb_all, D, M) -> np.ndarray:
    """dL_dsh (P,M,3) of a batch of views from their camera centres (N,3) and clamp-masked colour gradients (N,P,3)."""
    import ctypes as C

    m, c, r = _f32(means3D), _f32(campos_all), _f32(rgb_all)
    N, P = r.shape[0], r.shape[1]
    out = np.zeros((P, M, 3), np.float32)
    fn = lib().gsro_sh_grad_compose
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if P:
        fn(P, int(D), int(M), N, m.ctypes.data, c.ctypes.data, r.ctypes.data, out.ctypes.data)
    return out


def compact_rows(arrays, keep):
    """Prune restated: the rows of every array where `keep` is set, in their original order (numpy boolean indexing =
    what `tensor[mask]` does in GaussianModel._prune_optimizer, gaussian_model.py:568-591)."""
    k = np.asarray(keep).astype(bool)
    return [np.ascontiguousarray(np.asarray(a)[k]) for a in arrays]


def append_rows(arrays, extensions, n=None):
    """Densification restated: np.concatenate((a, e)) per tensor, None = n zero rows (what torch.cat((t, e)) /
    torch.cat((t, zeros_like(e))) do in GaussianModel.cat_tensors_to_optimizer, gaussian_model.py:609-641)."""
    out = []
    for a, e in zip(arrays, extensions):
        a = np.asarray(a)
        if e is None:
            e = np.zeros((n,) + a.shape[1:], dtype=a.dtype)
        out.append(np.ascontiguousarray(np.concatenate((a, np.asarray(e, dtype=a.dtype)), axis=0)))
    return out


def near_points(ref, query, dist_thresh):
    """The neighbour query of GaussianModel.get_near_gaussians_by_mask restated by brute force
    (gaussiansplatting/scene/gaussian_model.py:887-891 over gaussiansplatting/knn.py): the float32 points widened to
    float64 (numpy -> KDTree), the Euclidean 1-NN distance in float64, rounded to float32 (`.to(mean)`), compared
    `<= float32(dist_thresh)` (a float32 tensor against a Python scalar compares in float32).
    (n_ref,3), (n_query,3) -> (near (n_query,) bool, nn_dist (n_query,) float32; +inf with no reference point)."""
    r = np.asarray(ref, np.float32).reshape(-1, 3).astype(np.float64)
    q = np.asarray(query, np.float32).reshape(-1, 3).astype(np.float64)
    dist = np.full(q.shape[0], np.inf, np.float64)
    if r.shape[0]:
        step = max(1, (1 << 22) // max(r.shape[0], 1))
        for i in range(0, q.shape[0], step):
            d = q[i:i + step, None, :] - r[None, :, :]
            dist[i:i + step] = np.sqrt((d * d).sum(axis=2).min(axis=1))
    d32 = dist.astype(np.float32)
    return d32 <= np.float32(dist_thresh), d32


def _quantile_f32(x, q):
    """torch.quantile(x, q) for a 1-D float32 x, interpolation 'linear' (aten/src/ATen/native/Sorting.cpp
    quantile_compute: rank = q * (n - 1) in float32, lerp between the two neighbours with torch's two-sided lerp).  A box edge
    one ulp off only matters for a point exactly on it; the fixture and the live-reference test pin this form."""
    s = np.sort(np.asarray(x, np.float32))
    rank = np.float32(q) * np.float32(s.shape[0] - 1)
    lo = np.floor(rank)
    w = np.float32(rank - lo)
    a, b = s[int(lo)], s[min(int(lo) + 1, s.shape[0] - 1)]  # ceil_(ranks) == lo + 1 unless rank is whole; then w == 0
    diff = np.float32(b - a)
    # at::native lerp, vectorised form (aten/src/ATen/native/cpu/LerpKernel.cpp lerp_vec): one fused multiply-add
    # fmadd(coeff, end - start, base) with (coeff, base) = (w, start) for w < 0.5, else (w - 1, end).  The product of two
    # float32 values is exact in float64, so float32(float64 fma) is the fused result.
    coeff, base = (w, a) if w < np.float32(0.5) else (np.float32(w - np.float32(1)), b)
    return np.float32(np.float64(coeff) * np.float64(diff) + np.float64(base))


def get_near_gaussians_by_mask(xyz, mask, dist_thresh=0.1):
    """GaussianModel.get_near_gaussians_by_mask (gaussian_model.py:865-898) restated in numpy float32."""
    xyz = np.asarray(xyz, np.float32)
    mask = np.asarray(mask).astype(bool).reshape(-1)
    obj, rem = xyz[mask], xyz[~mask]
    lo = np.array([_quantile_f32(obj[:, c], 0.03) for c in range(3)], np.float32)
    hi = np.array([_quantile_f32(obj[:, c], 0.97) for c in range(3)], np.float32)
    scale = (hi - lo).astype(np.float32)
    mid = ((hi + lo) / np.float32(2)).astype(np.float32)
    scale = (scale * np.float32(1.3)).astype(np.float32)
    lo2, hi2 = (mid - scale / np.float32(2)).astype(np.float32), (mid + scale / np.float32(2)).astype(np.float32)
    in_bbox = ((rem >= lo2) & (rem <= hi2)).all(axis=1)
    near, _ = near_points(obj, rem[in_bbox], dist_thresh)
    out = np.zeros(rem.shape[0], bool)
    out[np.nonzero(in_bbox)[0][near]] = True
    return out
