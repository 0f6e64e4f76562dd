#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by IMPORTING THE REFERENCE'S OWN PYTHON
modules (only possible where /root/reference exists; the fixtures themselves are committed).

    sh_eval.npz   reference `eval_sh` (gaussiansplatting/utils/sh_utils.py:57-112) on seeded inputs,
                  degrees 0..3  -> pins the oracle's SH -> RGB step (forward.cu:20-71 is the same maths)
    cameras.npz   reference `getWorld2View2` / `getProjectionMatrix` (utils/graphics_utils.py:40-87) and the
                  Simple_Camera composition (scene/cameras.py:92-95) -> pins the host-side matrix conventions
    near_points.npz  reference `K_nearest_neighbors` (gaussiansplatting/knn.py, scipy KDTree) and the reference's own
                  `GaussianModel.get_near_gaussians_by_mask` (scene/gaussian_model.py:865-898; the method's source is
                  compiled out of the reference file and run on a stub holding `_xyz`, so none of the module's GPU-only
                  imports are needed) -> pins the oracle's near_points / get_near_gaussians_by_mask
"""
import ast
import importlib.util
import sys as _sys

_sys.dont_write_bytecode = True  # the reference tree is read-only input: no __pycache__ there
import math
import os

import numpy as np
import torch

REF = "/root/reference/gaussiansplatting"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sh_utils = load(os.path.join(REF, "utils", "sh_utils.py"), "ref_sh_utils")
    gfx = load(os.path.join(REF, "utils", "graphics_utils.py"), "ref_graphics_utils")

    g = torch.Generator().manual_seed(2024)
    P = 512
    shs = torch.randn(P, 16, 3, generator=g) * 0.4          # (P, M, 3) as the rasterizer receives them
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    out = {"shs": shs.numpy(), "dirs": dirs.numpy()}
    for deg in range(4):
        # reference layout for eval_sh is (P, 3, M): gaussian_renderer/__init__.py:112-114
        out[f"rgb_deg{deg}"] = sh_utils.eval_sh(deg, shs.transpose(1, 2), dirs).numpy()
    np.savez(os.path.join(HERE, "sh_eval.npz"), **out)

    cams = {}
    rng = np.random.default_rng(7)
    for i in range(4):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        # world-to-camera rotation; the reference stores R = its transpose (dataset_readers.py:85)
        Rw2c = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        R = Rw2c.T
        T = rng.normal(size=3) * 2.0
        fovx, fovy = math.radians(40 + 10 * i), math.radians(30 + 7 * i)
        wv = torch.tensor(gfx.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = gfx.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wv.inverse()[3, :3]
        cams[f"R{i}"], cams[f"T{i}"] = R, T
        cams[f"fov{i}"] = np.array([fovx, fovy])
        cams[f"world_view{i}"], cams[f"proj{i}"] = wv.numpy(), proj.numpy()
        cams[f"full_proj{i}"], cams[f"center{i}"] = full.numpy(), center.numpy()
    np.savez(os.path.join(HERE, "cameras.npz"), **cams)
    print("wrote", os.path.join(HERE, "sh_eval.npz"), os.path.join(HERE, "cameras.npz"))
    near_points_fixture()


def reference_method(path, name, globs):
    """The function `name` of the class body in `path`, compiled from the reference's source as it lies there."""
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), globs)
            return globs[name]
    raise KeyError(name)


def near_points_fixture():
    knn = load(os.path.join(REF, "knn.py"), "ref_knn")
    fn = reference_method(os.path.join(REF, "scene", "gaussian_model.py"), "get_near_gaussians_by_mask",
                          {"torch": torch, "K_nearest_neighbors": knn.K_nearest_neighbors})

    class Stub:
        pass

    g = torch.Generator().manual_seed(99)
    N = 6000
    centres = torch.randn(12, 3, generator=g) * 1.5
    xyz = centres[torch.randint(0, 12, (N,), generator=g)] + torch.randn(N, 3, generator=g) * 0.35
    # the "object": three of the clusters' neighbourhoods, ragged
    mask = ((xyz - centres[0]).norm(dim=1) < 0.45) | ((xyz - centres[5]).norm(dim=1) < 0.3) | \
           ((xyz - centres[9]).norm(dim=1) < 0.5)
    xyz[7] = xyz[mask.nonzero()[0, 0]]  # a duplicate of an object point among the remaining ones: distance 0
    mask[7] = False
    out = {"xyz": xyz.numpy(), "mask": mask.numpy()}
    stub = Stub()
    stub._xyz = xyz
    for i, th in enumerate((0.1, 0.2, 0.05)):
        out[f"thresh{i}"] = np.float64(th)
        out[f"near{i}"] = fn(stub, mask.unsqueeze(1), dist_thresh=th).numpy()
    ref, qry = xyz[mask], xyz[~mask]
    _, idx, dist = knn.K_nearest_neighbors(ref, 1, query=qry, return_dist=True)
    out["nn_dist"], out["nn_idx"] = dist.numpy(), idx.numpy()
    np.savez_compressed(os.path.join(HERE, "near_points.npz"), **out)
    print("wrote", os.path.join(HERE, "near_points.npz"), {k: int(out[k].sum()) for k in ("mask", "near0", "near1", "near2")})


if __name__ == "__main__":
    main()
