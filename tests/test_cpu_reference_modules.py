"""CPU, only where /root/reference exists (this container; skipped on the GPU box): the REFERENCE'S OWN Python modules
imported unmodified after `gaussianeditor_amd.install()` and executed against the drop-in.

  * gaussiansplatting/gaussian_renderer/__init__.py (`render` :45-150, `camera2rasterizer` :21-42) runs on the drop-in's L1
    API; its results equal the mirror gaussianeditor_amd/gaussian_renderer.py and the oracle (the L2
    boundary itself, not a restatement of it);
  * gaussiansplatting/scene/gaussian_model.py imports (it needs `simple_knn._C` and `plyfile` at import time: the shims),
    and its optimizer surgery -- `cat_tensors_to_optimizer` :609-641, `_prune_optimizer` :568-591 -- run as written pins
    the numpy restatements (`oracle.cpu.append_rows` / `compact_rows`) that the GPU tests check the HIP kernels against.

The native library is replaced by the oracle stand-in (tests/oracle_backend.py); `torch.zeros_like(..., device="cuda")` in
the reference's render() is redirected to the CPU because this container has no GPU.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle_backend
from helpers import make_case, oracle_backward, oracle_forward, rel_err, seed_gradient

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussiansplatting")),
                                reason="the reference checkout is only present in the development container")


@pytest.fixture()
def reference_modules(monkeypatch):
    import gaussianeditor_amd

    gaussianeditor_amd.install()
    monkeypatch.syspath_prepend(REF)
    import gaussiansplatting.gaussian_renderer as ref_renderer  # the reference's file, unmodified
    from gaussiansplatting.scene.gaussian_model import GaussianModel

    real = torch.zeros_like

    def zeros_like_on_cpu(t, *a, **kw):
        kw.pop("device", None)  # the reference hard-codes device="cuda" (gaussian_renderer/__init__.py:62)
        return real(t, *a, **kw)

    monkeypatch.setattr(torch, "zeros_like", zeros_like_on_cpu)
    return SimpleNamespace(renderer=ref_renderer, GaussianModel=GaussianModel)


class _PC:
    """Duck-typed GaussianModel getters (scene/gaussian_model.py:221-258)."""

    def __init__(self, sc):
        self._sc = {k: v.clone().requires_grad_(True) for k, v in sc.items() if isinstance(v, torch.Tensor) and k != "bg"}
        self.active_sh_degree = 3
        self.max_sh_degree = 3

    get_xyz = property(lambda s: s._sc["xyz"])
    get_opacity = property(lambda s: s._sc["opacity"])
    get_scaling = property(lambda s: s._sc["scaling"])
    get_rotation = property(lambda s: s._sc["rotation"])
    get_features = property(lambda s: s._sc["features"])


def test_reference_render_runs_on_the_drop_in(oracle, monkeypatch, reference_modules):
    from gaussianeditor_amd import gaussian_renderer as mirror

    oracle_backend.install(monkeypatch)
    ref = reference_modules.renderer
    assert ref.GaussianRasterizer is sys.modules["gaussianeditor_amd.diff_gaussian_rasterization"].GaussianRasterizer
    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    case = make_case(3000, 96, 64, seed=3, s0=0.05)
    f = oracle_forward(oracle, case)
    G = seed_gradient(64, 96, 2) * 64 * 96
    g = oracle_backward(oracle, case, f, G)
    outs = {}
    for name, fn in (("reference", ref.render), ("mirror", mirror.render)):
        pc = _PC(case["sc"])
        out = fn(case["cam"], pc, pipe, case["bg"])
        (out["render"] * G).sum().backward()
        outs[name] = (out, pc)
    (a, pa), (b, pb) = outs["reference"], outs["mirror"]
    assert set(a) == set(b) == {"render", "viewspace_points", "visibility_filter", "radii", "depth_3dgs"}
    for k in ("render", "depth_3dgs", "radii", "visibility_filter"):
        assert torch.equal(a[k], b[k]), k
    assert np.array_equal(a["render"].detach().numpy(), f["color"]) and np.array_equal(a["radii"].numpy(), f["radii"])
    for getter, key in (("get_xyz", "dL_dmeans3D"), ("get_features", "dL_dsh"), ("get_opacity", "dL_dopacity"),
                        ("get_scaling", "dL_dscales"), ("get_rotation", "dL_drotations")):
        ga, gb = getattr(pa, getter).grad, getattr(pb, getter).grad
        assert torch.equal(ga, gb), getter
        assert rel_err(ga.numpy(), g[key].reshape(ga.shape)) < 1e-6, getter
    assert torch.equal(a["viewspace_points"].grad, b["viewspace_points"].grad)
    assert rel_err(a["viewspace_points"].grad.numpy(), g["dL_dmeans2D"]) < 1e-6
    # override_color and SH evaluated in Python, through the reference's own code
    pc = _PC(case["sc"])
    mask = (torch.rand(3000, 1, generator=torch.Generator().manual_seed(1)) > 0.5).float().repeat(1, 3)
    c = ref.render(case["cam"], pc, pipe, case["bg"], override_color=mask)["render"]
    assert np.array_equal(c.detach().numpy(), oracle_forward(oracle, case, colors_precomp=mask)["color"])
    # (pipe.convert_SHs_python=True cannot be exercised through the reference: its render() dereferences `shs = None` at
    #  gaussian_renderer/__init__.py:125; the mirror keeps that path working and is tested on its own)
    # camera2rasterizer + apply_weights as GaussianModel.apply_weights calls them (scene/gaussian_model.py:817-832)
    w, cnt = torch.zeros(3000, 1), torch.zeros(3000, 1, dtype=torch.int32)
    m = (torch.rand(1, 64, 96, generator=torch.Generator().manual_seed(2)) > 0.3).float()
    sc = case["sc"]
    ref.camera2rasterizer(case["cam"], torch.zeros(3)).apply_weights(sc["xyz"], None, sc["opacity"], None, w, sc["scaling"],
                                                                      sc["rotation"], None, cnt, m)
    w2, c2 = np.zeros((3000, 1), np.float32), np.zeros(3000, np.int32)
    cam = case["cam"]
    oracle.apply_weights(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], None, cam.world_view_transform,
                         cam.full_proj_transform, cam.camera_center, 96, 64, case["tfx"], case["tfy"], m, w2, c2)
    assert np.array_equal(cnt.numpy().reshape(-1), c2) and np.array_equal(w.numpy(), w2)


def test_reference_point_cloud_render_equals_the_mirror(oracle, monkeypatch, reference_modules):
    """`point_cloud_render` (gaussian_renderer/__init__.py:156-250, imported by webui.py:39): the reference's own function on
    the drop-in against the mirror's, outputs and the gradient of the screen-space points bit for bit."""
    from gaussianeditor_amd import gaussian_renderer as mirror

    oracle_backend.install(monkeypatch)
    case = make_case(2000, 80, 48, seed=5, s0=0.05)
    G = seed_gradient(48, 80, 1) * 48 * 80
    outs = []
    for fn in (reference_modules.renderer.point_cloud_render, mirror.point_cloud_render):
        xyz = case["sc"]["xyz"].clone().requires_grad_(True)
        out = fn(case["cam"], xyz, None, case["bg"])
        (out["render"] * G).sum().backward()
        outs.append((out, xyz))
    (a, xa), (b, xb) = outs
    assert set(a) == set(b) == {"render", "viewspace_points", "visibility_filter", "radii", "depth_3dgs"}
    for k in ("render", "depth_3dgs", "radii", "visibility_filter"):
        assert torch.equal(a[k], b[k]), k
    assert int(a["visibility_filter"].sum()) > 100 and float(a["render"].max()) > 0.5
    assert torch.equal(xa.grad, xb.grad) and torch.equal(a["viewspace_points"].grad, b["viewspace_points"].grad)


def _adam_with_state(P, seed):
    gen = torch.Generator().manual_seed(seed)
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    params = {k: torch.nn.Parameter(torch.randn(s, generator=gen)) for k, s in shapes.items()}
    opt = torch.optim.Adam([{"params": [p], "lr": 1e-3, "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
    for p in params.values():
        p.grad = torch.randn(p.shape, generator=gen)
    opt.step()
    return opt, gen


def test_reference_optimizer_surgery_pins_the_restatements(oracle, reference_modules):
    """GaussianModel.cat_tensors_to_optimizer / _prune_optimizer, the reference's methods as written (called on a bare
    object that only carries `optimizer`), against oracle.cpu.append_rows / compact_rows."""
    GM = reference_modules.GaussianModel
    P = 257
    # --- densify
    opt, gen = _adam_with_state(P, 5)
    before = {g["name"]: (g["params"][0].detach().clone(), opt.state[g["params"][0]]["exp_avg"].clone(),
                          opt.state[g["params"][0]]["exp_avg_sq"].clone()) for g in opt.param_groups}
    ext = {g["name"]: torch.randn((19,) + tuple(g["params"][0].shape[1:]), generator=gen) for g in opt.param_groups}
    new = GM.cat_tensors_to_optimizer(SimpleNamespace(optimizer=opt), ext)
    for g in opt.param_groups:
        k, p = g["name"], g["params"][0]
        assert new[k] is p and p.shape[0] == P + 19
        want_p, want_m, want_v = oracle.append_rows([t.numpy() for t in before[k]], [ext[k].numpy(), None, None], 19)
        assert np.array_equal(p.detach().numpy(), want_p)
        assert np.array_equal(opt.state[p]["exp_avg"].numpy(), want_m)
        assert np.array_equal(opt.state[p]["exp_avg_sq"].numpy(), want_v)
    # --- prune
    opt, gen = _adam_with_state(P, 6)
    before = {g["name"]: (g["params"][0].detach().clone(), opt.state[g["params"][0]]["exp_avg"].clone(),
                          opt.state[g["params"][0]]["exp_avg_sq"].clone()) for g in opt.param_groups}
    keep = torch.rand(P, generator=gen) > 0.3
    new = GM._prune_optimizer(SimpleNamespace(optimizer=opt), keep)
    for g in opt.param_groups:
        k, p = g["name"], g["params"][0]
        want = oracle.compact_rows([t.numpy() for t in before[k]], keep.numpy())
        assert new[k] is p and np.array_equal(p.detach().numpy(), want[0])
        assert np.array_equal(opt.state[p]["exp_avg"].numpy(), want[1])
        assert np.array_equal(opt.state[p]["exp_avg_sq"].numpy(), want[2])


def test_reference_model_loads_and_saves_through_the_shims(reference_modules, tmp_path):
    """GaussianModel.save_ply (scene/gaussian_model.py:396-445), the reference's own writer, through whichever `plyfile`
    install() registered (compat/plyfile.py here); read back with the same package.  (__init__ and load_ply put tensors on
    "cuda" and cannot run in this container: the object is created bare.)"""
    GM = reference_modules.GaussianModel
    gm = GM.__new__(GM)
    P = 50
    gen = torch.Generator().manual_seed(4)
    gm._xyz = torch.randn(P, 3, generator=gen)
    gm._features_dc = torch.randn(P, 1, 3, generator=gen)
    gm._features_rest = torch.randn(P, 15, 3, generator=gen)
    gm._opacity = torch.randn(P, 1, generator=gen)
    gm._scaling = torch.randn(P, 3, generator=gen)
    gm._rotation = torch.randn(P, 4, generator=gen)
    path = str(tmp_path / "pc" / "point_cloud.ply")
    gm.save_ply(path)
    from plyfile import PlyData  # whichever `plyfile` install() left in sys.modules

    el = PlyData.read(path).elements[0]
    assert len(el["x"]) == P and np.allclose(np.asarray(el["x"]), gm._xyz[:, 0].numpy())
    assert np.allclose(np.asarray(el["f_rest_44"]), gm._features_rest.transpose(1, 2).flatten(start_dim=1)[:, 44].numpy())
    assert np.allclose(np.asarray(el["rot_3"]), gm._rotation[:, 3].numpy())
    # the one-command asset path (gaussianeditor_amd/scene_ply.py, bench.py --ply): what the reference's writer wrote comes
    # back, tensor for tensor, in the shapes load_ply (:455-533) produces; and a file written by scene_ply is byte-identical
    from gaussianeditor_amd import scene_ply

    back = scene_ply.load_gaussians_ply(path)
    for k, t in (("xyz", gm._xyz), ("f_dc", gm._features_dc), ("f_rest", gm._features_rest), ("opacity", gm._opacity),
                 ("scaling", gm._scaling), ("rotation", gm._rotation)):
        assert torch.equal(back[k], t), k
    assert back["max_sh_degree"] == 3
    mine = str(tmp_path / "pc" / "mine.ply")
    scene_ply.save_gaussians_ply(mine, gm._xyz, gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._rotation)
    assert open(mine, "rb").read() == open(path, "rb").read()
