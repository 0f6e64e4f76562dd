"""CPU: host-side logic of the drop-in (L1 autograd wrapper, L2 render(), install()) exercised with
the oracle standing in for the native library (tests/oracle_backend.py).  This is BASELINE config 1:
10k random Gaussians, one 256x256 camera, on the CPU."""
import math
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle_backend
from helpers import make_case, oracle_backward, oracle_forward, rel_err, seed_gradient, settings


class _PC:
    """Duck-typed GaussianModel (scene/gaussian_model.py:221-258)."""

    def __init__(self, sc, active_sh_degree=3):
        self._sc = {k: v.clone().requires_grad_(True) for k, v in sc.items() if isinstance(v, torch.Tensor) and k != "bg"}
        self.active_sh_degree = active_sh_degree
        self.max_sh_degree = 3

    get_xyz = property(lambda s: s._sc["xyz"])
    get_opacity = property(lambda s: s._sc["opacity"])
    get_scaling = property(lambda s: s._sc["scaling"])
    get_rotation = property(lambda s: s._sc["rotation"])
    get_features = property(lambda s: s._sc["features"])


PIPE = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)


def test_render_config1_cpu(oracle, monkeypatch):
    oracle_backend.install(monkeypatch)
    from gaussianeditor_amd.gaussian_renderer import render

    case = make_case(10000, 256, 256, seed=0, s0=0.03, nviews=1, bg=(0.0, 0.0, 0.0))
    pc = _PC(case["sc"])
    out = render(case["cam"], pc, PIPE, case["bg"])
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "depth_3dgs"}
    assert out["render"].shape == (3, 256, 256) and out["depth_3dgs"].shape == (1, 256, 256)
    assert out["radii"].dtype == torch.int32 and out["visibility_filter"].dtype == torch.bool
    f = oracle_forward(oracle, case)
    assert np.array_equal(out["render"].detach().numpy(), f["color"])
    G = seed_gradient(256, 256, 0) * 256 * 256
    (out["render"] * G).sum().backward()
    g = oracle_backward(oracle, case, f, G)
    assert rel_err(pc.get_xyz.grad.numpy(), g["dL_dmeans3D"]) < 1e-6
    assert rel_err(pc.get_features.grad.numpy(), g["dL_dsh"]) < 1e-6
    assert rel_err(pc.get_opacity.grad.numpy(), g["dL_dopacity"]) < 1e-6
    # the screen-space gradient lands on the dummy tensor, as add_densification_stats expects (gaussian_model.py:811-815)
    assert rel_err(out["viewspace_points"].grad.numpy(), g["dL_dmeans2D"]) < 1e-6
    assert float(out["viewspace_points"].grad[:, 2].abs().max()) == 0.0


def test_render_override_color_and_python_sh(oracle, monkeypatch):
    oracle_backend.install(monkeypatch)
    from gaussianeditor_amd.gaussian_renderer import render

    case = make_case(2000, 96, 64, seed=3, s0=0.05)
    pc = _PC(case["sc"])
    a = render(case["cam"], pc, PIPE, case["bg"])["render"]
    b = render(case["cam"], pc, SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=True, debug=False),
               case["bg"])["render"]
    assert float((a - b).abs().max()) < 2e-5  # SH evaluated in PyTorch vs in the rasterizer
    mask = (torch.rand(2000, 1, generator=torch.Generator().manual_seed(1)) > 0.5).float().repeat(1, 3)
    c = render(case["cam"], pc, PIPE, case["bg"], override_color=mask)["render"]
    f = oracle_forward(oracle, case, colors_precomp=mask)
    assert np.array_equal(c.detach().numpy(), f["color"])


def test_l1_validation_and_arity(monkeypatch):
    oracle_backend.install(monkeypatch)
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizer

    case = make_case(50, 32, 32, seed=1)
    sc = case["sc"]
    rast = GaussianRasterizer(settings(case, "cpu"))
    x, o = sc["xyz"], sc["opacity"]
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rast(x, x, o, scales=sc["scaling"], rotations=sc["rotation"])
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rast(x, x, o, shs=sc["features"], colors_precomp=torch.ones(50, 3), scales=sc["scaling"], rotations=sc["rotation"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        rast(x, x, o, shs=sc["features"], scales=sc["scaling"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        rast(x, x, o, shs=sc["features"], scales=sc["scaling"], rotations=sc["rotation"], cov3D_precomp=torch.ones(50, 6))
    out = rast(x, x, o, shs=sc["features"], scales=sc["scaling"], rotations=sc["rotation"])
    assert len(out) == 3 and out[0].shape == (3, 32, 32) and out[1].shape == (50,) and out[2].shape == (1, 32, 32)
    vis = rast.markVisible(x)
    assert vis.dtype == torch.bool and vis.shape == (50,)


def test_debug_snapshot_on_failure(monkeypatch, tmp_path):
    """raster_settings.debug=True: inputs are dumped to snapshot_fw.dump when the native call throws
    (DGR/diff_gaussian_rasterization/__init__.py:88-107)."""
    import gaussianeditor_amd.diff_gaussian_rasterization as dgr

    def boom(*a, **kw):
        raise RuntimeError("native failure")

    monkeypatch.setattr(dgr._C, "rasterize_gaussians", boom)
    monkeypatch.chdir(tmp_path)
    case = make_case(10, 16, 16, seed=1)
    sc = case["sc"]
    rast = dgr.GaussianRasterizer(settings(case, "cpu", debug=True))
    with pytest.raises(RuntimeError, match="native failure"):
        rast(sc["xyz"], sc["xyz"], sc["opacity"], shs=sc["features"], scales=sc["scaling"], rotations=sc["rotation"])
    dump = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)
    assert len(dump) == 19 and torch.equal(dump[1], sc["xyz"])


def test_apply_weights_facade(oracle, monkeypatch):
    """GaussianModel.apply_weights calling convention (scene/gaussian_model.py:817-832): positional args,
    in-place accumulation, camera2rasterizer with a zero background and SH degree 0."""
    oracle_backend.install(monkeypatch)
    from gaussianeditor_amd.gaussian_renderer import camera2rasterizer

    P, W, H = 1500, 80, 64
    case = make_case(P, W, H, seed=7, s0=0.06)
    sc = case["sc"]
    weights = torch.zeros(P, 1)
    cnt = torch.zeros(P, 1, dtype=torch.int32)
    mask = (torch.rand(1, H, W, generator=torch.Generator().manual_seed(2)) > 0.3).float()
    rast = camera2rasterizer(case["cam"], torch.tensor([0.0, 0.0, 0.0]))
    assert rast.apply_weights(sc["xyz"], None, sc["opacity"], None, weights, sc["scaling"], sc["rotation"], None, cnt,
                              mask) is None
    assert int(cnt.sum()) > 0 and float(weights.sum()) > 0
    sel = (weights / (cnt + 1e-7))[:, 0] > 0.5  # GassuianEditor.py:134-137
    assert sel.dtype == torch.bool


def test_install_registers_module_names():
    import gaussianeditor_amd

    gaussianeditor_amd.install()
    import diff_gaussian_rasterization as d

    assert d.GaussianRasterizer is sys.modules["gaussianeditor_amd.diff_gaussian_rasterization"].GaussianRasterizer
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
    assert GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg",
                                                     "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos",
                                                     "prefiltered", "debug")


def test_render_with_semantic_color_cpu(oracle, monkeypatch):
    """Extension: render(..., semantic_color=c) returns the normal outputs plus the image a
    second render(..., override_color=c) would give, without gradients and without disturbing the main backward."""
    oracle_backend.install(monkeypatch)
    from gaussianeditor_amd.gaussian_renderer import render

    case = make_case(3000, 128, 96, seed=2, s0=0.05, nviews=1, bg=(0.1, 0.0, 0.2))
    pc = _PC(case["sc"])
    mask = (torch.rand(3000, generator=torch.Generator().manual_seed(0)) > 0.5).float()
    sem = mask[:, None].repeat(1, 3)
    out = render(case["cam"], pc, PIPE, case["bg"], semantic_color=sem)
    two = render(case["cam"], _PC(case["sc"]), PIPE, case["bg"], override_color=sem)["render"]
    assert out["semantic"].shape == (3, 96, 128) and not out["semantic"].requires_grad
    assert torch.equal(out["semantic"], two.detach())
    plain = render(case["cam"], _PC(case["sc"]), PIPE, case["bg"])
    assert torch.equal(out["render"].detach(), plain["render"].detach())
    G = seed_gradient(96, 128, 3) * 96 * 128
    (out["render"] * G).sum().backward()
    pc2 = _PC(case["sc"])
    (render(case["cam"], pc2, PIPE, case["bg"])["render"] * G).sum().backward()
    assert torch.equal(pc.get_xyz.grad, pc2.get_xyz.grad) and torch.equal(pc.get_features.grad, pc2.get_features.grad)
