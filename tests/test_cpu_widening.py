"""CPU: the SURVEY.md section 8(f) rank-1 "import enablers" -- the oracle of simple_knn.distCUDA2 pinned against an
independent exact k-NN (scipy cKDTree), and the PLY shim against the byte layout the reference's save_ply/load_ply and
fetchPly/storePly rely on (scene/gaussian_model.py:410-505, scene/dataset_readers.py:155-178)."""
import io
import sys

import numpy as np
import pytest

from gaussianeditor_amd.compat import plyfile as ply


def _kdtree_mean3(pts):
    from scipy.spatial import cKDTree

    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    return (d[:, 1:] ** 2).mean(axis=1)


@pytest.mark.parametrize("P,kind", [(4, "uniform"), (257, "uniform"), (3000, "uniform"), (3000, "clustered")])
def test_oracle_knn_matches_kdtree(oracle, P, kind):
    rng = np.random.default_rng(P)
    pts = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    if kind == "clustered":
        pts = (pts * 0.01 + rng.integers(0, 5, (P, 1)).astype(np.float32)).astype(np.float32)
    got = oracle.knn_mean_dist2(pts)
    want = _kdtree_mean3(pts)
    assert got.dtype == np.float32 and got.shape == (P,)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-12)


def test_oracle_knn_duplicates_and_tiny(oracle):
    # coincident points are neighbours at distance 0 (only the query index itself is skipped, simple_knn.cu:133-160)
    pts = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]], np.float32)
    got = oracle.knn_mean_dist2(pts)
    assert got[0] == np.float32((0 + 1 + 4) / 3.0) and got[1] == got[0]
    assert oracle.knn_mean_dist2(np.zeros((0, 3), np.float32)).shape == (0,)


def _gaussian_table(P, n_rest=45, seed=0):
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    rng = np.random.default_rng(seed)
    attrs = rng.standard_normal((P, len(names))).astype(np.float32)
    tab = np.empty(P, dtype=[(n, "f4") for n in names])
    tab[:] = list(map(tuple, attrs))
    return names, attrs, tab


def test_ply_roundtrip_gaussian_checkpoint(tmp_path):
    names, attrs, tab = _gaussian_table(123)
    path = str(tmp_path / "point_cloud.ply")
    ply.PlyData([ply.PlyElement.describe(tab, "vertex")]).write(path)
    # the exact bytes: ASCII header, then P rows of 62 little-endian floats
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode("ascii").splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 123"]
    assert lines[3:] == [f"property float {n}" for n in names]
    assert body == attrs.astype("<f4").tobytes()

    back = ply.PlyData.read(path)
    v = back.elements[0]
    assert v.name == "vertex" and v.count == 123 and len(back) == 1 and "vertex" in back
    assert [p.name for p in v.properties] == names
    rest = sorted((p.name for p in v.properties if p.name.startswith("f_rest_")), key=lambda s: int(s.split("_")[-1]))
    assert len(rest) == 45  # load_ply's 3*(max_sh_degree+1)**2 - 3 check (gaussian_model.py:479)
    for i, n in enumerate(names):
        assert np.array_equal(np.asarray(v[n]), attrs[:, i])
    assert np.array_equal(np.asarray(back["vertex"]["opacity"]), attrs[:, names.index("opacity")])


def test_ply_colmap_points_mixed_types_and_streams():
    # storePly / fetchPly layout: float xyz + normals, uchar rgb (dataset_readers.py:155-178)
    P = 50
    rng = np.random.default_rng(3)
    dt = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("red", "u1"), ("green", "u1"),
          ("blue", "u1")]
    tab = np.empty(P, dtype=dt)
    for n, t in dt:
        tab[n] = rng.uniform(0, 255, P).astype(t)
    buf = io.BytesIO()
    ply.PlyData([ply.PlyElement.describe(tab, "vertex")]).write(buf)
    assert len(buf.getvalue().split(b"end_header\n", 1)[1]) == P * 27
    buf.seek(0)
    v = ply.PlyData.read(buf)["vertex"]
    for n, t in dt:
        assert np.array_equal(v[n], tab[n]) and v[n].dtype == np.dtype(t)

    # ascii and big-endian files (as written by other tools) are readable too
    txt = b"ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty uchar red\nend_header\n" \
          b"0.5 7\n-1.25 255\n"
    a = ply.PlyData.read(io.BytesIO(txt))
    assert a.comments == ["made by hand"] and a.text
    assert np.array_equal(a["vertex"]["x"], np.float32([0.5, -1.25])) and np.array_equal(a["vertex"]["red"], np.uint8([7, 255]))
    be = b"ply\nformat binary_big_endian 1.0\nelement vertex 1\nproperty int k\nproperty double w\nend_header\n" + \
         np.array([(5, 2.5)], dtype=[("k", ">i4"), ("w", ">f8")]).tobytes()
    b = ply.PlyData.read(io.BytesIO(be))["vertex"]
    assert int(b["k"][0]) == 5 and float(b["w"][0]) == 2.5
    out = io.BytesIO()
    ply.PlyData(a.elements, text=True).write(out)
    again = ply.PlyData.read(io.BytesIO(out.getvalue()))["vertex"]
    assert np.array_equal(again["x"], a["vertex"]["x"]) and np.array_equal(again["red"], a["vertex"]["red"])


def test_ply_errors():
    with pytest.raises(ply.PlyParseError):
        ply.PlyData.read(io.BytesIO(b"plx\n"))
    with pytest.raises(ply.PlyParseError):  # truncated body
        ply.PlyData.read(io.BytesIO(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nend_header\n\0\0\0\0"))
    with pytest.raises(ply.PlyParseError):  # list properties (meshes) are outside what the path needs
        ply.PlyData.read(io.BytesIO(b"ply\nformat ascii 1.0\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n"))
    with pytest.raises(TypeError):
        ply.PlyElement.describe(np.zeros((3, 3), np.float32), "vertex")
    with pytest.raises(KeyError):
        ply.PlyData([])["vertex"]


def test_install_registers_import_enablers(monkeypatch):
    import gaussianeditor_amd

    for name in ("simple_knn", "simple_knn._C", "plyfile", "diff_gaussian_rasterization", "diff_gaussian_rasterization._C"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    try:
        gaussianeditor_amd.install()
    except OSError:
        pytest.skip("libgsr_hip.so not built")
    import plyfile
    from simple_knn._C import distCUDA2

    assert hasattr(plyfile, "PlyData") and hasattr(plyfile, "PlyElement")
    import torch

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        distCUDA2(torch.zeros(8, 3))
    for name in ("simple_knn", "simple_knn._C", "plyfile", "diff_gaussian_rasterization", "diff_gaussian_rasterization._C"):
        monkeypatch.delitem(sys.modules, name, raising=False)


# ---------------------------------------------------------------------------------------------------------------------
# the Adam oracle pinned against torch.optim.Adam itself (the reference's optimizer,
# gaussiansplatting/scene/gaussian_model.py:369), with the gradient mask of apply_grad_mask (:841-856) and the
# anchor-loss gradient (:152-184) obtained from autograd
def _ref_groups(P, seed):
    import torch

    g = torch.Generator().manual_seed(seed)
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    params = {k: torch.randn(s, generator=g).requires_grad_(True) for k, s in shapes.items()}
    return params, lrs, g


def test_oracle_adam_matches_torch_adam(oracle):
    import torch

    P = 257
    params, lrs, gen = _ref_groups(P, 0)
    opt = torch.optim.Adam([{"params": [p], "lr": lrs[k], "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
    mine = {k: [p.detach().numpy().copy(), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)] for k, p in params.items()}
    for step in range(1, 6):
        for k, p in params.items():
            p.grad = torch.randn(p.shape, generator=gen) * (0.1 if step % 2 else 10.0)
            if step == 3:
                p.grad[: P // 2] = 0.0  # rows without a gradient still decay their moments and move
        for k, p in params.items():
            oracle.adam_step(mine[k][0], p.grad.numpy(), mine[k][1], mine[k][2], lrs[k], step, eps=1e-15)
        opt.step()
        for k, p in params.items():
            st = opt.state[p]
            # tolerance: 1e-6 of the tensor's magnitude (torch's CPU lerp / addcmul kernels contract a*b+c into fma, the
            # oracle rounds every operation: differences of one ulp of the larger operand where terms cancel)
            for mine_t, ref_t in ((mine[k][0], p.detach().numpy()), (mine[k][1], st["exp_avg"].numpy()),
                                  (mine[k][2], st["exp_avg_sq"].numpy())):
                np.testing.assert_allclose(mine_t, ref_t, rtol=1e-5, atol=1e-6 * float(np.abs(ref_t).max()))


def test_oracle_adam_mask_and_anchor_match_autograd(oracle):
    import torch

    P = 100
    params, lrs, gen = _ref_groups(P, 1)
    mask = torch.rand(P, generator=gen) > 0.4
    weight = torch.rand(P, generator=gen)  # anchor_weight_list per row (schedule[generation])
    lam = 3.0
    anchors = {k: p.detach().clone() + 0.05 * torch.randn(p.shape, generator=gen) for k, p in params.items()}
    opt = torch.optim.Adam([{"params": [p], "lr": lrs[k]} for k, p in params.items()], lr=0.0, eps=1e-15)
    masked_fields = ("xyz", "f_dc", "f_rest", "opacity", "scaling")  # apply_grad_mask's list: rotation is not in it
    for k in masked_fields:  # the reference's hook
        params[k].register_hook(lambda grad, m=mask: grad * (m[:, None] if grad.ndim == 2 else m[:, None, None]))
    mine = {k: [p.detach().numpy().copy(), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)] for k, p in params.items()}
    for step in range(1, 4):
        data = {k: torch.randn(p.shape, generator=gen) for k, p in params.items()}
        # a data term + the reference's anchor term: mean over the masked rows of w_row * (p - a)^2
        loss = sum((p * data[k]).sum() for k, p in params.items())
        for k, p in params.items():
            delta = torch.nn.functional.mse_loss(p[mask], anchors[k][mask], reduction="none")
            delta = delta * (weight[mask][:, None] if delta.ndim == 2 else weight[mask][:, None, None])
            loss = loss + lam * delta.mean()
        for p in params.values():
            p.grad = None
        loss.backward()
        nsel = int(mask.sum())
        for k, p in params.items():
            n_elem = nsel * (p.numel() // P)
            oracle.adam_step(mine[k][0], data[k].numpy(), mine[k][1], mine[k][2], lrs[k], step, eps=1e-15,
                             row_mask=mask.numpy(), masked=k in masked_fields, anchor=anchors[k].numpy(),
                             anchor_scale=lam * 2.0 / n_elem, row_weight=(weight * mask).numpy())
        opt.step()
        for k, p in params.items():
            np.testing.assert_allclose(mine[k][0], p.detach().numpy(), rtol=1e-5, atol=2e-6, err_msg=k)


def test_fused_adam_has_no_cpu_fallback():
    import torch

    from gaussianeditor_amd.optim import FusedMaskedAdam

    p = torch.zeros(4, 3, requires_grad=True)
    p.grad = torch.ones(4, 3)
    try:
        opt = FusedMaskedAdam([p], lr=0.1)
    except ImportError:
        pytest.skip("libgsr_hip.so not built")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_oracle_compact_rows_is_boolean_indexing(oracle):
    import torch

    g = torch.Generator().manual_seed(0)
    keep = torch.rand(1000, generator=g) > 0.3
    ts = [torch.randn(1000, 3, generator=g), torch.randn(1000, 15, 3, generator=g), torch.arange(1000), keep.clone()]
    out = oracle.compact_rows([t.numpy() for t in ts], keep.numpy())
    for o, t in zip(out, ts):
        assert np.array_equal(o, t[keep].numpy())


def test_fused_adam_rejects_stale_anchor_state():
    """after densify / prune the parameters are new tensors; anchors keyed by the old ones and a row_weight
    of the old length must raise instead of being ignored / read out of bounds (checked before any native call)."""
    import torch

    from gaussianeditor_amd.optim import FusedMaskedAdam

    p = torch.nn.Parameter(torch.zeros(8, 3))
    opt = FusedMaskedAdam([{"params": [p], "name": "xyz"}], lr=1e-3)
    opt.set_anchor(p, torch.zeros(8, 3), 1.0, row_weight=torch.ones(8))
    q = torch.nn.Parameter(torch.zeros(10, 3))  # what densification_postfix leaves in the group
    opt.param_groups[0]["params"][0] = q
    q.grad = torch.zeros_like(q)
    with pytest.raises(RuntimeError, match="no longer parameters"):
        opt.step()
    opt.clear_anchors()
    assert opt._row_weight is None and not opt._anchors
    opt.set_anchor(q, torch.zeros(10, 3), 1.0, row_weight=torch.ones(8))  # a weight vector of the old length
    with pytest.raises(RuntimeError):  # CPU tensors: "no CPU fallback" is raised first on this machine; on the GPU the length check
        opt.step()


def test_scene_ply_round_trip_and_layout(tmp_path):
    """gaussianeditor_amd.scene_ply against the byte layout of GaussianModel.save_ply / load_ply
    (gaussiansplatting/scene/gaussian_model.py:396-445, 455-533): column order, channel-major f_dc / f_rest, raw (logit /
    log) values, and the round trip through the file."""
    import torch

    from gaussianeditor_amd import scene_ply
    from gaussianeditor_amd.compat import plyfile

    g = torch.Generator().manual_seed(2)
    P = 37
    raw = dict(xyz=torch.randn(P, 3, generator=g), f_dc=torch.randn(P, 1, 3, generator=g), f_rest=torch.randn(P, 15, 3, generator=g),
               opacity=torch.randn(P, 1, generator=g), scaling=torch.randn(P, 3, generator=g), rotation=torch.randn(P, 4, generator=g))
    path = str(tmp_path / "point_cloud.ply")
    scene_ply.save_gaussians_ply(path, **raw)
    v = plyfile.PlyData.read(path).elements[0]
    names = [p.name for p in v.properties]
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9:54] == [f"f_rest_{i}" for i in range(45)] and names[54:] == ["opacity", "scale_0", "scale_1", "scale_2",
                                                                                  "rot_0", "rot_1", "rot_2", "rot_3"]
    assert all(p.val_dtype == "f4" for p in v.properties) and len(v.data) == P
    # channel-major: f_rest_k for k < 15 is the RED channel of coefficient 1 + k (save_ply transposes (P,15,3) -> (P,3,15))
    assert np.array_equal(np.asarray(v["f_rest_3"]), raw["f_rest"][:, 3, 0].numpy())
    assert np.array_equal(np.asarray(v["f_rest_18"]), raw["f_rest"][:, 3, 1].numpy())
    assert np.array_equal(np.asarray(v["nx"]), np.zeros(P, np.float32))
    back = scene_ply.load_gaussians_ply(path)
    assert back["max_sh_degree"] == 3
    for k, t in raw.items():
        assert torch.equal(back[k], t), k
    act = scene_ply.activated(back)
    assert act["features"].shape == (P, 16, 3) and torch.allclose(act["rotation"].norm(dim=1), torch.ones(P))
    assert torch.equal(act["opacity"], torch.sigmoid(raw["opacity"])) and torch.equal(act["scaling"], torch.exp(raw["scaling"]))
    # SH degree 1 file (3 * 4 - 3 = 9 f_rest columns)
    scene_ply.save_gaussians_ply(path, raw["xyz"], raw["f_dc"], raw["f_rest"][:, :3], raw["opacity"], raw["scaling"], raw["rotation"])
    assert scene_ply.load_gaussians_ply(path)["max_sh_degree"] == 1
