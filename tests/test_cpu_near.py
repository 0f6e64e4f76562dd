"""CPU: the Delete path's neighbour query.  The oracle's restatement of
GaussianModel.get_near_gaussians_by_mask / K_nearest_neighbors (gaussiansplatting/scene/gaussian_model.py:865-898,
gaussiansplatting/knn.py) against the fixture produced by running the reference's own code (tests/golden/make_golden.py),
against scipy's KDTree directly, and -- where /root/reference exists -- against the reference's method run live."""
import ctypes
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "near_points.npz")


def test_oracle_near_points_matches_reference_fixture(oracle):
    z = np.load(GOLD)
    xyz, mask = z["xyz"], z["mask"]
    near, dist = oracle.near_points(xyz[mask], xyz[~mask], 0.1)
    assert dist.dtype == np.float32 and np.array_equal(dist.view(np.uint32), z["nn_dist"].view(np.uint32))  # bit for bit
    assert np.array_equal(near, z["nn_dist"] <= np.float32(0.1))
    assert near[7 - int(mask[:7].sum())] and dist[7 - int(mask[:7].sum())] == 0  # the planted duplicate of an object point
    for i in range(3):
        got = oracle.get_near_gaussians_by_mask(xyz, mask, float(z[f"thresh{i}"]))
        assert got.dtype == bool and np.array_equal(got, z[f"near{i}"]), i
    assert z["near0"].sum() > 20 and z["near1"].sum() > z["near0"].sum() > z["near2"].sum() > 0  # the fixture discriminates


@pytest.mark.parametrize("n_ref,n_query,kind", [(1, 50, "uniform"), (500, 1, "uniform"), (3000, 4000, "uniform"),
                                                (3000, 4000, "clustered")])
def test_oracle_near_points_matches_kdtree(oracle, n_ref, n_query, kind):
    from scipy.spatial import KDTree

    rng = np.random.default_rng(n_ref + n_query)
    ref = rng.uniform(-1, 1, (n_ref, 3)).astype(np.float32)
    qry = rng.uniform(-1, 1, (n_query, 3)).astype(np.float32)
    if kind == "clustered":
        ref = (ref * 0.05 + rng.integers(0, 3, (n_ref, 3))).astype(np.float32)
        qry = (qry * 0.08 + rng.integers(0, 3, (n_query, 3))).astype(np.float32)
    want = KDTree(ref).query(qry, k=1)[0].astype(np.float32)
    for th in (0.0, 0.03, 0.1, 10.0):
        near, dist = oracle.near_points(ref, qry, th)
        assert np.array_equal(dist, want) and np.array_equal(near, want <= np.float32(th))


def test_oracle_near_points_edges(oracle):
    near, dist = oracle.near_points(np.zeros((0, 3), np.float32), np.ones((5, 3), np.float32), 0.1)
    assert not near.any() and np.isinf(dist).all()
    near, dist = oracle.near_points(np.ones((5, 3), np.float32), np.zeros((0, 3), np.float32), 0.1)
    assert near.shape == (0,) and dist.shape == (0,)
    # the comparison is made in float32: float32(0.1) > 0.1, and a distance that rounds to float32(0.1) is "near"
    ref = np.zeros((1, 3), np.float32)
    qry = np.array([[np.float32(0.1), 0, 0], [np.nextafter(np.float32(0.1), np.float32(1)), 0, 0]], np.float32)
    near, _ = oracle.near_points(ref, qry, 0.1)
    assert near.tolist() == [True, False]


@pytest.mark.skipif(not os.path.isdir("/root/reference/gaussiansplatting"), reason="the reference checkout is only in the development container")
def test_oracle_matches_reference_method_live(oracle):
    import importlib.util
    import sys

    import torch

    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    knn = mg.load(os.path.join(mg.REF, "knn.py"), "ref_knn")
    fn = mg.reference_method(os.path.join(mg.REF, "scene", "gaussian_model.py"), "get_near_gaussians_by_mask",
                             {"torch": torch, "K_nearest_neighbors": knn.K_nearest_neighbors})
    rng = np.random.default_rng(5)
    for n, th in ((2000, 0.1), (5000, 0.04)):
        xyz = (rng.standard_normal((n, 3)) * 0.6).astype(np.float32)
        mask = np.linalg.norm(xyz, axis=1) < 0.5

        class Stub:
            _xyz = torch.from_numpy(xyz)

        want = fn(Stub(), torch.from_numpy(mask)[:, None], dist_thresh=th).numpy()
        assert np.array_equal(oracle.get_near_gaussians_by_mask(xyz, mask, th), want) and want.any()


def test_abi_near_points_argument_checks():
    from gaussianeditor_amd import _native

    L = _native.lib()
    sz = ctypes.c_size_t(0)
    assert L.gsr_near_workspace_size(1000, ctypes.byref(sz)) == 0 and sz.value > 0
    assert L.gsr_near_workspace_size(-1, ctypes.byref(sz)) == -1 and L.gsr_near_workspace_size(5, None) == -1
    assert L.gsr_near_points(None, 10, None, 0, None, ctypes.c_float(0.1), None, None, None) == 0        # no queries: no-op
    assert L.gsr_near_points(None, 10, None, 5, None, ctypes.c_float(0.1), None, None, None) == -1       # null pointers
    assert L.gsr_near_points(None, -1, None, 5, ctypes.c_void_p(256), ctypes.c_float(0.1), None, ctypes.c_void_p(256), None) == -1
    assert L.gsr_near_points(None, 10, ctypes.c_void_p(256), 5, ctypes.c_void_p(256), ctypes.c_float(-1.0), ctypes.c_void_p(256),
                             ctypes.c_void_p(256), None) == -1                                             # negative radius
    assert L.gsr_near_points(None, 10, ctypes.c_void_p(256), 5, ctypes.c_void_p(256), ctypes.c_float(float("nan")),
                             ctypes.c_void_p(256), ctypes.c_void_p(256), None) == -1


def test_near_module_refuses_host_tensors():
    import torch

    from gaussianeditor_amd import near

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        near.near_points(torch.zeros(4, 3), torch.zeros(4, 3), 0.1)
