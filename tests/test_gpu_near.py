"""GPU: the Delete path's neighbour query over the C ABI (gsr_near_points) against the oracle's restatement of
GaussianModel.get_near_gaussians_by_mask / K_nearest_neighbors (gaussiansplatting/scene/gaussian_model.py:865-898,
gaussiansplatting/knn.py) and the fixture generated from the reference's own code (tests/golden/near_points.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "near_points.npz")


def _near(ref, qry, th, dist=True):
    from gaussianeditor_amd.near import near_points

    out = near_points(torch.from_numpy(ref).to(DEV), torch.from_numpy(qry).to(DEV), th, return_dist=dist)
    torch.cuda.synchronize()
    return (out[0].cpu().numpy(), out[1].cpu().numpy()) if dist else out.cpu().numpy()


def _check(oracle, ref, qry, th):
    want_near, want_dist = oracle.near_points(ref, qry, th)
    near, dist = _near(ref, qry, th)
    assert near.dtype == bool and np.array_equal(near, want_near)
    # the distance is exact (bit for bit) wherever it decides the mask; beyond the search radius it is +inf or exact
    assert np.array_equal(dist[want_near].view(np.uint32), want_dist[want_near].view(np.uint32))
    far = ~want_near
    assert (np.isinf(dist[far]) | (dist[far] == want_dist[far])).all() and (dist[far] > np.float32(th)).all()
    assert np.array_equal(_near(ref, qry, th, dist=False), want_near)  # the first-hit variant: same mask
    return int(want_near.sum())


@pytest.mark.parametrize("n_ref,n_query,kind", [(1, 300, "uniform"), (700, 1, "uniform"), (1024, 1025, "uniform"),
                                                (5000, 7000, "uniform"), (5000, 7000, "clustered"), (3000, 3000, "plane"),
                                                (2000, 2000, "duplicates")])
def test_near_points_vs_oracle(oracle, n_ref, n_query, kind):
    rng = np.random.default_rng(n_ref + 3 * n_query + len(kind))
    ref = rng.uniform(-1, 1, (n_ref, 3)).astype(np.float32)
    qry = rng.uniform(-1, 1, (n_query, 3)).astype(np.float32)
    if kind == "clustered":
        ref = (ref * 0.05 + rng.integers(0, 3, (n_ref, 3))).astype(np.float32)
        qry = (qry * 0.08 + rng.integers(0, 3, (n_query, 3))).astype(np.float32)
    elif kind == "plane":
        ref[:, 2] = 0.25  # degenerate bounding box of the reference set along one axis
    elif kind == "duplicates":
        qry[: n_query // 2] = ref[: n_query // 2]  # distance exactly 0
    hits = [_check(oracle, ref, qry, th) for th in (0.0, 0.02, 0.1, 0.5, 100.0)]
    assert hits == sorted(hits) and hits[-1] == n_query
    if kind == "duplicates":
        assert hits[0] >= n_query // 2


def test_near_points_threshold_is_compared_in_float32(oracle):
    ref = np.zeros((1, 3), np.float32)
    qry = np.array([[np.float32(0.1), 0, 0], [np.nextafter(np.float32(0.1), np.float32(1)), 0, 0], [0, 0, 0]], np.float32)
    near, dist = _near(ref, qry, 0.1)
    assert near.tolist() == [True, False, True] and dist[0] == np.float32(0.1) and dist[2] == 0
    # distances that need float64: 3-4-5 scaled triangles whose float32 squared sum would round differently
    rng = np.random.default_rng(1)
    ref = (rng.standard_normal((4000, 3)) * 100).astype(np.float32)
    qry = (ref[rng.integers(0, 4000, 6000)] + rng.standard_normal((6000, 3)).astype(np.float32) * 0.07).astype(np.float32)
    assert 1000 < _check(oracle, ref, qry, 0.1) < 6000


def test_near_points_empty_sets_and_validation():
    from gaussianeditor_amd.near import near_points

    z = torch.zeros(0, 3, device=DEV)
    o = torch.ones(5, 3, device=DEV)
    near, dist = near_points(z, o, 0.1, return_dist=True)
    assert near.shape == (5,) and not near.any() and torch.isinf(dist).all()
    near, dist = near_points(o, z, 0.1, return_dist=True)
    assert near.shape == (0,) and dist.shape == (0,)
    assert near_points(o[:, [2, 1, 0]][::2], o, 0.1).all()  # non-contiguous input is made contiguous on the way in
    with pytest.raises(RuntimeError):
        near_points(o.cpu(), o, 0.1)
    with pytest.raises(RuntimeError):
        near_points(o, o.double(), 0.1)
    with pytest.raises(RuntimeError):
        near_points(o, torch.ones(5, 4, device=DEV), 0.1)
    with pytest.raises(ValueError):
        near_points(o, o, -1.0)
    # non-finite coordinates: the reference's KDTree raises ValueError on either side; without the check they are never near
    bad = o.clone()
    bad[2, 1] = float("nan")
    with pytest.raises(ValueError, match="ref_xyz must be finite"):
        near_points(bad, o, 0.1)
    with pytest.raises(ValueError, match="query_xyz must be finite"):
        near_points(o, bad, 0.1)
    assert near_points(o, bad, 0.1, check_finite=False).tolist() == [True, True, False, True, True]


def test_get_near_gaussians_by_mask_matches_reference_fixture(oracle):
    from gaussianeditor_amd import near

    z = np.load(GOLD)
    xyz, mask = torch.from_numpy(z["xyz"]).to(DEV), torch.from_numpy(z["mask"]).to(DEV)
    for i in range(3):
        got = near.get_near_gaussians_by_mask(xyz, mask[:, None], float(z[f"thresh{i}"]))
        assert got.dtype == torch.bool and got.is_cuda and np.array_equal(got.cpu().numpy(), z[f"near{i}"]), i
    # the raw query of the fixture: the KDTree's float64 -> float32 distances, bit for bit where they decide the mask
    near_m, dist = _near(z["xyz"][z["mask"]], z["xyz"][~z["mask"]], 0.2)
    sel = z["nn_dist"] <= np.float32(0.2)
    assert np.array_equal(near_m, sel) and np.array_equal(dist[sel].view(np.uint32), z["nn_dist"][sel].view(np.uint32))

    class Model:  # the binding INTEGRATION.md section 6 describes: the method of the reference's GaussianModel replaced
        _xyz = xyz

    near.patch_gaussian_model(Model)
    assert np.array_equal(Model().get_near_gaussians_by_mask(mask).cpu().numpy(), z["near0"])


def test_near_points_scene_size_vs_kdtree():
    """An edit-sized query (200 k object points, 800 k remaining) against scipy's KDTree -- the reference's own path."""
    from scipy.spatial import KDTree

    rng = np.random.default_rng(11)
    xyz = (rng.standard_normal((1_000_000, 3)) * 2.0).astype(np.float32)
    mask = np.linalg.norm(xyz - np.float32([0.5, 0, 0]), axis=1) < 1.25
    ref, qry = xyz[mask], xyz[~mask]
    near, dist = _near(ref, qry, 0.1)
    sel = rng.choice(qry.shape[0], 40000, replace=False)
    want = KDTree(ref).query(qry[sel], k=1)[0].astype(np.float32)
    assert np.array_equal(near[sel], want <= np.float32(0.1)) and 0.005 < near.mean() < 0.5
    hit = near[sel]
    assert np.array_equal(dist[sel][hit], want[hit])
    from gaussianeditor_amd import near as near_mod

    got = near_mod.get_near_gaussians_by_mask(torch.from_numpy(xyz).to(DEV), torch.from_numpy(mask).to(DEV), 0.1)
    assert got.shape == (qry.shape[0],) and 0 < int(got.sum()) <= int(near.sum())
    assert not (got.cpu().numpy() & ~near).any()  # the bounding-box filter only removes candidates
