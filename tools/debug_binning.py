"""Development aid: dump the grouped binning's intermediate arrays for one small case and compare them with a numpy model
built from the CPU oracle's per-tile lists (run on the GPU box)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import hip_state, make_case, oracle_forward  # noqa: E402

from gaussianeditor_amd import _native  # noqa: E402
from oracle import cpu as O  # noqa: E402

DEV = "cuda:0"
P, W, H, s0, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (10000, 256, 256, 0.03, 1)
O.build()
case = make_case(P, W, H, seed=seed, s0=s0)
f = oracle_forward(O, case)
sc, cam = case["sc"], case["cam"]
L = _native.lib()
d = lambda t: t.to(DEV).contiguous()  # noqa: E731
s = torch.cuda.current_stream().cuda_stream
gb, _, ib = _native.scratch_sizes(P, 0, W, H)
geom = torch.zeros(gb, dtype=torch.uint8, device=DEV)
img = torch.zeros(ib, dtype=torch.uint8, device=DEV)
radii = torch.zeros(P, dtype=torch.int32, device=DEV)
counts = (ctypes.c_int64 * 2)()
t = dict(xyz=d(sc["xyz"]), sca=d(sc["scaling"]), rot=d(sc["rotation"]), op=d(sc["opacity"]), sh=d(sc["features"]),
         view=d(cam.world_view_transform), proj=d(cam.full_proj_transform), cp=d(cam.camera_center))
p = lambda x: x.data_ptr()  # noqa: E731
_native.check("pre", L.gsr_preprocess(s, P, 3, 16, p(t["xyz"]), p(t["sca"]), 1.0, p(t["rot"]), p(t["op"]), p(t["sh"]), None, None,
                                      p(t["view"]), p(t["proj"]), p(t["cp"]), W, H, case["tfx"], case["tfy"], 0, 0, 0, p(radii),
                                      p(geom), counts))
R, G = int(counts[0]), int(counts[1])
print("R", R, "oracle", f["num_rendered"], "G", G)
_, bb, _ = _native.scratch_sizes(P, R, W, H, G)
binning = torch.full((bb,), 0xAB, dtype=torch.uint8, device=DEV)
_native.check("bin", L.gsr_bin(s, P, R, G, W, H, p(geom), p(binning), p(img)))
torch.cuda.synchronize()
B = binning.cpu().numpy()
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
sgx, sgy = (gx + 7) // 8, (gy + 7) // 8
S = sgx * sgy
al = lambda v: (v + 255) // 256 * 256  # noqa: E731
batches = (G + 63) // 64
per = 1
while per < 8 and batches > 8192 * per:
    per *= 2
chunk = 64 * per
chunks = G // chunk + S
padded = chunks * chunk
bins = 256 if S <= 256 else 2048
sort_blocks = (G + 4095) // 4096
off = 0


def take(nbytes, dtype):
    global off
    a = B[off:off + nbytes].view(dtype)
    off += al(nbytes)
    return a


point_list = take(4 * R, np.uint32)
tile_total = take(4 * T, np.uint32)
tile_start = take(4 * (T + 1), np.uint32)
gval0 = take(4 * G, np.uint32)
gval1 = take(4 * padded, np.uint32)
gkey0 = take(4 * G, np.uint32)
gkey1 = take(4 * padded, np.uint32)
ghist = take(4 * bins * sort_blocks, np.uint32)
gbin_total = take(4 * bins, np.uint32)
chunk_cnt = take(2 * 64 * chunks, np.uint16).reshape(chunks, 64)
chunk_pre = take(4 * 64 * chunks, np.uint32).reshape(chunks, 64)
print("bytes carved", off, "allocated", bb, "chunk", chunk, "chunks", chunks, "S", S)
# ---- model from the oracle
rng = f["ranges"].reshape(T, 2).astype(np.int64)
exp_total = rng[:, 1] - rng[:, 0]
print("tile_total ok:", np.array_equal(tile_total, exp_total), "sum", int(tile_total.sum()), "expected", int(exp_total.sum()))
bad = np.nonzero(tile_total != exp_total)[0]
print("  tiles differing:", bad[:20], tile_total[bad[:20]], exp_total[bad[:20]])
print("gkey0 histogram:", np.bincount(gkey0 & 0xFFFF, minlength=S)[:S], "gbin_total:", gbin_total[:S], "sum", int(gbin_total.sum()))
real = gkey1 != 0xFFFFFFFF
print("real slots in the sorted side:", int(real.sum()), "of", padded)
cf = 0
for g in range(S):
    n = int(gbin_total[g])
    nch = (n + chunk - 1) // chunk
    seg = slice(cf * chunk, (cf + nch) * chunk)
    ok_keys = bool(np.all((gkey1[seg][real[seg]] & 0xFFFF) == g))
    print(f"  group {g}: len {n}, chunks [{cf},{cf + nch}), real in segment {int(real[seg].sum())}, keys ok {ok_keys}, "
          f"count rows sum {int(chunk_cnt[cf:cf + nch].sum())}")
    cf += nch
pl_ok = np.array_equal(point_list, f["point_list"])
print("point_list ok:", pl_ok)
if not pl_ok:
    i = int(np.nonzero(point_list != f["point_list"])[0][0])
    print("  first mismatch at", i, point_list[i:i + 8], f["point_list"][i:i + 8])
