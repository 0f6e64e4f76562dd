import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianeditor_amd.diff_gaussian_rasterization import _C
from gaussianeditor_amd.synth import ring_cameras, synth_scene
P = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
dev = torch.device("cuda:0")
sc = synth_scene(P, seed=0, s0=0.01)
cam = ring_cameras(8, 1920, 1080)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
d = lambda t: t.to(dev)
e = torch.empty(0, device=dev)
args = (d(sc["bg"]), d(sc["xyz"]), e, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), 1.0, e, d(cam.world_view_transform),
        d(cam.full_proj_transform), tfx, tfy, 1080, 1920, d(sc["features"]), 3, d(cam.camera_center), False, False)
for _ in range(3): out = _C.rasterize_gaussians(*args)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): out = _C.rasterize_gaussians(*args)
torch.cuda.synchronize(); print("ms/render", (time.perf_counter() - t0) / 10 * 1e3, "R", out[0])
