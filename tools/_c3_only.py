import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
from gaussianeditor_amd.gaussian_renderer import render
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene
dev = torch.device("cuda:0")
PIPE = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
class PC:
    def __init__(self, sc):
        self.t = {k: sc[k].to(dev).requires_grad_(True) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        self.active_sh_degree = 3; self.max_sh_degree = 3
    get_xyz = property(lambda s: s.t["xyz"]); get_opacity = property(lambda s: s.t["opacity"])
    get_scaling = property(lambda s: s.t["scaling"]); get_rotation = property(lambda s: s.t["rotation"])
    get_features = property(lambda s: s.t["features"])
sc = synth_scene(1_000_000, seed=0, s0=0.01); pc = PC(sc)
cam = ring_cameras(8, 512, 512)[0].to(dev); G = seed_gradient(512, 512, 0).to(dev); bg = torch.zeros(3, device=dev)
mask = (torch.rand(1_000_000, 1, device=dev) > 0.5).float().repeat(1, 3)
def step():
    a = render(cam, pc, PIPE, bg, semantic_color=mask)
    (a["render"] * G).sum().backward()
    for v in pc.t.values(): v.grad = None
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
