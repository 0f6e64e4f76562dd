"""The editor's loop shape at its own resolution (512x512, SURVEY.md 8(d) C3 substitute): render + semantic image + backward
through gaussian_renderer.render(); prints ms/step.  Run under rocprofv3 --kernel-trace and feed the database to
tools/rocpd_timeline.py for the kernel timeline of one step."""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianeditor_amd.gaussian_renderer import render  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
PIPE = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)


class PC:
    def __init__(self, sc):
        self.t = {k: sc[k].to(dev).requires_grad_(True) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        self.active_sh_degree = 3
        self.max_sh_degree = 3

    get_xyz = property(lambda s: s.t["xyz"])
    get_opacity = property(lambda s: s.t["opacity"])
    get_scaling = property(lambda s: s.t["scaling"])
    get_rotation = property(lambda s: s.t["rotation"])
    get_features = property(lambda s: s.t["features"])


pc = PC(synth_scene(P, seed=0, s0=0.01))
cam = ring_cameras(8, 512, 512)[0].to(dev)
G = seed_gradient(512, 512, 0).to(dev)
bg = torch.zeros(3, device=dev)
mask = (torch.rand(P, 1, device=dev) > 0.5).float().repeat(1, 3)


def step():
    a = render(cam, pc, PIPE, bg, semantic_color=mask)
    (a["render"] * G).sum().backward()
    for v in pc.t.values():
        v.grad = None


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
