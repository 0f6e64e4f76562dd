"""Development aid / profiles row: one densify-and-prune step's TENSOR SURGERY at 1 M Gaussians (SH3), three ways --
  torch     the reference's own operations: boolean-mask indexing and torch.cat per tensor + new nn.Parameters
            (gaussiansplatting/scene/gaussian_model.py:568-641), 6 parameters + 12 Adam moments + 4 bookkeeping tensors;
  fresh     gaussianeditor_amd.densify (one compaction / one append launch for all tensors, into fresh tensors);
  arena     gaussianeditor_amd.arena (compaction into the other half, appends in place, Parameters re-pointed).
A step = clone 1 % of the rows (append), split 1 % (append 2 %, prune the parents), prune 2 % -- the sequence of
densify_and_prune (:768-809).  Wall clock with a device synchronisation on both sides, median of the repetitions."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianeditor_amd import densify  # noqa: E402
from gaussianeditor_amd.arena import OptimizerArena  # noqa: E402

DEV = "cuda:0"
P0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
SHAPES = dict(xyz=(3,), f_dc=(1, 3), f_rest=(15, 3), opacity=(1,), scaling=(3,), rotation=(4,))


def make():
    g = torch.Generator(device=DEV).manual_seed(0)
    params = {k: torch.nn.Parameter(torch.randn((P0,) + s, device=DEV, generator=g)) for k, s in SHAPES.items()}
    opt = torch.optim.Adam([dict(params=[p], lr=1e-3, name=k) for k, p in params.items()], lr=0.0, eps=1e-15)
    for p in params.values():
        opt.state[p] = dict(step=torch.tensor(1.0), exp_avg=torch.randn_like(p), exp_avg_sq=torch.rand_like(p))
    extra = dict(accum=torch.rand(P0, 1, device=DEV), denom=torch.ones(P0, 1, device=DEV), radii=torch.rand(P0, device=DEV),
                 mask=torch.ones(P0, dtype=torch.bool, device=DEV))
    return params, opt, extra


def selections(P, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = torch.rand(P, device=DEV, generator=g)
    return r < 0.01, (r >= 0.01) & (r < 0.02)


def torch_step(opt, extra):
    def cat(ext):
        for group in opt.param_groups:
            e = ext[group["name"]]
            st = opt.state.pop(group["params"][0])
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(e)))
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(e)))
            group["params"][0] = torch.nn.Parameter(torch.cat((group["params"][0], e)).requires_grad_(True))
            opt.state[group["params"][0]] = st
        n = next(iter(ext.values())).shape[0]
        for k in extra:
            extra[k] = torch.cat((extra[k], torch.zeros((n,) + tuple(extra[k].shape[1:]), dtype=extra[k].dtype, device=DEV)))

    def prune(keep):
        for group in opt.param_groups:
            st = opt.state.pop(group["params"][0])
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep], st["exp_avg_sq"][keep]
            group["params"][0] = torch.nn.Parameter(group["params"][0][keep].requires_grad_(True))
            opt.state[group["params"][0]] = st
        for k in extra:
            extra[k] = extra[k][keep]

    P = opt.param_groups[0]["params"][0].shape[0]
    clone, split = selections(P, 1)
    cat({g["name"]: g["params"][0].detach()[clone] for g in opt.param_groups})
    P1 = opt.param_groups[0]["params"][0].shape[0]
    split = torch.cat((split, torch.zeros(P1 - P, dtype=torch.bool, device=DEV)))
    cat({g["name"]: g["params"][0].detach()[split].repeat(*([2] + [1] * (g["params"][0].dim() - 1))) for g in opt.param_groups})
    P2 = opt.param_groups[0]["params"][0].shape[0]
    prune(~torch.cat((split, torch.zeros(P2 - P1, dtype=torch.bool, device=DEV))))
    P3 = opt.param_groups[0]["params"][0].shape[0]
    prune(torch.rand(P3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)) >= 0.02)


def fresh_step(opt, extra):
    names = [g["name"] for g in opt.param_groups]
    P = opt.param_groups[0]["params"][0].shape[0]
    clone, split = selections(P, 1)

    def grow(ext):
        densify.cat_tensors_to_optimizer(opt, ext)
        n = next(iter(ext.values())).shape[0]
        out = densify.append_rows(list(extra.values()), [None] * len(extra), n=n)
        for k, v in zip(list(extra), out):
            extra[k] = v

    def prune(keep):
        densify.prune_optimizer(opt, keep)
        out = densify.compact_rows(list(extra.values()), keep)
        for k, v in zip(list(extra), out):
            extra[k] = v

    grow(dict(zip(names, densify.compact_rows([g["params"][0] for g in opt.param_groups], clone))))
    P1 = opt.param_groups[0]["params"][0].shape[0]
    split = torch.cat((split, torch.zeros(P1 - P, dtype=torch.bool, device=DEV)))
    picked = densify.compact_rows([g["params"][0] for g in opt.param_groups], split)
    grow({k: v.repeat(*([2] + [1] * (v.dim() - 1))) for k, v in zip(names, picked)})
    P2 = opt.param_groups[0]["params"][0].shape[0]
    prune(~torch.cat((split, torch.zeros(P2 - P1, dtype=torch.bool, device=DEV))))
    P3 = opt.param_groups[0]["params"][0].shape[0]
    prune(torch.rand(P3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)) >= 0.02)


def arena_step(oa):
    names = [n for n, _, _ in oa.groups]
    P = oa.P
    clone, split = selections(P, 1)
    oa.append(dict(zip(names, densify.compact_rows([p for _, p, _ in oa.groups], clone))))
    P1 = oa.P
    split = torch.cat((split, torch.zeros(P1 - P, dtype=torch.bool, device=DEV)))
    picked = densify.compact_rows([p for _, p, _ in oa.groups], split)
    oa.append({k: v.repeat(*([2] + [1] * (v.dim() - 1))) for k, v in zip(names, picked)})
    P2 = oa.P
    oa.prune(~torch.cat((split, torch.zeros(P2 - P1, dtype=torch.bool, device=DEV))))
    oa.prune(torch.rand(oa.P, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)) >= 0.02)


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return statistics.median(ts), min(ts)


res = {}
params, opt, extra = make()
res["torch"] = timed(lambda: torch_step(opt, extra))
p_torch = opt.param_groups[0]["params"][0].detach().clone()
params, opt, extra = make()
res["fresh"] = timed(lambda: fresh_step(opt, extra))
p_fresh = opt.param_groups[0]["params"][0].detach().clone()
params, opt, extra = make()
oa = OptimizerArena(opt, extra=extra, headroom=1.3)
res["arena"] = timed(lambda: arena_step(oa))
p_arena = oa.params()["xyz"].detach().clone()
assert torch.equal(p_torch, p_fresh) and torch.equal(p_torch, p_arena), "the three ways must leave identical tensors"
print(f"P = {P0}, 22 per-Gaussian tensors (6 parameters, 12 moments, 4 bookkeeping), 7 repetitions of one densify-and-prune surgery; "
      f"arena buffers allocated: {oa.arena.allocations}")
for k, (med, mn) in res.items():
    print(f"  {k:6s} median {med:7.2f} ms   min {mn:7.2f} ms")
