"""One structure-of-arrays ARENA for everything a Gaussian model keeps per Gaussian.

The reference re-allocates on every densification step: `_prune_optimizer` (gaussiansplatting/scene/gaussian_model.py:
568-591) indexes six parameters and their twelve Adam moments with a boolean mask -- 18 new tensors --, `prune_points`
(:593-607) five bookkeeping tensors more, and `cat_tensors_to_optimizer` (:609-641) builds 18 new tensors with `torch.cat`
(+ 12 `zeros_like`), each wrapped in a NEW `nn.Parameter` whose optimizer state has to be re-keyed.  `densify.py` already
does each of those with one kernel launch, but still into fresh tensors.  Here every per-Gaussian tensor is a view of one
buffer with head-room:

    arena = [ half 0 | half 1 ],  half = [ region(tensor 0) | region(tensor 1) | ... ],  region = capacity x row_bytes

  * PRUNE = one stable compaction (`gsr_compact_plan` + `gsr_compact_apply`) from the live half into the other one, then
    the halves swap: no allocation of parameter size, nothing freed;
  * DENSIFY = the new rows written BEHIND the live rows of the live half (`gsr_append_rows` with no old rows to move):
    the old rows are not touched at all, where `torch.cat` copies all of them;
  * the optimizer's moments ARE arena views; its parameters are new `nn.Parameter` objects wrapping the arena's views after
    every prune / append, exactly as in the reference (re-pointing `param.data` to a view of another shape leaves autograd's
    accumulator of the old shape behind: "invalid gradient ... expected shape"), and their state entries are re-keyed --
    dictionary operations, no device work.
When an append does not fit, the arena grows geometrically (new buffer of `growth` = 1.5 times the capacity, one copy).

Footprint: two halves of `capacity` rows each, i.e. 2 x headroom x the live bytes (3 x at the default head-room of 1.5;
~12 GB for 6 M SH3 Gaussians with their Adam moments), and while a grown buffer is being filled the old one is still alive:
the peak at that moment is (2 + 2 x growth) x capacity rows.  That is the moment memory is tightest, so an allocation
failure there is reported as such (RuntimeError naming the sizes) instead of surfacing as a bare HIP out-of-memory error;
`OptimizerArena(..., headroom=, growth=)` / `RowArena(..., headroom=, growth=)` choose the trade.

Results are bit-identical to `tensor[mask]` / `torch.cat` (the same kernels as densify.compact_rows / append_rows; tests).
No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch

from . import _native

__all__ = ["RowArena", "OptimizerArena"]


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


class RowArena:
    """`tensors`: name -> tensor with leading dimension P (same P, same ROCm device, any dtype / trailing shape).  Their
    contents are copied into the arena once; `arena[name]` is from then on THE tensor (a (P, ...) view)."""

    def __init__(self, tensors: Dict[str, torch.Tensor], capacity: Optional[int] = None, headroom: float = 1.5,
                 growth: float = 1.5):
        if not (headroom >= 1.0 and growth > 1.0):
            raise ValueError("RowArena: need headroom >= 1 and growth > 1")
        self.growth = float(growth)
        if not tensors:
            raise ValueError("RowArena: no tensors")
        first = next(iter(tensors.values()))
        if not first.is_cuda:
            raise RuntimeError("RowArena: tensors must live on the ROCm GPU; there is no CPU fallback")
        self.device = first.device
        self.P = int(first.shape[0])
        self.names: List[str] = list(tensors)
        self.dtypes = {k: t.dtype for k, t in tensors.items()}
        self.tails = {k: tuple(t.shape[1:]) for k, t in tensors.items()}
        self.row_bytes = {}
        for k, t in tensors.items():
            if t.device != self.device or t.dim() < 1 or int(t.shape[0]) != self.P:
                raise RuntimeError("RowArena: every tensor needs the same device and leading dimension")
            rb = t.element_size() * int(torch.Size(t.shape[1:]).numel())
            if rb == 0:
                raise RuntimeError("RowArena: tensors with empty rows are not supported")
            self.row_bytes[k] = rb
        self.capacity = 0
        self.allocations = 0  # buffers ever allocated (1 in the steady state)
        self._buf = None
        self._live = 0
        self._reserve(max(int(capacity or 0), int(self.P * headroom) + 1024))
        for k, t in tensors.items():
            self[k].copy_(t.detach())

    # --- layout
    def _reserve(self, capacity: int) -> None:
        old = None if self._buf is None else {k: self[k] for k in self.names}
        # the new layout is computed aside and installed only once its buffer exists: a caller that catches the
        # allocation failure below keeps a consistent (smaller) arena
        capacity = int(capacity)
        offsets, off = {}, 0
        for k in self.names:
            offsets[k] = off
            off += _align(capacity * self.row_bytes[k])
        try:
            buf = torch.empty(2 * off, dtype=torch.uint8, device=self.device)
        except torch.OutOfMemoryError as ex:
            held = 0 if old is None else sum(int(t.numel()) * t.element_size() for t in old.values())
            raise RuntimeError(f"RowArena: cannot allocate {2 * off / 2**30:.2f} GiB for {capacity} rows in two halves"
                               f" (the {held / 2**30:.2f} GiB of live rows stay allocated until they are copied); construct "
                               "the arena with a larger capacity up front, or with smaller headroom / growth") from ex
        self.capacity, self._offsets, self._half_bytes, self._buf = capacity, offsets, off, buf
        self.allocations += 1
        self._live = 0
        if old is not None:
            for k, t in old.items():
                self[k].copy_(t)

    def _region(self, name: str, half: int, rows: int, row0: int = 0) -> torch.Tensor:
        rb = self.row_bytes[name]
        start = half * self._half_bytes + self._offsets[name] + row0 * rb
        flat = self._buf[start:start + rows * rb]
        return flat.view(self.dtypes[name]).view((rows,) + self.tails[name])

    def __getitem__(self, name: str) -> torch.Tensor:
        return self._region(name, self._live, self.P)

    def views(self) -> Dict[str, torch.Tensor]:
        return {k: self[k] for k in self.names}

    # --- prune
    def compact(self, keep: torch.Tensor) -> int:
        """Every tensor loses the rows where `keep` (P, bool / uint8) is False; survivors keep their order.  -> new P."""
        if keep.dim() != 1 or int(keep.numel()) != self.P or keep.device != self.device:
            raise RuntimeError("RowArena.compact: keep must be a (P,) mask on the arena's device")
        if self.P == 0:
            return 0
        k8 = keep.contiguous().view(torch.uint8) if keep.dtype == torch.bool else keep.to(torch.uint8).contiguous()
        L = _native.lib()
        nbytes = ctypes.c_size_t(0)
        _native.check("gsr_compact_workspace_size", L.gsr_compact_workspace_size(self.P, ctypes.byref(nbytes)))
        work = torch.empty(int(nbytes.value), dtype=torch.uint8, device=self.device)  # (P / 256 words: not parameter-sized)
        kept = ctypes.c_int64(0)
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream
            _native.check("gsr_compact_plan", L.gsr_compact_plan(s, self.P, k8.data_ptr(), work.data_ptr(), ctypes.byref(kept)))
            n = int(kept.value)
            other = self._live ^ 1
            if n > 0:
                for lo in range(0, len(self.names), 32):
                    chunk = self.names[lo:lo + 32]
                    arr = (_native.CompactTensor * len(chunk))()
                    for i, name in enumerate(chunk):
                        arr[i] = _native.CompactTensor(self._region(name, self._live, self.P).data_ptr(),
                                                       self._region(name, other, n).data_ptr(), self.row_bytes[name])
                    _native.check("gsr_compact_apply", L.gsr_compact_apply(s, self.P, k8.data_ptr(), work.data_ptr(), len(chunk), arr))
        self._live ^= 1
        self.P = n
        return n

    # --- densify
    def append(self, extensions: Dict[str, Optional[torch.Tensor]], n: Optional[int] = None) -> int:
        """Every tensor grows by n rows: `extensions[name]` (n rows), or n zero rows for a name that is missing / None.
        The live rows stay where they are.  -> new P."""
        for e in extensions.values():
            if e is not None:
                n = int(e.shape[0]) if n is None else n
                if int(e.shape[0]) != n:
                    raise RuntimeError("RowArena.append: all extensions need the same number of rows")
        if n is None:
            raise RuntimeError("RowArena.append: give n when every extension is None")
        if n == 0:
            return self.P
        if self.P + n > self.capacity:
            self._reserve(max(int(self.growth * self.capacity), self.P + n + 1024))  # (rare: one copy of everything)
        L = _native.lib()
        keep_alive = []
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream
            for lo in range(0, len(self.names), 32):
                chunk = self.names[lo:lo + 32]
                arr = (_native.AppendTensor * len(chunk))()
                for i, name in enumerate(chunk):
                    e = extensions.get(name)
                    if e is not None:
                        if e.device != self.device or e.dtype != self.dtypes[name] or tuple(e.shape[1:]) != self.tails[name]:
                            raise RuntimeError(f"RowArena.append: extension of `{name}` does not match its tensor")
                        e = e.detach().contiguous()
                        keep_alive.append(e)
                    dst = self._region(name, self._live, n, row0=self.P)
                    arr[i] = _native.AppendTensor(None, None if e is None else e.data_ptr(), dst.data_ptr(), self.row_bytes[name])
                _native.check("gsr_append_rows", L.gsr_append_rows(s, 0, n, len(chunk), arr))  # no old rows to move
        self.P += n
        return self.P


class OptimizerArena:
    """A RowArena that holds a Gaussian model's optimizer: each param group's single parameter, its `exp_avg` /
    `exp_avg_sq` (created as zeros if the optimizer has not stepped yet) and any `extra` per-Gaussian tensors (gradient
    accumulators, radii, masks ...).  After `prune` / `append` the optimizer's groups hold new Parameters wrapping the arena's
    views (returned as {group name: Parameter}, like the reference's `_prune_optimizer` / `cat_tensors_to_optimizer`), the
    moments and `self.extra[...]` are the arena's new views."""

    def __init__(self, optimizer: torch.optim.Optimizer, extra: Optional[Dict[str, torch.Tensor]] = None, headroom: float = 1.5,
                 growth: float = 1.5):
        self.optimizer = optimizer
        self.extra_names = list(extra or {})
        tensors: Dict[str, torch.Tensor] = {}
        self.groups = []
        for group in optimizer.param_groups:
            assert len(group["params"]) == 1
            p = group["params"][0]
            st = optimizer.state[p]
            if "exp_avg" not in st:
                st.setdefault("step", torch.tensor(0.0, dtype=torch.float32))
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            name = group["name"]
            self.groups.append((name, p, st))
            tensors[name] = p.detach()
            tensors[name + ".exp_avg"] = st["exp_avg"]
            tensors[name + ".exp_avg_sq"] = st["exp_avg_sq"]
        for k, t in (extra or {}).items():
            tensors["extra." + k] = t
        self.arena = RowArena(tensors, headroom=headroom, growth=growth)
        self.extra: Dict[str, torch.Tensor] = {}
        self._repoint()

    P = property(lambda s: s.arena.P)

    def _repoint(self) -> None:
        groups = []
        for group, (name, p, st) in zip(self.optimizer.param_groups, self.groups):
            new_p = torch.nn.Parameter(self.arena[name].requires_grad_(True))
            st["exp_avg"] = self.arena[name + ".exp_avg"]
            st["exp_avg_sq"] = self.arena[name + ".exp_avg_sq"]
            self.optimizer.state.pop(p, None)
            self.optimizer.state[new_p] = st
            group["params"][0] = new_p
            groups.append((name, new_p, st))
        self.groups = groups
        self.extra = {k: self.arena["extra." + k] for k in self.extra_names}

    def params(self) -> Dict[str, torch.nn.Parameter]:
        return {name: p for name, p, _ in self.groups}

    def prune(self, keep: torch.Tensor) -> Dict[str, torch.nn.Parameter]:
        """`prune_points` / `_prune_optimizer` (gaussian_model.py:568-607): rows where `keep` is False leave every tensor."""
        self.arena.compact(keep)
        self._repoint()
        return self.params()

    def append(self, tensors_dict: Dict[str, torch.Tensor], extra: Optional[Dict[str, torch.Tensor]] = None
               ) -> Dict[str, torch.nn.Parameter]:
        """`cat_tensors_to_optimizer` (gaussian_model.py:609-641): `tensors_dict[group name]` appended to each parameter,
        zero rows to its moments; extras get `extra[name]` or zero rows."""
        ext: Dict[str, Optional[torch.Tensor]] = {name: tensors_dict[name] for name, _, _ in self.groups}
        for k in self.extra_names:
            ext["extra." + k] = None if extra is None else extra.get(k)
        n = int(next(iter(tensors_dict.values())).shape[0])
        self.arena.append(ext, n=n)
        self._repoint()
        return self.params()
