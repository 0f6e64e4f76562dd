"""Multi-view data parallelism for the rasterizer path (SURVEY.md section 8(e) -- new design;
the reference is single-GPU and renders the views of a batch sequentially,
threestudio/systems/GassuianEditor.py:165-207).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The Gaussian
parameters are replicated; a batch of K views is sharded one view per rank; after the local
backward the rasterizer-input gradients of all ranks are summed with ONE all-reduce over a
single flat fp32 bucket, and the per-Gaussian screen radii are combined with one MAX
all-reduce (the reference takes `torch.max` over the views, GassuianEditor.py:175-178).

The bucket is filled without copies: the backward's gradient tensors are *allocated as views
into the bucket* (see `_C.attach_grad_allocator`), so the kernels write straight into the
buffer RCCL reduces.

The SH gradient is 3M of the bucket's 14+3M floats per Gaussian (77 % at M = 16), but per view it is rank one:
dL_dsh[k] = c_k(dir) * dL_dRGB with dir = normalize(mean - camera centre) (backward.cu:44-98).  In the "rgb" exchange
mode (`GradBucket(..., sh_exchange="rgb")`, the default when more than one rank runs) the backward therefore emits
the 3-float colour gradient instead, the ranks ALL-GATHER those (12 B per Gaussian and view) together with their
camera centres, and every rank rebuilds the sum over views with one kernel (`gsr_sh_grad_compose`), views in
ascending rank order -- bit for bit what one process accumulating the views gives, identical on every replica.  The
SUM all-reduce then carries 14 floats per Gaussian: 56 MB + 12 MB per rank instead of 248 MB at 1 M Gaussians.

"Touched rows" (default on top of the "rgb" mode): a view only produces gradients for the Gaussians it blends (the
front layers, ~10 % of the benchmark scene per view), so most of those dense buffers are zeros.  Each rank packs the rows
that are not entirely zero -- index + 14 floats + the 3-float colour gradient = 72 B -- with one stable compaction, the
ranks all-gather the packed rows (padded to the largest count) and every rank adds them per Gaussian, view 0 first
(`gsr_view_messages_accumulate`: one kernel that writes the dense gradients, SH gradient rebuilt from the colour
gradients).  The sums are those of one process accumulating the views one after the other, bit for bit, on every replica
(a ring all-reduce gives no such order), and a rank receives 72 B x (touched rows of all views) instead of
12 B x N x P + the all-reduce's 2 x 56 B x P.  If the views together touch too many rows for that to pay, all ranks
take the dense route (the decision is made from the gathered counts, so it is the same everywhere).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

import contextlib

from . import options as _options
from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C

__all__ = ["GradBucket", "render_view_grads", "allreduce_view_grads", "multiview_step", "multiview_batch_step", "views_of_rank",
           "batch_loss_scale",
           "densify_synchronized", "replicas_identical"]

import os as _os
import threading as _threading

#: GSR_VIEW_PIPELINE=0: multiview_batch_step renders its views one after the other on the launch stream (default: two streams)
_VIEW_PIPELINE = _os.environ.get("GSR_VIEW_PIPELINE", "1") != "0"
#: GSR_VIEW_AHEAD: how many views' forwards the pipelined batch finishes ahead of the backward it issues (0 or 1; one stream
#: per view in flight)
_VIEW_AHEAD = max(0, min(2, int(_os.environ.get("GSR_VIEW_AHEAD", "0"))))

#: The helper streams of this module, ONE set per device and process, shared by every GradBucket: torch's caching allocator
#: keeps a block pool per stream, so streams created per bucket stranded the cached blocks of every bucket that was dropped
#: (bench.py, the tests and every rebuild after densification make new buckets: `reserved` grew by ~700 MiB per bucket at
#: 10^6 Gaussians, profiles/r05_zzz_views8_probe.txt).  Keyed by (device index, role); created on first use.
_DEVICE_STREAMS = {}
_DEVICE_STREAMS_LOCK = _threading.Lock()


def _device_stream(dev, role: str):
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), role)
    st = _DEVICE_STREAMS.get(key)
    if st is None:
        with _DEVICE_STREAMS_LOCK:
            st = _DEVICE_STREAMS.get(key)
            if st is None:
                st = _DEVICE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st

#: GSR_DEBUG_PERSISTENT_ROWS=1: every backward into a `persistent_rows` bucket first checks that rows marked "holds zeros" do
_DEBUG_ROWS = _os.environ.get("GSR_DEBUG_PERSISTENT_ROWS", "0") == "1"

#: bucket layout per Gaussian: name -> number of floats (3M for the SH block is filled in at construction)
_SLOTS = ("means3D", "sh", "scales", "rotations", "means2D", "opacities")


def _pad4(n: int) -> int:
    return (n + 3) & ~3


class GradBucket:
    """Flat fp32 buffer `[means3D 3P | sh 3MP | scales 3P | rotations 4P | means2D 3P | opacities P]`
    = (14 + 3M) * P floats (248 MB at P = 1M, M = 16), every segment START rounded up to a multiple of 4 floats:
    the kernels use dwordx4 accesses on (P,4) / (P,M,3) rows, so a segment must begin on a 16-byte boundary whatever
    P is (P is arbitrary after a densification or a prune; the padding words stay zero and travel with the all-reduce).
    (The blend backward's accumulator table -- 16 floats per Gaussian, include/gsr.h GSR_ACC_* -- is workspace of the binding,
    not of the bucket: means2D / opacities are copied out of it by K8+K9.)"""

    def __init__(self, P: int, M: int, device, sh_exchange: str = "auto", sparse_rows: bool = False,
                 persistent_rows: bool = False):
        self.P, self.M = int(P), int(M)
        #: `persistent_rows`: the bucket's gradient tensors live across iterations, and a view leaves nine rows of ten zero.
        #: With `row_state` (uint8 (P,), 1 = the row may hold anything) next to them the backward rewrites a zero row only if
        #: it does not already hold the zeros of an earlier backward (gsr_preprocess_backward_rows): K8+K9 72 -> 54 us at 1 M
        #: Gaussians, 401 -> 244 us at 6 M.  Every row is valid and correct after every backward, as without the option.
        #: Whatever else writes into the bucket's gradients must call invalidate_rows() (the exchange routes do).
        self.row_state = torch.ones(int(P), dtype=torch.uint8, device=device) if persistent_rows else None
        self._handed = set()  # names of the gradients handed out as this bucket's tensors in the current backward
        #: `sparse_rows`: the touched-rows exchange writes the summed gradients only for the Gaussians some view touched and
        #: marks them in `row_valid` (uint8 (P,)); the other rows of the gradient tensors are stale or uninitialised and count
        #: as zeros -- hand `row_valid` to the consumer (FusedMaskedAdam.set_grad_valid).  After any other route every row is
        #: valid (row_valid is all ones).
        self.row_valid = torch.ones(int(P), dtype=torch.uint8, device=device) if sparse_rows else None
        if sh_exchange == "auto":
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            sh_exchange = "rgb" if (multi and M > 0) else "direct"
        if sh_exchange not in ("direct", "rgb"):
            raise ValueError("sh_exchange must be 'auto', 'direct' or 'rgb'")
        self.sh_exchange = sh_exchange
        shapes = {"means3D": (P, 3), "sh": (P, M, 3), "opacities": (P, 1), "scales": (P, 3), "rotations": (P, 4),
                  "means2D": (P, 3)}
        slots = _SLOTS if sh_exchange == "direct" else tuple(s for s in _SLOTS if s != "sh")
        offs, off = {}, 0
        for name in slots:
            offs[name] = off
            off = _pad4(off + int(torch.Size(shapes[name]).numel()))
        self._buf = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat = self._buf[:off]
        if self.flat.data_ptr() % 16 != 0:  # (torch's allocators hand out >= 256-byte alignment; be explicit anyway)
            raise RuntimeError("GradBucket: the flat buffer is not 16-byte aligned")
        self.views: Dict[str, torch.Tensor] = {}
        #: "rgb" mode: this rank's clamp-masked colour gradient (P,3), written by the backward
        self.rgb = torch.zeros((P, 3), dtype=torch.float32, device=device) if sh_exchange == "rgb" else None
        self.sh_degree = None  # active SH degree of the last backward ("rgb" mode needs it to rebuild dL_dsh)
        #: called with the blend backward's row mask (uint8 (P,)) between K7 and K8+K9 (multiview_step sets it: the
        #: touched-row counts are exchanged from there, underneath K8+K9)
        self.on_blend_done = None
        self._pending_counts = None
        self._side_stream = None
        self._view_streams = None  # multiview_batch_step: the two streams successive views alternate on
        self._cap_hint = None  # message capacity the next touched-rows exchange speculates on (rows)
        self.last_route = None  # what the last multiview_step's exchange did: "local" | "rows" | "sparse" | "dense"
        self.last_counts = None  # touched rows per view, as gathered by the last touched-rows exchange
        self.last_exchange = None  # multiview_batch_step: views, rows per message, bytes sent / received by this rank
        for name in slots:
            cnt = int(torch.Size(shapes[name]).numel())
            self.views[name] = self.flat[offs[name]:offs[name] + cnt].view(shapes[name])

    def check_persistent_rows(self) -> None:
        """Debug (GSR_DEBUG_PERSISTENT_ROWS=1 runs it in front of every backward): the contract of `persistent_rows` is
        enforced by the CALLER -- whoever writes into the bucket's gradients outside the backward / the exchange (a
        regulariser added in place, a checkpointed gradient loaded, a custom exchange) must call invalidate_rows().  A row
        with row_state == 0 that is not all-zero means somebody did not: the next backward would leave it as it is."""
        if self.row_state is None:
            return
        stale = self.row_state == 0
        for name in ("means3D", "scales", "rotations", "sh"):
            v = self.views.get(name) if not (name == "sh" and self.sh_exchange == "rgb") else self.rgb
            if v is None or int(v.shape[0]) != self.P:
                continue
            bad = stale & (v.reshape(self.P, -1) != 0).any(dim=1)
            if bool(bad.any()):
                raise RuntimeError(f"GradBucket(persistent_rows=True): {int(bad.sum())} rows of `{name}` are marked as holding "
                                   "zeros but do not -- something wrote into the bucket's gradients without invalidate_rows()")

    def invalidate_rows(self) -> None:
        """Something other than the backward has written to the gradient tensors: every row is rewritten next time."""
        if self.row_state is not None:
            self.row_state.fill_(1)

    def allocator(self, name: str, shape: Tuple[int, ...], zero: bool):
        if name == "row_state":
            # asked last.  Only if every gradient the state stands for was answered with this bucket's own tensor in THIS
            # backward: a row that is not rewritten must be a row of a tensor that lives across iterations
            own = {"means2D", "opacities", "means3D", "scales", "rotations"} <= self._handed and ({"sh", "sh_rgb"} & self._handed)
            ok = self.row_state is not None and own and tuple(shape) == (self.P,)
            if ok and _DEBUG_ROWS:
                self.check_persistent_rows()
            return self.row_state if ok else None
        if name == "after_blend_backward":  # a notification, not an allocation (`shape` = K7's row mask, uint8 (P,))
            if self.on_blend_done is not None:
                self.on_blend_done(shape)
            return None
        if name == "sh_rgb":  # "rgb" exchange mode: ask the backward for dL_dRGB instead of dL_dsh
            ok = self.rgb is not None and tuple(shape) == (self.P, 3)
            if ok:
                self._handed.add("sh_rgb")
            return self.rgb if ok else None
        if name == "acc_rows":  # (the first request of a backward) the blend backward's accumulator table: the binding's own
            self._handed = set()  # -- kept across backwards and left zero by K8+K9 where nothing reads it in between
            return None
        v = self.views.get(name)
        if v is None or tuple(v.shape) != tuple(shape):
            return None
        if zero:
            v.zero_()
        self._handed.add(name)
        return v

    def attach(self, color: torch.Tensor) -> None:
        """The backward of the render that produced `color` writes its gradients into this bucket.  (The allocator
        rides on that render's autograd node: no module or thread state, whichever thread runs the backward.)"""
        _C.attach_grad_allocator(color, self.allocator)

    def grads(self) -> Dict[str, torch.Tensor]:
        return dict(self.views)

    def flat_views(self) -> Dict[str, torch.Tensor]:
        """The segments that live in `flat` (everything except a rebuilt SH gradient in "rgb" mode)."""
        return {k: v for k, v in self.views.items() if self.sh_exchange == "direct" or k != "sh"}


#: GSR_VIEW_AUTOGRAD=1: one view's forward / backward go through the L1 autograd.Function and torch.autograd.grad, as up to
#: round 4, instead of calling the L0 entry points directly (same native calls, same numbers; for A/B measurements)
_VIEW_AUTOGRAD = _os.environ.get("GSR_VIEW_AUTOGRAD", "0") == "1"


def _view_forward(settings, means3D, opacities, shs, scales, rotations):
    """Forward of one view -> (color, radii, depth, saved): `saved` is what `_view_backward` needs.

    The L0 entry points are called DIRECTLY (_C.rasterize_gaussians / _C.rasterize_gaussians_backward with the argument order
    of the reference's rasterize_points.h:17-60, exactly what the L1 autograd.Function passes them): going through
    autograd.Function.apply and torch.autograd.grad ran every view's backward on the engine's device thread and cost the
    launch thread 280 us per view at 10^6 Gaussians -- the two-stream batch was bound by that, not by the GPU
    (profiles/r05_c_view_pipelining.md).  GSR_VIEW_AUTOGRAD=1 restores the L1 route."""
    if _VIEW_AUTOGRAD or settings.debug:  # (debug renders keep the L1 route: it dumps the inputs of a failing call)
        leaves = [t.detach().requires_grad_(True) for t in (means3D, shs, opacities, scales, rotations)]
        m3, sh, op, sc, rot = leaves
        # the screen-space dummy only carries a gradient; its values are never read (forward.cu ignores means2D), so it is
        # not zero-filled here (the reference's render() does: gaussian_renderer/__init__.py:60-69)
        m2 = torch.empty_like(m3).requires_grad_(True)
        color, radii, depth = GaussianRasterizer(settings)(m3, m2, op, shs=sh, scales=sc, rotations=rot)
        return color, radii, depth, leaves + [m2]
    rs = settings
    flags = _options.current_flags()
    m3, sh, op, sc, rot = (t.detach() for t in (means3D, shs, opacities, scales, rotations))
    absent = m3.new_empty(0)  # "not provided": colors_precomp, cov3D_precomp (diff_gaussian_rasterization/__init__.py:_absent)
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        rs.bg, m3, absent, op, sc, rot, rs.scale_modifier, absent, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
        rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, flags=flags)
    return color, radii, depth, (flags, R, geom, binning, img, m3, sh, sc, rot, absent)


def _view_forward_begin(settings, means3D, opacities, shs, scales, rotations):
    """`_view_forward` in two halves (the pipelined view batch): K1 of the view is enqueued on the current stream and the call
    returns without waiting for its counts (_C.rasterize_gaussians_begin).  The L1 route has no such split: it is deferred to
    `_view_forward_finish` as a whole."""
    if _VIEW_AUTOGRAD or settings.debug or means3D.size(0) == 0:
        return ("whole", settings, means3D, opacities, shs, scales, rotations)
    rs = settings
    flags = _options.current_flags()
    m3, sh, op, sc, rot = (t.detach() for t in (means3D, shs, opacities, scales, rotations))
    absent = m3.new_empty(0)
    pending = _C.rasterize_gaussians_begin(
        rs.bg, m3, absent, op, sc, rot, rs.scale_modifier, absent, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
        rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, flags=flags)
    return ("split", pending, (flags, m3, sh, sc, rot, absent))


def _view_forward_finish(begun):
    """-> what `_view_forward` returns, on the stream `_view_forward_begin` ran on."""
    if begun[0] == "whole":
        return _view_forward(*begun[1:])
    _, pending, (flags, m3, sh, sc, rot, absent) = begun
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians_finish(pending)
    return color, radii, depth, (flags, R, geom, binning, img, m3, sh, sc, rot, absent)


def _view_backward(settings, state, dL_dcolor, bucket):
    """Backward of the view `_view_forward` rendered, with `dL_dcolor` as the pixel gradient -> the six rasterizer-input
    gradients (views of `bucket` when one is given)."""
    color, radii, depth, saved = state
    names = ("means3D", "sh", "opacities", "scales", "rotations", "means2D")
    if len(saved) == 6:  # (the L1 route's leaves, see _view_forward)
        m3, sh, op, sc, rot, m2 = saved
        if bucket is not None:
            bucket.attach(color)
        # ("rgb" exchange mode: the SH gradient comes back as None here and is rebuilt by the exchange)
        g = torch.autograd.grad([color], [m3, sh, op, sc, rot, m2], grad_outputs=[dL_dcolor], allow_unused=True)
    else:
        rs = settings
        flags, R, geom, binning, img, m3, sh, sc, rot, absent = saved
        g_m2, _, g_op, g_m3, _, g_sh, g_sc, g_rot = _C.rasterize_gaussians_backward(
            rs.bg, m3, radii, absent, sc, rot, rs.scale_modifier, absent, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, dL_dcolor, sh, rs.sh_degree, rs.campos, geom, R, binning, img, rs.debug, flags=flags,
            grad_allocator=None if bucket is None else bucket.allocator)
        g = (g_m3, g_sh, g_op, g_sc, g_rot, g_m2)
    grads = dict(zip(names, g))
    if bucket is not None:
        # Every gradient must now BE its bucket segment (all segments are 16-byte aligned, so the backward can always
        # write there).  Exchanging a bucket the backward did not fill would silently reduce stale data: refuse.
        for name, t in grads.items():
            v = bucket.views.get(name)
            if t is None or v is None or (name == "sh" and bucket.sh_exchange == "rgb"):
                continue
            if t.data_ptr() != v.data_ptr():
                raise RuntimeError(f"render_view_grads: the gradient of `{name}` was not written into the bucket")
    if bucket is not None and bucket.sh_exchange == "rgb":
        bucket.sh_degree = int(settings.sh_degree)
        bucket.means3D_ref = m3.detach()
        bucket.campos = settings.campos.detach().reshape(3).to(torch.float32)
        grads["sh"] = None
    return grads


def render_view_grads(settings: GaussianRasterizationSettings, means3D, opacities, shs, scales, rotations,
                      dL_dcolor: torch.Tensor, bucket: Optional[GradBucket] = None, after_forward=None):
    """Forward + backward of ONE view (the L0 entry points the drop-in L1 API calls) with `dL_dcolor` as the
    pixel gradient.  Returns (color, radii, depth, grads) where grads has the six
    rasterizer-input gradients (views of `bucket` when one is given).  `after_forward(radii)` is called between the
    forward and the backward (multiview_step starts the radii's MAX all-reduce there, so that it overlaps the backward)."""
    state = _view_forward(settings, means3D, opacities, shs, scales, rotations)
    color, radii, depth, _ = state
    if after_forward is not None:
        after_forward(radii)
    grads = _view_backward(settings, state, dL_dcolor, bucket)
    return color.detach(), radii, depth.detach(), grads


def _touched_rows(bucket: GradBucket) -> torch.Tensor:
    """(P,) bool: Gaussians with at least one non-zero gradient entry.  A Gaussian that no pixel of this
    rank's view blended has an exactly zero row in every segment (the backward writes zeros there)."""
    P = bucket.P
    t = torch.zeros(P, dtype=torch.bool, device=bucket.flat.device)
    for name, v in bucket.flat_views().items():
        # dL_dsh[k] = basis_k(dir) * dL_dRGB with basis_0 = SH_C0 != 0 (backward.cu:47-48): the whole SH row is zero
        # iff its coefficient-0 triple is, so the 12(M-1) other bytes per Gaussian need not be scanned.
        rows = v[:, 0, :] if name == "sh" and v.shape[1] > 0 else v.reshape(P, -1)
        t |= (rows != 0).any(dim=1)
    return t


#: bytes a touched row costs in a message: the row number + means3D 3, scales 3, rotations 4, means2D 3, opacities 1, rgb 3
_ROW_BYTES = 4 * 18
_ROW_SEGS = ("means3D", "scales", "rotations", "means2D", "opacities")


def _start_counts_exchange(bucket: GradBucket, touched, group, n: int):
    """Called between K7 and K8+K9 of this rank's backward with K7's row mask (uint8 (P,): the Gaussians whose accumulator
    rows it adds to): plans the message from it, and exchanges the ranks' row counts -- on a side stream, so that the plan kernels, the tiny all-gather AND
    the host's wait for its result all run while K8+K9 (~100 us) occupies the launch stream.  The step's one host
    synchronisation then costs the GPU nothing: when K8+K9 retires, pack / all-gather / accumulate are already queued."""
    dev = touched.device
    if dev.type != "cuda":  # gloo on CPU (tests): nothing to overlap
        plan, mine = _C.view_message_plan_blend(touched)
        gathered = [torch.empty_like(mine) for _ in range(n)]
        dist.all_gather(gathered, mine, group=group)
        return plan, torch.cat(gathered), None
    main = torch.cuda.current_stream(dev)
    side = bucket._side_stream = _device_stream(dev, "side")
    ready = torch.cuda.Event()
    ready.record(main)  # K7 is enqueued in front of this
    with torch.cuda.stream(side):
        side.wait_event(ready)
        touched.record_stream(side)  # (allocated under the launch stream by the backward, read here)
        plan, mine = _C.view_message_plan_blend(touched)
        gathered = torch.empty(n, dtype=torch.int64, device=dev)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(gathered, mine.contiguous(), group=group)
        else:
            parts = [torch.empty_like(mine) for _ in range(n)]
            dist.all_gather(parts, mine, group=group)
            gathered.copy_(torch.cat(parts))
        planned = torch.cuda.Event()
        planned.record(side)  # the plan itself (mask + row offsets): all the pack kernel waits for
        host = torch.empty(n, dtype=torch.int64, pin_memory=True)
        host.copy_(gathered, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
    for t in plan[:2]:  # allocated under the side stream, consumed by the pack kernel on the launch stream
        if t is not None:
            t.record_stream(main)
    return plan, host, (planned, done)


def _exchange_touched_rows(bucket: GradBucket, group, n: int, force: bool) -> Optional[str]:
    """The touched-rows exchange of the "rgb" mode (module docstring).  Returns "rows", or None when the dense route is
    cheaper (nothing has been modified then)."""
    P, dev = bucket.P, bucket.flat.device
    grads5 = [bucket.views[name] for name in _ROW_SEGS]
    prepared, bucket._pending_counts = bucket._pending_counts, None

    def send_messages(plan, cap):
        words = _C.view_message_words(P, cap)
        send = torch.empty(words, dtype=torch.float32, device=dev)
        _C.view_message_pack(plan, grads5, bucket.rgb, bucket.campos, cap, send)
        recv = torch.empty((n, words), dtype=torch.float32, device=dev)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(recv, send, group=group)  # straight into the rows of recv
        else:
            dist.all_gather(list(recv.unbind(0)), send, group=group)
        return recv

    recv = cap = None
    if prepared is not None:  # the counts were exchanged underneath K8+K9 (_start_counts_exchange)
        plan, host, events = prepared
        if events is not None:
            torch.cuda.current_stream(dev).wait_event(events[0])  # (a GPU-side wait: the pack kernel reads the plan)
        if bucket._cap_hint is not None:
            # SPECULATE on the message size (the last step's largest count + 25 %): pack and all-gather are enqueued
            # without waiting for this step's counts, so the host's wait below runs underneath them (and underneath
            # the collective's wire time); nothing the messages feed -- the accumulate kernel -- is enqueued before
            # the counts have been checked, and a message that turns out too small is simply sent again.
            cap = bucket._cap_hint
            recv = send_messages(plan, cap)
        if events is not None:
            events[1].synchronize()  # the host waits for the SIDE stream only
        counts = [int(c) for c in host.tolist()]
    else:
        plan, mine = _C.view_message_plan(grads5, bucket.rgb, readback=False)  # the count stays on the device ...
        gathered = [torch.empty_like(mine) for _ in range(n)]
        dist.all_gather(gathered, mine, group=group)
        counts = [int(c) for c in torch.cat(gathered).tolist()]  # ... the step's one host sync; identical on every rank
    bucket.last_counts = counts
    bucket._cap_hint = (int(1.25 * max(max(counts), 1)) + 1023) // 1024 * 1024  # identical on every rank
    dense_bytes = 12 * n * P + 2 * 56 * P  # what a rank receives on the dense route (rgb all-gather + ring all-reduce)
    if not force and _ROW_BYTES * sum(counts) > 0.6 * dense_bytes:
        return None  # (a speculative send is dropped: nothing was modified)
    if recv is None or max(counts) > cap:
        cap = max(max(counts), 1)
        recv = send_messages(plan, cap)
    bucket.last_exchange = {"views": n, "views_local": 1, "message_rows": int(cap), "bytes_sent": int(recv.shape[1]) * 4,
                            "bytes_received": int(recv.numel()) * 4}
    sh = torch.empty((P, bucket.M, 3), dtype=torch.float32, device=dev)
    # one kernel: per Gaussian, the views' rows added in ascending view order (zeros where no view touched it)
    # (`sparse_rows` buckets: only for the Gaussians some view touched, marked in bucket.row_valid)
    _C.view_messages_accumulate(recv, P, cap, bucket.sh_degree, bucket.M, bucket.means3D_ref, grads5 + [sh],
                                row_valid=bucket.row_valid)
    bucket.views["sh"] = sh
    return "rows"


def allreduce_view_grads(bucket: GradBucket, radii: Optional[torch.Tensor] = None, group=None, sparse="auto",
                         sparse_threshold: float = 0.5, rows="auto", force_exchange: bool = False):
    """The exchange step of an iteration: SUM over ranks of the gradient bucket, MAX over ranks of the screen
    radii.  No-op in a single-process run.

    A view only produces gradients for the Gaussians it actually blends (the front layers: ~10 % of the
    benchmark scene per view), so summing the dense (14+3M)*P buffer mostly moves zeros over xGMI -- 248 MB per
    step at 1 M Gaussians, more than the step's compute time at 8 GPUs.  With `sparse` enabled the ranks first
    MAX-all-reduce a one-byte-per-Gaussian "touched" mask (P bytes), and if the union is at most
    `sparse_threshold` of the scene only the union rows are packed, summed with ONE all-reduce and scattered
    back; otherwise the dense bucket is reduced.  Every rank takes the same branch (the mask is reduced), and
    rows outside the union are zero on every rank, so the result equals the dense all-reduce.

    In the "rgb" mode of the bucket `rows` ("auto" | True | False) selects the touched-rows exchange described in the
    module docstring; "auto" uses it unless the gathered row counts say the dense route moves fewer bytes.  Returns the
    route taken: "local", "rows", "sparse" or "dense".  `force_exchange` runs the collectives even in a group of one
    rank (tests: every route on the RCCL backend of a single-GPU box)."""
    single = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1
    if single and not (force_exchange and dist.is_available() and dist.is_initialized()):
        if bucket.row_valid is not None:
            bucket.row_valid.fill_(1)
        if bucket.sh_exchange == "rgb":  # a single view: the "sum" has one term
            bucket.views["sh"] = _C.sh_grad_compose(bucket.means3D_ref, bucket.campos.view(1, 3), bucket.rgb.view(1, bucket.P, 3),
                                                     bucket.sh_degree, bucket.M)
        return "local"
    if bucket.sh_exchange == "rgb" and rows in ("auto", True):
        if _exchange_touched_rows(bucket, group, dist.get_world_size(group), force=rows is True) is not None:
            if bucket.row_state is not None:
                # persistent rows: the accumulate kernel wrote sums into the rows it marks valid (all rows without sparse_rows);
                # a row it did not write still holds the backward's zeros
                if bucket.row_valid is not None:
                    bucket.row_state.copy_(bucket.row_valid)
                else:
                    bucket.invalidate_rows()
            if radii is not None:
                dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
            return "rows"
    bucket.invalidate_rows()  # every route below writes sums into all rows of the bucket's gradient tensors
    if bucket.row_valid is not None:
        bucket.row_valid.fill_(1)  # every other route writes every row
    if bucket.sh_exchange == "rgb":
        # colour gradients + camera centres of all views, then the SH gradient of the batch, rebuilt locally
        n = dist.get_world_size(group)
        rgb_all = torch.empty((n, bucket.P, 3), dtype=torch.float32, device=bucket.rgb.device)
        cam_all = torch.empty((n, 3), dtype=torch.float32, device=bucket.rgb.device)
        dist.all_gather(list(rgb_all.unbind(0)), bucket.rgb.contiguous(), group=group)
        dist.all_gather(list(cam_all.unbind(0)), bucket.campos.contiguous(), group=group)
        bucket.views["sh"] = _C.sh_grad_compose(bucket.means3D_ref, cam_all, rgb_all, bucket.sh_degree, bucket.M)
    if radii is not None:
        dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
    mode = "dense"
    if sparse == "auto":
        # worth its extra collective, host sync and pack / unpack kernels only for the 248 MB bucket that carries the SH
        # gradient; the 56 MB bucket of the colour-gradient exchange is reduced densely
        sparse = bucket.sh_exchange == "direct"
    if sparse:
        P = bucket.P
        mask = _touched_rows(bucket).to(torch.uint8)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
        idx = mask.nonzero(as_tuple=False).view(-1)  # one host sync; identical on every rank
        if idx.numel() <= sparse_threshold * P:
            segs = [v.reshape(P, -1) for v in bucket.flat_views().values()]
            packed = torch.cat([sg.index_select(0, idx) for sg in segs], dim=1)  # (U, 14+3M)
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
            off = 0
            for sg in segs:
                w = sg.shape[1]
                sg.index_copy_(0, idx, packed[:, off:off + w])
                off += w
            mode = "sparse"
    if mode == "dense":
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=group)
    return mode


def _mark(marks, name, dev):
    """`marks` (a dict, optional argument of the step functions): an event on the launch stream at the named point, so that a
    caller (bench.py) can tell the local render time from the exchange time without adding any synchronisation."""
    if marks is not None and dev.type == "cuda":
        ev = marks.get(name)
        if not isinstance(ev, torch.cuda.Event):  # (a caller that times many steps hands the events in: creating one can
            ev = torch.cuda.Event(enable_timing=True)  # cost the launch thread milliseconds when the runtime grows its pool)
            marks[name] = ev
        ev.record(torch.cuda.current_stream(dev))


def multiview_step(settings: GaussianRasterizationSettings, params: Dict[str, torch.Tensor], dL_dcolor: torch.Tensor,
                   bucket: GradBucket, group=None, rows="auto", sparse="auto", force_exchange: bool = False, marks=None):
    """One data-parallel iteration for this rank's view: forward, backward, gradient all-reduce.
    `params`: xyz, opacity, features, scaling, rotation (activated, as the rasterizer consumes them).
    After the call `bucket.views[...]` hold the batch-summed gradients on every rank and
    `radii` the batch-max radii."""
    pending = []

    def start_radii(radii):  # known after the forward: the collective runs while the backward computes
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_exchange):
            batch_max = radii.clone()  # (the backward still needs this view's own radii)
            pending.append((batch_max, dist.all_reduce(batch_max, op=dist.ReduceOp.MAX, group=group, async_op=True)))

    exchanging = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_exchange)
    if exchanging and bucket.sh_exchange == "rgb" and rows in ("auto", True):
        n = dist.get_world_size(group)

        def start_counts(touched):  # between K7 and K8+K9 of this rank's backward
            bucket._pending_counts = _start_counts_exchange(bucket, touched, group, n)

        bucket.on_blend_done = start_counts
    try:
        color, radii, depth, grads = render_view_grads(settings, params["xyz"], params["opacity"], params["features"],
                                                       params["scaling"], params["rotation"], dL_dcolor, bucket,
                                                       after_forward=start_radii)
    except BaseException:
        # a backward / exchange that raised after start_counts ran must not leave its plan, host buffer and events behind:
        # the next step's exchange would pair the stale plan with new gradients
        bucket._pending_counts = None
        bucket._cap_hint = None
        raise
    finally:
        bucket.on_blend_done = None
    _mark(marks, "local_done", bucket.flat.device)
    bucket.last_route = allreduce_view_grads(bucket, None, group, rows=rows, sparse=sparse, force_exchange=force_exchange)
    for batch_max, work in pending:
        work.wait()
        radii = batch_max
    _mark(marks, "step_done", bucket.flat.device)
    return color, radii, depth, grads


# ----------------------------------------------------------------------------------------------------------------------
# K views on N ranks (round 4): the reference's batch loop, sharded
# ----------------------------------------------------------------------------------------------------------------------
def views_of_rank(num_views: int, world_size: int, rank: int) -> range:
    """The contiguous block of global view indices rank `rank` renders when a batch of `num_views` views is dealt to
    `world_size` ranks: views [r K / N, (r + 1) K / N) -- ranks in ascending order hold ascending views, so the ranks'
    messages, concatenated in rank order by the all-gather, are the views in ascending global order."""
    K, N, r = int(num_views), int(world_size), int(rank)
    if N < 1 or not 0 <= r < N or K < N or K % N != 0:
        raise ValueError("views_of_rank: need world_size >= 1, 0 <= rank < world_size and num_views a multiple of world_size")
    per = K // N
    return range(r * per, (r + 1) * per)


def _relayout_message(src: torch.Tensor, P: int, cap_src: int, dst: torch.Tensor, cap_dst: int) -> None:
    """Copy a packed view message laid out for `cap_src` rows into the layout for `cap_dst` >= its count (include/gsr.h:
    header of 4 + ceil(P / 1024) words, then SoA segments of cap rows: index 1, means3D 3, scales 3, rotations 4, means2D 3,
    opacities 1, rgb 3 words per row).  Rows beyond the count are padding on both sides."""
    head = 4 + (int(P) + 1023) // 1024
    dst[:head].copy_(src[:head])
    n = min(int(cap_src), int(cap_dst))
    off = 0
    for w in (1, 3, 3, 4, 3, 1, 3):
        dst[head + off * cap_dst: head + off * cap_dst + w * n].copy_(src[head + off * cap_src: head + off * cap_src + w * n])
        off += w


def multiview_batch_step(settings_list, params: Dict[str, torch.Tensor], dL_dcolor_list, bucket: GradBucket, group=None,
                         marks=None):
    """One iteration over a batch of K views of which THIS rank renders `settings_list` (its block of the batch, in ascending
    global view order: `views_of_rank`) -- the reference's loop over `batch["camera"]` (threestudio/systems/GassuianEditor.py:
    165-207: every view rendered, the gradients accumulated by autograd view after view, radii combined with torch.max
    :175-178), sharded over the ranks of `group`.

    Every local view is rendered forward + backward into the bucket and, before the next view overwrites it, packed into a
    touched-rows message; the ranks all-gather their messages ONCE per step (rank order = global view order) and one kernel
    adds all K messages per Gaussian in ascending view order (`gsr_view_messages_accumulate`).  The sums are therefore those
    of a single process accumulating views 0 .. K - 1 one after the other, bit for bit, whatever N is -- including N = 1,
    where nothing is communicated at all -- and identical on every replica.

    The host never waits inside the loop over the views: the messages are sized by speculation (1.5 x the largest count of
    the previous step), the plan of a view's touched rows runs on a side stream underneath its K8+K9, the pack kernel follows
    on the launch stream behind a GPU-side wait, and the true counts -- every message's header carries its own -- are read
    ONCE per step, after the all-gather has been enqueued and before the accumulate kernel.  A first step, and a step in
    which some view outgrew the speculation (the same decision on every rank: it is taken from the gathered counts), runs
    the exact form instead: one host read per view underneath that view's K8+K9, messages of exactly the needed size.

    The bucket must be in the "rgb" exchange mode (the messages carry the colour gradient; the SH gradient is rebuilt).
    Returns (colors, radii, depths, grads): lists for this rank's views, the batch-max radii, and the batch-summed gradients
    (`bucket.views`).  `bucket.last_counts` holds the touched rows of all K views, `bucket.last_exchange` what was sent."""
    k_local = len(settings_list)
    if k_local < 1 or len(dL_dcolor_list) != k_local:
        raise ValueError("multiview_batch_step: one pixel gradient per local view, at least one view")
    if bucket.sh_exchange != "rgb":
        raise RuntimeError('multiview_batch_step: construct the bucket with GradBucket(..., sh_exchange="rgb")')
    out = None
    if bucket._cap_hint is not None:
        out = _batch_step(settings_list, params, dL_dcolor_list, bucket, group, marks, int(bucket._cap_hint))
    if out is None:  # no speculation yet, or it was too small: the exact form
        out = _batch_step(settings_list, params, dL_dcolor_list, bucket, group, marks, None)
    return out


def _batch_step(settings_list, params, dL_dcolor_list, bucket, group, marks, spec_cap):
    k_local = len(settings_list)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    n = dist.get_world_size(group) if multi else 1
    P, dev = bucket.P, bucket.flat.device
    grads5 = [bucket.views[name] for name in _ROW_SEGS]
    on_gpu = dev.type == "cuda"
    speculate = spec_cap is not None
    state = {}

    def after_blend(touched):  # between K7 and K8+K9 of a local view: mask + count on a side stream, underneath K8+K9
        if not on_gpu:
            plan, mine = _C.view_message_plan_blend(touched)
            state.update(plan=plan, count=mine, planned=None, done=None)
            return
        main = torch.cuda.current_stream(dev)
        side = bucket._side_stream = _device_stream(dev, "side")
        ready = torch.cuda.Event()
        ready.record(main)
        free = state.pop("bucket_free", None)
        if free is not None:  # (pipelined batch) K8+K9 is the first kernel of this view that writes the bucket: only now
            main.wait_event(free)  # must the previous view's message be packed -- this view's K7 ran underneath that
        with torch.cuda.stream(side):
            side.wait_event(ready)
            touched.record_stream(side)  # (allocated under the view's stream by the backward, read here)
            plan, mine = _C.view_message_plan_blend(touched)
            planned = torch.cuda.Event()
            planned.record(side)
            host, done = mine, None
            if not speculate:
                host = torch.empty(1, dtype=torch.int64, pin_memory=True)
                host.copy_(mine, non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
        for t in plan[:2]:
            if t is not None:
                t.record_stream(main)
        state.update(plan=plan, count=host, planned=planned, done=done)

    cap = spec_cap
    words = _C.view_message_words(P, cap) if speculate else 0
    send = torch.empty((k_local, words), dtype=torch.float32, device=dev) if speculate else None
    tight, counts, colors, depths = [], [], [], []
    radii_max = None
    bucket.on_blend_done = after_blend
    # Two-stream VIEW PIPELINING (speculative form, several views on this rank): the views alternate between two streams, so
    # that view v + 1's forward -- K1, the depth sort, the binning: bandwidth and latency-bound kernels -- runs underneath view
    # v's backward, whose K7 is bound by VALU issue.  The forward is issued in two halves (_view_forward_begin / _finish): the
    # launch thread never waits for a view's counts.  The backward of view v + 1 waits (on the GPU) until view v's message has
    # been packed: the bucket is the one buffer all backwards write.  Results do not depend on the schedule.  Measured on one
    # MI355X: profiles/r04_c_pipelining.md (0.61 -> 0.52 ms per view), profiles/r06_i_view_pipelining.md (the split forward).
    # GSR_VIEW_PIPELINE=0 turns it off.
    pipeline = speculate and on_gpu and k_local > 1 and _VIEW_PIPELINE
    # ... and the library, not its caller, says so to the rasterizer: the renders of a pipelined batch carry
    # GSR_FLAG_SHARED_SIMDS (2 persistent blend waves per SIMD; until round 4 bench.py set GSR_BLEND_WAVES_PER_SIMD=2 for the
    # whole process, so any other caller of this function ran the pipeline at 4)
    shared = _options.override(_options.current_flags() | _options.FLAG_SHARED_SIMDS) if pipeline else contextlib.nullcontext()
    try:
        if pipeline:
            with shared:
                main = torch.cuda.current_stream(dev)
                # `ahead` forwards are FINISHED (binning, K6) before the backward in front of them is issued, one more is
                # begun (K1): with ahead = 1 view v's K7 runs over view v + 1's binning and K6 and view v + 2's K1, and view
                # v + 1's K7 can follow it at once.  One stream per view in flight.
                ahead = _VIEW_AHEAD
                ns = 2 + ahead
                S = bucket._view_streams = tuple(_device_stream(dev, f"view{i}") for i in range(ns))
                for st_ in S:
                    st_.wait_stream(main)
                args = (params["xyz"], params["opacity"], params["features"], params["scaling"], params["rotation"])
                begun, fstates = {}, {}

                def begin(v):
                    with torch.cuda.stream(S[v % ns]):
                        begun[v] = _view_forward_begin(settings_list[v], *args)

                def finish(v):
                    with torch.cuda.stream(S[v % ns]):
                        fstates[v] = _view_forward_finish(begun.pop(v))

                # the launch thread never waits for a view's counts: a view's K1 is enqueued at least one backward before its
                # binning (include/gsr.h gsr_preprocess_begin / _end; until round 6 the thread spent a K1 per view spinning
                # and the batch was bound by the host, not by the GPU: profiles/r06_i_view_pipelining.md)
                begin(0)
                for u in range(1, min(ahead + 1, k_local)):
                    begin(u)
                for u in range(0, min(ahead, k_local)):
                    finish(u)
                packed_prev, all_radii = None, []
                for v in range(k_local):
                    if v + ahead < k_local:
                        finish(v + ahead)
                    if v + ahead + 1 < k_local:
                        begin(v + ahead + 1)
                    fstate = fstates.pop(v)
                    with torch.cuda.stream(S[v % ns]):
                        # the bucket is free once view v - 1's message holds its rows.  K7 writes the stream's accumulator
                        # table and its row mask, not the bucket, so the wait sits between K7 and K8+K9 (in after_blend),
                        # except on the L1 route, whose backward runs on the autograd engine's thread
                        if packed_prev is not None:
                            if len(fstate[3]) == 6:
                                S[v % ns].wait_event(packed_prev)
                            else:
                                state["bucket_free"] = packed_prev
                        _view_backward(settings_list[v], fstate, dL_dcolor_list[v], bucket)
                        if state.pop("bucket_free", None) is not None:
                            raise RuntimeError("multiview_batch_step: the backward did not announce its blend half "
                                               "(after_blend_backward); the bucket may have been overwritten too early")
                        colors.append(fstate[0].detach())
                        depths.append(fstate[2].detach())
                        all_radii.append(fstate[1])
                        if state["planned"] is not None:
                            S[v % ns].wait_event(state["planned"])
                        _C.view_message_pack(state["plan"], grads5, bucket.rgb, bucket.campos, cap, send[v])
                        packed_prev = torch.cuda.Event()
                        packed_prev.record(S[v % ns])
                for st_ in S:
                    main.wait_stream(st_)
                radii_max = all_radii[0].clone()
                for r in all_radii[1:]:
                    torch.maximum(radii_max, r, out=radii_max)
                for t in colors + depths + all_radii:
                    t.record_stream(main)
        for v in range(0 if not pipeline else k_local, k_local):
            color, radii, depth, _ = render_view_grads(settings_list[v], params["xyz"], params["opacity"], params["features"],
                                                       params["scaling"], params["rotation"], dL_dcolor_list[v], bucket)
            colors.append(color)
            depths.append(depth)
            radii_max = radii.clone() if radii_max is None else torch.maximum(radii_max, radii, out=radii_max)
            if state["planned"] is not None:
                torch.cuda.current_stream(dev).wait_event(state["planned"])  # GPU-side: the pack kernel reads the plan
            if speculate:
                _C.view_message_pack(state["plan"], grads5, bucket.rgb, bucket.campos, cap, send[v])
            else:
                if state["done"] is not None:
                    state["done"].synchronize()  # the side stream only: this view's K8+K9 still runs on the launch stream
                c = max(int(state["count"].item()), 1)
                counts.append(c)
                m = torch.empty(_C.view_message_words(P, c), dtype=torch.float32, device=dev)
                _C.view_message_pack(state["plan"], grads5, bucket.rgb, bucket.campos, c, m)
                tight.append((m, c))
    finally:
        bucket.on_blend_done = None
    _mark(marks, "local_done", dev)
    recv = None
    if speculate:
        # the true counts sit in the messages' headers (word 3); the all-gather of the messages is enqueued BEFORE they are
        # read, so the host's one wait of the step runs underneath the collective
        mine = send[:, 3].contiguous().view(torch.int32).to(torch.int64)
        if multi:
            recv = torch.empty((n * k_local, words), dtype=torch.float32, device=dev)
            if dist.get_backend(group) == "nccl":
                dist.all_gather_into_tensor(recv, send, group=group)  # rank-major = global view order
            else:
                parts = [torch.empty_like(send) for _ in range(n)]
                dist.all_gather(parts, send.contiguous(), group=group)
                recv.copy_(torch.cat(parts))
            all_counts = [int(c) for c in recv[:, 3].contiguous().view(torch.int32).tolist()]
        else:
            recv = send
            all_counts = [int(c) for c in mine.tolist()]
        if max(all_counts) > cap:
            return None  # some message was cut short: every rank sees the same counts and re-runs the step in its exact form
    else:
        # all ranks agree on one capacity: the largest count of the whole batch (one tiny all-gather)
        if multi:
            mine = torch.tensor(counts, dtype=torch.int64, device=dev)
            gathered = [torch.empty_like(mine) for _ in range(n)]
            dist.all_gather(gathered, mine, group=group)
            all_counts = [int(c) for c in torch.cat(gathered).tolist()]
        else:
            all_counts = list(counts)
        cap = max(max(all_counts), 1)
        words = _C.view_message_words(P, cap)
        send = torch.empty((k_local, words), dtype=torch.float32, device=dev)
        for v, (m, c) in enumerate(tight):
            _relayout_message(m, P, c, send[v], cap)
        if multi:
            recv = torch.empty((n * k_local, words), dtype=torch.float32, device=dev)
            if dist.get_backend(group) == "nccl":
                dist.all_gather_into_tensor(recv, send, group=group)
            else:
                parts = [torch.empty_like(send) for _ in range(n)]
                dist.all_gather(parts, send.contiguous(), group=group)
                recv.copy_(torch.cat(parts))
        else:
            recv = send
    if multi:
        dist.all_reduce(radii_max, op=dist.ReduceOp.MAX, group=group)
    bucket.last_counts = all_counts
    bucket._cap_hint = (int(1.5 * max(max(all_counts), 1)) + 1023) // 1024 * 1024  # identical on every rank
    sh = torch.empty((P, bucket.M, 3), dtype=torch.float32, device=dev)
    _C.view_messages_accumulate(recv, P, cap, bucket.sh_degree, bucket.M, bucket.means3D_ref, grads5 + [sh],
                                row_valid=bucket.row_valid)
    bucket.views["sh"] = sh
    if bucket.row_state is not None:
        if bucket.row_valid is not None:
            bucket.row_state.copy_(bucket.row_valid)
        else:
            bucket.invalidate_rows()
    bucket.last_route = "rows"
    bucket.last_exchange = {"views": n * k_local, "views_local": k_local, "message_rows": int(cap), "speculated": bool(speculate),
                            "bytes_sent": int(send.numel() * 4) if multi else 0,
                            "bytes_received": int(recv.numel() * 4) if multi else 0}
    _mark(marks, "step_done", dev)
    return colors, radii_max, depths, dict(bucket.views)


# ----------------------------------------------------------------------------------------------------------------------
# Keeping the sharded loop equal to the reference's single-process loop
# ----------------------------------------------------------------------------------------------------------------------
def batch_loss_scale(world_size: int, rank: int = 0, per_step: str = "split") -> Dict[str, float]:
    """Factors for a rank's LOCAL loss terms (its one view) such that the SUM over the ranks -- which is what the gradient
    exchange forms -- is the batch loss the reference computes in one process over K = world_size views:

      "mean"      a term the reference takes as a MEAN over the stacked batch, e.g. `torch.nn.functional.l1_loss(images,
                  gt_images)` with images (K,H,W,3) (threestudio/systems/GassuianEditorEdit.py:100): local value x 1/K;
      "sum"       a term the reference SUMS over the batch, e.g. the perceptual loss `.sum()` (:101-104): x 1;
      "per_step"  a term computed once per step from the parameters alone, e.g. `gaussian.anchor_loss()` (:133-145,
                  gaussiansplatting/scene/gaussian_model.py:152-184): every rank holds the same parameters, so either
                  every rank adds 1/K of it (per_step="split", the default: all replicas run the same code) or rank 0 adds
                  all of it (per_step="rank0").

    Multiply before backward(); the exchanged gradients are then those of the reference's loss."""
    K = int(world_size)
    if K < 1 or not 0 <= int(rank) < K:
        raise ValueError("batch_loss_scale: need world_size >= 1 and 0 <= rank < world_size")
    if per_step not in ("split", "rank0"):
        raise ValueError('per_step must be "split" or "rank0"')
    return {"mean": 1.0 / K, "sum": 1.0,
            "per_step": (1.0 / K) if per_step == "split" else (1.0 if int(rank) == 0 else 0.0)}


def replicas_identical(tensors, group=None) -> bool:
    """True iff every rank of `group` holds bit-identical copies of `tensors` (shapes included).  One MAX and one MIN
    all-reduce over a small per-tensor fingerprint (element count, the sum of the raw words and their sum weighted by
    position, over the raw 16-bit words and exact modulo the prime 2^31 - 1: any single differing element -- a 1-ulp
    divergence anywhere, in a tensor of any size -- changes it).  Collective: every rank must call it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    fp = []
    for t in tensors:
        raw = t.detach().contiguous().reshape(-1)
        if raw.element_size() % 2 == 0:
            w = raw.view(torch.int16).to(torch.int64) & 0xFFFF  # 16-bit words: two per float32 / int32, four per 8-byte element
        else:
            w = raw.view(torch.uint8).to(torch.int64)
        # Every sum is EXACT: int64 arithmetic reduced modulo the prime 2^31 - 1 (every partial product and the final value
        # fit an int64 and, exactly, a float64) -- never an int64 -> float64 cast of a large sum.  The positional weight is
        # never zero (1 .. 65521) and a 16-bit word differs by less than the modulus, so a single differing word changes
        # both sums; a differing 32-bit word (two adjacent halves, weights a and a + 1) always changes the weighted one.
        M31 = (1 << 31) - 1
        pos = torch.arange(w.numel(), dtype=torch.int64, device=w.device) % 65521 + 1
        wm = w % M31
        fp += [torch.tensor(float(w.numel()), dtype=torch.float64, device=w.device).view(1),
               (wm.sum() % M31).to(torch.float64).view(1),
               (((wm * pos) % M31).sum() % M31).to(torch.float64).view(1)]
    mine = torch.cat(fp) if fp else torch.zeros(1, dtype=torch.float64)
    hi, lo = mine.clone(), mine.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    return bool(torch.equal(hi, lo))


def densify_synchronized(densify_fn, replicated=None, group=None, src: int = 0, check: bool = True):
    """Run a densification step so that the replicas stay bit-identical.

    `densify_and_split` draws `torch.normal` samples from the process RNG (gaussiansplatting/scene/gaussian_model.py:
    685-687), and the launcher seeds every rank differently (`pl.seed_everything(cfg.seed + get_rank())`, launch.py:
    102-103), so unsynchronised replicas diverge at the first split: different new positions, and from then on different
    gradients, selections and Gaussian counts -- the next gradient exchange would add rows of different Gaussians.
    Everything else in `densify_and_prune` (:768-809) is a deterministic function of state the exchange already keeps
    identical on all ranks (parameters, Adam moments, the summed view-space gradient, the batch-max radii).

    So: rank `src` draws one 62-bit seed from its process RNG (its own stream advances by that one draw, so successive
    densifications get different seeds), broadcasts it, every rank seeds its CPU and device generators with it, runs
    `densify_fn()` and restores the generators it had before (so the rest of the loop keeps its per-rank randomness, e.g.
    camera sampling).  With `check` the tensors `replicated()` returns afterwards (parameters, moments, masks, ...) are
    verified identical on all ranks with `replicas_identical` -- it raises rather than let a diverged replica train on.
    Returns what `densify_fn` returns."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if multi:
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64) if dist.get_rank(group) == src else torch.zeros(1, dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            seed = seed.cuda()
        dist.broadcast(seed, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        seed = int(seed.item())
    else:
        seed = None
    cpu_state = torch.random.get_rng_state()
    cuda_state = torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None
    try:
        if seed is not None:
            torch.manual_seed(seed)  # seeds the CPU generator and every device generator
        out = densify_fn()
    finally:
        torch.random.set_rng_state(cpu_state)
        if cuda_state is not None:
            torch.cuda.set_rng_state_all(cuda_state)
    if multi and check and replicated is not None:
        if not replicas_identical(list(replicated()), group=group):
            raise RuntimeError("densify_synchronized: the replicas differ after the densification step "
                               "(was the state identical before it? is densify_fn deterministic apart from the RNG?)")
    return out
