"""CPU: the identity half of view reuse (diff_gaussian_rasterization/_reuse.py) -- which tensors count as "the tensor the
remembered render read", without any device work."""
import torch

from gaussianeditor_amd.diff_gaussian_rasterization import _reuse


def test_tracked_tensor_identity_rules():
    t = torch.arange(12, dtype=torch.float32).reshape(4, 3)
    tr = _reuse._Tracked(t)
    assert tr.match(t) is True
    assert tr.match(t.float()) is True                 # .float() of a float32 tensor is the tensor itself
    assert tr.match(t.view(4, 3)) is True              # another view of the same memory, same layout
    assert tr.match(t.clone()) is None                 # equal content elsewhere: only a comparison can tell
    assert tr.match(t[:3]) is False                    # another shape
    assert tr.match(t.double()) is False
    assert tr.match(t.t().contiguous().t()) is None    # same shape, other memory
    t.add_(1.0)                                        # an in-place write (optimizer step): the old content is gone
    assert tr.match(t) is False and tr.match(t.clone()) is False
    # a parameter behind an activation: every call makes a fresh tensor
    p = torch.nn.Parameter(torch.randn(5, 1))
    a, b = torch.sigmoid(p), torch.sigmoid(p)
    assert _reuse._Tracked(a).match(b) is None
    # absent features (empty tensors) match whatever object carries them
    assert _reuse._Tracked(torch.empty(0)).match(torch.empty(0)) is True


def test_switch_and_thread_local_state():
    import threading

    import gaussianeditor_amd

    was = gaussianeditor_amd.get_view_reuse()
    try:
        gaussianeditor_amd.set_view_reuse(False)
        assert not gaussianeditor_amd.get_view_reuse()
        gaussianeditor_amd.set_view_reuse(True)
        _reuse._local.entries = {"x": 1}
        seen = []
        th = threading.Thread(target=lambda: seen.append(getattr(_reuse._local, "entries", None)))
        th.start()
        th.join()
        assert seen == [None]  # what one thread remembers, another does not see (the web UI renders from a second thread)
        _reuse.forget()
        assert _reuse._local.entries == {}
    finally:
        gaussianeditor_amd.set_view_reuse(was)
