"""CPU: the C-ABI library loads and exports exactly what include/gsr.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gsr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported_and_typed():
    from gaussianeditor_amd import _native

    syms = _declared_symbols()
    assert len(syms) >= 15
    lib = ctypes.CDLL(_native.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gsr.h but not exported by libgsr_hip.so"
    # the Python binding types every declared symbol, and nothing that is not declared
    assert sorted(_native.SIGNATURES) == syms
    L = _native.lib()
    assert L.gsr_abi_version() == _native.GSR_ABI_VERSION == 1
    assert L.gsr_status_string(0) == b"ok" and b"channels" in L.gsr_status_string(-2)


def test_scratch_sizes_and_sort_bits():
    from gaussianeditor_amd import _native

    L = _native.lib()
    # 256^2 -> 41 bits, 512^2 -> 43, 1080p -> 45 (rasterizer_impl.cu:36-49,253)
    assert L.gsr_sort_key_bits(256, 256) == 41
    assert L.gsr_sort_key_bits(512, 512) == 43
    assert L.gsr_sort_key_bits(1920, 1080) == 45
    g0, b0, i0 = _native.scratch_sizes(1000, 0, 640, 480)
    g1, b1, i1 = _native.scratch_sizes(2000, 5000, 640, 480)
    assert b0 == 0 and b1 > 5000 * 16 and g1 > g0 > 1000 * 48 and i0 == i1 > 640 * 480 * 8
    with pytest.raises(_native.GsrError):
        _native.scratch_sizes(-1, 0, 640, 480)


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU, never route through a CPU implementation."""
    import torch

    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    rs = GaussianRasterizationSettings(32, 32, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(rs)(x, x, torch.ones(4, 1), colors_precomp=torch.ones(4, 3), scales=torch.ones(4, 3),
                               rotations=torch.ones(4, 4))
    # and nothing under gaussianeditor_amd/ imports the oracle
    pkg = os.path.join(ROOT, "gaussianeditor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libgsr_oracle" not in src and "gsro_" not in src, f
