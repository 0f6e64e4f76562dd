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
    assert L.gsr_abi_version() == _native.GSR_ABI_VERSION == 6
    # ... and the library exports NOTHING else: no C++ internals, kernel handles or toolchain objects (-fvisibility=hidden
    # + csrc/gsr.map).  Read from the dynamic symbol table with nm.
    import shutil
    import subprocess

    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.strip()})
    assert exported == syms, sorted(set(exported) ^ set(syms))[:10]
    assert L.gsr_status_string(0) == b"ok" and b"channels" in L.gsr_status_string(-2)


def test_scratch_sizes_and_sort_bits():
    from gaussianeditor_amd import _native

    L = _native.lib()
    # 256^2 -> 41 bits, 512^2 -> 43, 1080p -> 45 (rasterizer_impl.cu:36-49,253)
    assert L.gsr_sort_key_bits(256, 256) == 41
    assert L.gsr_sort_key_bits(512, 512) == 43
    assert L.gsr_sort_key_bits(1920, 1080) == 45
    g0, b0, i0 = _native.scratch_sizes(1000, 0, 640, 480)
    g1, b1, i1 = _native.scratch_sizes(2000, 5000, 640, 480, 1500)
    # grouped binning: 4 bytes per tile instance (the point list) + 12 per group instance (ping-pong group ids and indices)
    # (the image scratch sized before the counts are known, R = 0, includes the forward's checkpoint pool; with the counts of
    #  a view with short lists it does not: 64 KB per tile less -- room for 16 slots of 256 float4 each, round 6)
    assert b0 == 0 and b1 > 5000 * 4 + 1500 * 12 and g1 > g0 > 1000 * 48 and i0 > i1 > 640 * 480 * 8
    assert i0 - i1 == 40 * 30 * 16 * 4096 and _native.scratch_sizes(2000, 2048 * 1200, 640, 480, 1500)[2] == i0
    # (images of up to 4 096 tiles checkpoint from a mean list of 1 200 entries per tile, larger ones from 2 048)
    assert _native.scratch_sizes(2000, 1200 * 1200, 640, 480, 1500)[2] == i0 and _native.scratch_sizes(2000, 1200 * 1200 - 1, 640, 480, 1500)[2] == i1
    hd0 = _native.scratch_sizes(2000, 0, 1920, 1080)[2]
    assert _native.scratch_sizes(2000, 2048 * 8160, 1920, 1080, 1500)[2] == hd0 > _native.scratch_sizes(2000, 2048 * 8160 - 1, 1920, 1080, 1500)[2]
    assert _native.scratch_sizes(2000, 5000, 640, 480, 3000)[1] > b1
    # beyond 131 072 tiles (2048 groups of 8 x 8) the tile-pair sort: ping-pong tile ids (uint32 there) + ping-pong indices
    assert _native.scratch_sizes(2000, 5000, 5808, 5808)[1] > 5000 * 16
    with pytest.raises(_native.GsrError):
        _native.scratch_sizes(-1, 0, 640, 480)


def test_argument_validation_needs_no_gpu():
    """Every entry point checks its arguments before it touches the device: size queries work and NULL / negative /
    inconsistent arguments come back as GSR_ERR_BAD_ARGUMENT (-1) on a machine without a GPU."""
    import ctypes

    from gaussianeditor_amd import _native

    L = _native.lib()
    n = ctypes.c_int64(0)
    assert L.gsr_view_message_words(1_000_000, 111_926, ctypes.byref(n)) == 0
    assert n.value == 4 + 977 + 18 * 111_926  # header, block offsets of ceil(P / 1024) row blocks, 18 words per row
    assert L.gsr_view_message_words(-1, 0, ctypes.byref(n)) == -1
    sz = ctypes.c_size_t(0)
    assert L.gsr_compact_workspace_size(1_000_000, ctypes.byref(sz)) == 0 and sz.value >= 2 * 4 * 977
    assert L.gsr_knn_workspace_size(1000, ctypes.byref(sz)) == 0 and sz.value > 0
    r = (ctypes.c_int64 * 2)(7, 7)
    # (P = 0 is a valid empty call everywhere; no pointer is dereferenced)
    assert L.gsr_preprocess(None, 0, 3, 16, None, None, 1.0, None, None, None, None, None, None, None, None, 64, 64, 1.0, 1.0,
                            0, 0, 0, None, None, r) == 0 and r[0] == 0 and r[1] == 0
    assert L.gsr_preprocess(None, 10, 3, 16, None, None, 1.0, None, None, None, None, None, None, None, None, 64, 64, 1.0, 1.0,
                            0, 0, 0, None, None, r) == -1
    # the split form: no ticket pointer, an empty scene, missing arrays; _end without a ticket
    tk = ctypes.c_void_p()
    assert L.gsr_preprocess_begin(None, 10, 3, 16, None, None, 1.0, None, None, None, None, None, None, None, None, 64, 64, 1.0,
                                  1.0, 0, 0, 0, None, None, None) == -1
    assert L.gsr_preprocess_begin(None, 0, 3, 16, None, None, 1.0, None, None, None, None, None, None, None, None, 64, 64, 1.0,
                                  1.0, 0, 0, 0, None, None, ctypes.byref(tk)) == -1
    assert L.gsr_preprocess_begin(None, 10, 3, 16, None, None, 1.0, None, None, None, None, None, None, None, None, 64, 64, 1.0,
                                  1.0, 0, 0, 0, None, None, ctypes.byref(tk)) == -1 and not tk.value
    assert L.gsr_preprocess_end(None, 10, 64, 64, ctypes.c_void_p(256), None, r) == -1
    assert L.gsr_preprocess(None, 10, 3, 16, None, None, 1.0, None, None, None, None, None, None, None, None, 64, 64, 1.0, 1.0,
                            0, 0, 0, None, None, None) == -1
    assert L.gsr_bin(None, 10, -1, 0, 64, 64, None, None, None) == -1
    assert L.gsr_bin(None, 10, 1 << 31, 5, 64, 64, None, None, ctypes.c_void_p(256)) == -3  # GSR_ERR_TOO_MANY
    assert L.gsr_bin(None, 10, 100, 5, 64, 64, None, None, ctypes.c_void_p(16)) == -1  # scratch must be 256-byte aligned
    assert L.gsr_bin(None, 10, 100, 101, 64, 64, ctypes.c_void_p(256), ctypes.c_void_p(256), ctypes.c_void_p(256)) == -1  # G <= R
    assert L.gsr_debug_cov3d(None, 10, None, 1.0, None, None) == -1
    assert L.gsr_sh_grad_compose(None, 10, 4, 16, 1, None, None, None, None) == -1  # degree > 3
    assert L.gsr_view_message_plan(None, 10, None, None, None, None, None) == -1
    assert L.gsr_view_messages_accumulate(None, 10, 3, 16, 0, None, 0, 0, None, None) == -1  # no views
    assert L.gsr_adam_step(None, 0, None, 1, 0.9, 0.999, 1e-15, None, None) in (0, -1)
    assert L.gsr_append_rows(None, 10, 5, 0, None) == 0 and L.gsr_append_rows(None, 0, 0, 3, None) == 0  # nothing to do
    assert L.gsr_append_rows(None, 10, 5, 3, None) == -1 and L.gsr_append_rows(None, -1, 5, 1, None) == -1
    assert L.gsr_append_rows(None, 10, 5, 33, None) == -1
    bad = (_native.AppendTensor * 1)(_native.AppendTensor(None, None, ctypes.c_void_p(16), 12))  # P > 0 without a source
    assert L.gsr_append_rows(None, 10, 5, 1, bad) == -1
    assert b"31-bit" in L.gsr_status_string(-3)
    # per-call flags (ABI 2): unknown bits are rejected before anything else is looked at; the library has no option state
    assert not hasattr(L, "gsr_set_option")
    one = ctypes.c_void_p(256)
    assert L.gsr_preprocess(None, 10, 3, 16, one, one, 1.0, one, one, one, None, None, one, one, one, 64, 64, 1.0, 1.0,
                            0, 0, 64, one, one, r) == -1
    assert L.gsr_blend_forward(None, 10, 5, 64, 64, one, one, one, one, one, one, 64) == -1
    assert L.gsr_blend_backward(None, 10, 5, 64, 64, one, one, one, one, one, one, None, 64) == -1
    assert L.gsr_trace_weights(None, 10, 5, 64, 64, 1, one, one, one, one, one, one, 64) == -1
    # the blend backward's work items carry the tile id in 20 bits (GSR_MAX_TILES of include/gsr.h): a larger image is
    # refused, not walked with masked tile ids (16 400^2 pixels = 1 025^2 tiles > 2^20)
    assert L.gsr_blend_backward(None, 10, 5, 16400, 16400, one, one, one, one, one, one, None, 0) == -1
    # the accumulator table must not straddle 64-byte lines (one memory-side request per row)
    assert L.gsr_blend_backward(None, 10, 5, 64, 64, one, one, one, one, one, ctypes.c_void_p(256 + 16), None, 0) == -1
    assert "GSR_MAX_TILES (1 << 20)" in open(os.path.join(ROOT, "include", "gsr.h")).read()


def test_options_are_per_render_and_per_thread():
    """The Python-side default of the flags, its thread-local override, and their validation (no GPU needed)."""
    import threading

    import gaussianeditor_amd
    from gaussianeditor_amd import options

    assert options.current_flags() == 0 and gaussianeditor_amd.get_tile_bounds() == "reference"
    gaussianeditor_amd.set_tile_bounds("alpha")
    gaussianeditor_amd.set_fast_exp(True)
    try:
        assert options.current_flags() == 3 and gaussianeditor_amd.get_fast_exp()
        seen = {}
        with options.override(0):
            assert options.current_flags() == 0
            t = threading.Thread(target=lambda: seen.setdefault("other", options.current_flags()))
            t.start()
            t.join()
        assert seen["other"] == 3  # the override belonged to this thread only
        assert options.current_flags() == 3
        with pytest.raises(ValueError):
            options.set_default_flags(8)
        with pytest.raises(ValueError):
            gaussianeditor_amd.set_tile_bounds("tight")
    finally:
        gaussianeditor_amd.set_tile_bounds("reference")
        gaussianeditor_amd.set_fast_exp(False)
    assert options.current_flags() == 0


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


def test_forward_only_flag_is_refused_by_the_backward_entry_points():
    """a backward declared with GSR_FLAG_FORWARD_ONLY is an argument error (checked before any device
    work, so it runs here without a GPU); the buffer-level guard is covered on the GPU (test_gpu_round4.py)."""
    import ctypes

    from gaussianeditor_amd import _native

    L = _native.lib()
    one = ctypes.c_void_p(256)  # never dereferenced: the flags are rejected first
    st = L.gsr_blend_backward(None, 4, 4, 32, 32, one, one, one, one, one, one, None, 8)
    assert st == -1 and b"bad argument" in L.gsr_status_string(st)
    st = L.gsr_blend_backward(None, 4, 4, 32, 32, one, one, one, one, one, one, None, 8 | 4)
    assert st == -1
    # GSR_FLAG_ACC_SELF_CLEAN (32) belongs to gsr_backward alone -- K7 on its own cannot leave the table zero --, and not next to
    # GSR_FLAG_CLEAR_GRADS (4)
    assert L.gsr_blend_backward(None, 4, 4, 32, 32, one, one, one, one, one, one, None, 32) == -1
    assert L.gsr_backward(None, 4, 3, 16, 4, 32, 32, one, one, one, None, one, 1.0, one, None, one, one, one, 1.0, 1.0, one, one, one,
                          one, one, one, one, one, None, one, one, one, one, one, 32 | 4) == -1


def test_header_states_the_memory_and_alignment_contracts():
    """The C-ABI footguns live in include/gsr.h, not only in the Makefile."""
    hdr = open(os.path.join(ROOT, "include", "gsr.h")).read()
    assert "coarse-grained" in hdr and "munsafe-fp-atomics" in hdr
    assert "256-byte aligned" in hdr
    assert "forward-only" in hdr and "GSR_ERR_BAD_ARGUMENT" in hdr


def test_the_test_process_never_writes_bytecode_into_the_reference_tree():
    import sys

    assert sys.dont_write_bytecode and os.environ.get("PYTHONDONTWRITEBYTECODE") == "1"
