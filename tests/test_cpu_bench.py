"""CPU tests of bench.py's bookkeeping: the byte models of SURVEY.md section 8(d), the workload keys, the stamp that ties
profiles/traffic_latest.json to the kernel sources, and the roofline block's arithmetic (no GPU: nothing is launched)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_byte_models_are_section_8d_and_the_compulsory_model_never_exceeds_it():
    # the headline view's figures (profiles/r05_zz_bench.json): 1 M Gaussians, SH3, 1920x1080
    P, V, R, N, T, M = 1_000_000, 1_000_000, 4_841_672, 1920 * 1080, 120 * 68, 16
    ab, cb = bench.algorithmic_bytes(P, V, R, N, T, M), bench.compulsory_bytes(P, V, R, N, T, M)
    assert set(ab) == set(cb) == set(bench.STAGES)
    assert ab["preprocess"] == (48 + 12 * M) * P + 67 * V            # SURVEY.md section 8(d), K1+K2
    assert ab["bin"] == 36 * R + 16 * T                              # K3 + K4 + K5
    assert ab["blend_forward"] == 44 * R + 24 * N                    # K6
    assert ab["blend_backward"] == 20 * N + 76 * R                   # K7
    assert ab["preprocess_backward"] == (103 + 12 * M) * V + (56 + 12 * M) * P
    assert abs(ab["blend_backward"] / 1e6 - 409.4) < 0.5             # (the 409 MB DESIGN.md section 6 quotes)
    for k in bench.STAGES:
        assert 0 < cb[k] <= ab[k]
    assert cb["preprocess"] == ab["preprocess"] and cb["blend_backward"] == 20 * N + 4 * R + 92 * V


def test_workload_keys():
    assert bench.workload_key(1_000_000, 1920, 1080, 0.01) == "synth-v1:1000000:1920x1080:s0=0.01"
    assert bench.workload_key(1e6, 1920, 1080, 0.05) == "synth-v1:1000000:1920x1080:s0=0.05"
    assert bench.workload_key(6_000_000, 1920, 1080, 0.0, "v2") == "synth-v2:6000000:1920x1080"


def test_committed_counters_belong_to_the_committed_kernel_sources(monkeypatch):
    """profiles/traffic_latest.json is quoted by the bench line only while its stamp equals the hash of gaussianeditor_amd/csrc:
    the file in the tree must be the one measured on the sources in the tree (a kernel edit without new PMC passes would
    silently null `roofline.traffic` on the driver's run), and a stale stamp must be refused, with the reason."""
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert tj["csrc_sha16"] == bench.csrc_sha16()
    for key in (bench.workload_key(1_000_000, 1920, 1080, 0.01), bench.workload_key(1_000_000, 1920, 1080, 0.05),
                bench.workload_key(6_000_000, 1920, 1080, 0.01), bench.workload_key(1_000_000, 1920, 1080, 0.0, "v2")):
        c, src = bench.load_counters(key)
        assert c is not None and "rocprofv3" in src
        assert set(c["per_launch_bytes"]) == set(bench.STAGES) and all(v > 0 for v in c["per_launch_bytes"].values())
        assert c["valu_wave_insts"]["blend_backward"] > c["valu_wave_insts"]["blend_forward"] > 1e6
    c, why = bench.load_counters("synth-v1:123:4x4:s0=1")
    assert c is None and "no counters" in why
    monkeypatch.setattr(bench, "csrc_sha16", lambda: "0" * 16)
    c, why = bench.load_counters(bench.workload_key(1_000_000, 1920, 1080, 0.01))
    assert c is None and why.startswith("stale")


def test_roofline_block_arithmetic(monkeypatch):
    import types

    import torch

    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev: types.SimpleNamespace(multi_processor_count=256))
    st = {"preprocess": 0.130, "bin": 0.080, "blend_forward": 0.100, "blend_backward": 0.200, "preprocess_backward": 0.066}
    P, V, R, N, T, M = 1_000_000, 1_000_000, 4_841_672, 1920 * 1080, 8160, 16
    ab, cb = bench.algorithmic_bytes(P, V, R, N, T, M), bench.compulsory_bytes(P, V, R, N, T, M)
    counters = {"per_launch_bytes": {"blend_backward": 246e6}, "valu_wave_insts": {"blend_backward": 68.9e6}}
    rf = bench.roofline_block(st, ab, cb, "blend_backward", counters, "unit test", "cuda:0", 97_394_339)
    assert rf["bound"] == "hbm" and rf["kernel"] == "blend_backward" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["achieved"] == pytest.approx(ab["blend_backward"] / 0.2e-3 / 1e9) and rf["frac"] == pytest.approx(rf["achieved"] / 8000.0)
    assert rf["frac"] == rf["frac_8d"] and rf["frac_compulsory"] == pytest.approx(cb["blend_backward"] / 0.2e-3 / 1e9 / 8000.0)
    assert rf["traffic"] == 246e6 and rf["frac_counter"] == pytest.approx(246e6 / 0.2e-3 / 1e9 / 8000.0)
    # 1 024 SIMDs at 2.4 GHz; the measured issue cost of a VALU wave-instruction (K7's group body replayed, 4 waves per SIMD) and
    # the nominal 4 cycles next to it
    assert rf["valu_issue_frac"] == pytest.approx(68.9e6 * bench.ISSUE_CYCLES_PER_VALU / (1024 * 0.2e-3 * 2.4e9))
    assert rf["valu_issue_frac_4cycle"] == pytest.approx(68.9e6 / (1024 * 0.2e-3 * 2.4e9 / 4))
    assert rf["pixel_instances_per_s"] == pytest.approx(97_394_339 / 0.2e-3) and "limiter" in rf
    none = bench.roofline_block(st, ab, cb, "preprocess", None, "stale: ...", "cuda:0", 0)
    assert none["traffic"] is None and none["frac_counter"] is None and "limiter" not in none and 0 < none["frac"] < 1
