#!/usr/bin/env python3
"""Spare GPU minutes as assurance: tests/test_gpu_parity.py::test_random_configuration_sweep (seeded random shapes, forward
stage by stage + all six gradients against the oracle) for MORE seeds than the suite runs (it runs 0..11).
    python tools/fuzz_parity.py --first 12 --count 150 [--seconds 240]"""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=12)
    ap.add_argument("--count", type=int, default=150)
    ap.add_argument("--seconds", type=float, default=240.0)
    a = ap.parse_args()
    import test_gpu_parity as tp
    from oracle import cpu

    cpu.build()
    t0, done, bad = time.time(), 0, []
    for seed in range(a.first, a.first + a.count):
        if time.time() - t0 > a.seconds:
            break
        try:
            tp.test_random_configuration_sweep(cpu, seed)
        except Exception:  # noqa: BLE001 (report and go on: the seed is what matters)
            bad.append(seed)
            print(f"seed {seed} FAILED\n{traceback.format_exc()[-1500:]}", flush=True)
        done += 1
    print(f"fuzz: {done} seeds from {a.first} in {time.time() - t0:.0f} s, failures: {bad if bad else 'none'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
