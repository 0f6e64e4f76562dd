#!/usr/bin/env python3
"""Spare GPU minutes as assurance: tests/test_gpu_parity.py::test_random_configuration_sweep (seeded random shapes, forward
stage by stage + all six gradients against the oracle) for MORE seeds than the suite runs (it runs 0..11).
    python tools/fuzz_parity.py --first 12 --count 150 [--seconds 240] [--judge] [--only SEED ...]
--judge: a seed whose gradients exceed the 1e-5 bar is taken to the four-way comparison of tests/test_gpu_round5.py::four_way
(reference's own backward / oracle / product / float64 autograd): is the product further from float64 than the reference's
backward is?  (The oracle adds a Gaussian's pixels one after the other in binary32; for a splat that covers 10^4..10^5 pixels its
own sum is the least accurate of the three.)"""
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
    ap.add_argument("--judge", action="store_true")
    ap.add_argument("--only", type=int, nargs="*", default=None, help="only these seeds")
    a = ap.parse_args()
    import test_gpu_parity as tp
    from oracle import cpu

    cpu.build()
    t0, done, bad, judged_ok, judged_bad, unjudged = time.time(), 0, [], [], [], []
    for seed in (a.only if a.only else range(a.first, a.first + a.count)):
        if time.time() - t0 > a.seconds:
            break
        try:
            tp.test_random_configuration_sweep(cpu, seed)
        except Exception:  # noqa: BLE001 (report and go on: the seed is what matters)
            bad.append(seed)
            print(f"seed {seed} FAILED\n{traceback.format_exc()[-1500:]}", flush=True)
            if a.judge:
                import test_gpu_round5 as t5

                case, sm, D = tp.sweep_case(seed)
                try:
                    t5.four_way(cpu, case, sm, D, seed, image_tol=5e-4)
                    judged_ok.append(seed)
                    print(f"seed {seed}: four-way: no further from float64 than the reference's backward", flush=True)
                except t5.RestatementMismatch as e:
                    unjudged.append(seed)
                    print(f"seed {seed}: not judged: {e}", flush=True)
                except Exception:  # noqa: BLE001
                    judged_bad.append(seed)
                    print(f"seed {seed}: four-way FAILED\n{traceback.format_exc()[-1500:]}", flush=True)
        done += 1
    print(f"fuzz: {done} seeds from {a.first} in {time.time() - t0:.0f} s, failures: {bad if bad else 'none'}"
          + (f"; four-way: no further from float64 than the reference's backward {judged_ok}, FAILED {judged_bad if judged_bad else 'none'}, "
             f"not judged (float64 renders another image) {unjudged}" if a.judge else ""))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
