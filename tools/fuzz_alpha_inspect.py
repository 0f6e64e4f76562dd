#!/usr/bin/env python3
"""Re-run alpha-tile-bounds configurations tools/fuzz_more.py reported and print which assertion stopped them."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tp  # noqa: E402
from oracle import cpu  # noqa: E402

cpu.build()
CASES = [(6215, 95, 238, 0.05, 529), (4703, 483, 196, 0.05, 94), (7434, 309, 379, 0.02, 557), (2956, 461, 121, 0.2, 464),
         (2196, 320, 45, 0.05, 358)]
for c in CASES:
    try:
        tp.test_alpha_tile_bounds_leave_results_unchanged(cpu, *c)
        print(c, "passed")
    except AssertionError as e:
        fr = traceback.extract_tb(e.__traceback__)[-1]
        print(c, "STOPPED AT line", fr.lineno, ":", fr.line, "|", str(e)[:200])
