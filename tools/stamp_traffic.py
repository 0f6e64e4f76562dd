#!/usr/bin/env python3
"""Write profiles/traffic_latest.json: per-launch HBM bytes of the stage kernels (from rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE passes, tools/rocpd_pmc.py), stamped with the hash of the kernel sources they were measured on -- bench.py
only quotes the file while that hash matches the build it runs.
Usage: tools/stamp_traffic.py "<source note>" blend_backward=<bytes> blend_forward=<bytes> preprocess=<bytes> ..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16  # noqa: E402

note, pairs = sys.argv[1], dict(a.split("=") for a in sys.argv[2:])
out = {"workload": {"gaussians": 1000000, "width": 1920, "height": 1080}, "source": note, "csrc_sha16": csrc_sha16(),
       "per_launch_bytes": {k: int(float(v)) for k, v in pairs.items()}}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
