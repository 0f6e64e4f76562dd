#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_near.py tests/test_gpu_properties.py -q -x 2>&1 | tail -3
timeout 900 python bench.py > $O/r04_end_bench.json 2> $O/r04_end_bench.err; echo "rc=$? lines=$(wc -l < $O/r04_end_bench.json)"
python - <<PY
import json
d=json.load(open("$O/r04_end_bench.json"))
print(round(d["value"],1), d["ms_per_step"], d["step_ms_gpu"], d["roofline"]["traffic"] is not None, list(d["extra_configs"].keys()), d["cpu_baseline"]["value"])
PY
