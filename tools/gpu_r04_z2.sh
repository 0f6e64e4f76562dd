#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
nproc; cat /proc/loadavg
timeout 600 python -m pytest tests/test_gpu_round4.py -q -x -k "bench" 2>&1 | tail -3
timeout 900 python bench.py > $O/r04_z2_bench.json 2> $O/r04_z2_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/r04_z2_bench_20.json 2>> $O/r04_z2_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --prewarm 0 --no-cpu-baseline --no-extra-configs > $O/r04_z2_bench_20_cold.json 2>> $O/r04_z2_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline > $O/r04_z2_bench_views8.json 2>> $O/r04_z2_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-view-pipeline > $O/r04_z2_bench_views8_serial.json 2>> $O/r04_z2_bench.err
timeout 300 python bench.py --steps 100 --force-exchange --no-cpu-baseline > $O/r04_z2_bench_forced_exchange.json 2>> $O/r04_z2_bench.err
cat /proc/loadavg
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/r04_z2_bench*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), d["step_ms_gpu"])
PY
