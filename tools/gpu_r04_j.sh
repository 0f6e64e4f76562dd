#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r04_j}
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -8 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
for extra in "" "--no-view-pipeline"; do
  timeout 300 python bench.py --views 8 --steps 30 --warmup 5 --no-cpu-baseline $extra > $O/${TAG}_v8$extra.json 2>> $O/${TAG}.err
  timeout 300 python bench.py --views 8 --steps 30 --warmup 5 --no-cpu-baseline --s0 0.05 $extra > $O/${TAG}_v8_deep$extra.json 2>> $O/${TAG}.err
done
GSR_BLEND_WAVES_PER_SIMD=3 timeout 300 python bench.py --views 8 --steps 30 --warmup 5 --no-cpu-baseline > $O/${TAG}_v8_w3.json 2>> $O/${TAG}.err
GSR_BLEND_WAVES_PER_SIMD=4 timeout 300 python bench.py --views 8 --steps 30 --warmup 5 --no-cpu-baseline > $O/${TAG}_v8_w4.json 2>> $O/${TAG}.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/${TAG}_v8*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["value"], 1), "view-it/s", round(d["ms_per_step"], 3), "ms/step", d["config"]["view_pipeline"], d["config"]["blend_waves_per_simd"])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $O/${TAG}.err
