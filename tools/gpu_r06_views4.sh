#!/bin/bash
TAG=${1:-r06_y2}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
R=$GRAFT_REPO_ROOT
O=gpurun_out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4))"; }
for a in 0 1 2; do
  export GSR_VIEW_AHEAD=$a
  timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k "batch or views" 2>&1 | tail -2
  for i in 1 2; do
    timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs 2>/dev/null | line "views8 ahead=$a"
  done
  mkdir -p $R/$O/${TAG}_pipe
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/$O/${TAG}_pipe -o p -- python $R/bench.py --views 8 --steps 10 --warmup 5 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1)
  DB=$(find $R/$O/${TAG}_pipe -name "*.db" | head -1)
  python tools/rocpd_overlap.py $DB 3000 > $O/${TAG}_ahead${a}_overlap.md 2> /dev/null
  rm -rf $R/$O/${TAG}_pipe
done
