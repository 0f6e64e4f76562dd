#!/bin/bash
# The split forward in the pipelined view batch: tests that cover it, the batch's line, its overlap timeline, the host profile.
TAG=${1:-r06_x}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
R=$GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -q -x -k "halves or batch or pipelin or views or rccl or exchange" 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
for i in 1 2; do
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('views8', d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --no-view-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('views8 serial', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'])"
mkdir -p $R/$O/${TAG}_pipe
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/$O/${TAG}_pipe -o p -- python $R/bench.py --views 8 --steps 10 --warmup 5 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1)
DB=$(find $R/$O/${TAG}_pipe -name "*.db" | head -1)
python tools/rocpd_overlap.py $DB 3000 > $O/${TAG}_pipe_overlap.md 2> /dev/null
rm -rf $R/$O/${TAG}_pipe
tail -34 $O/${TAG}_pipe_overlap.md
bash tools/gpu_r06_hostprof.sh ${TAG} > /dev/null 2>&1
