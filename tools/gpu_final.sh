#!/bin/bash
# The round's closing session when only non-hot-path sources changed since the last full one (tools/gpu_session.sh): the
# -m gpu suite, smoke(), the PMC passes (the csrc hash in profiles/traffic_latest.json must match the shipped sources) and
# the bench lines that quote them.
#   gpurun --timeout 2400 -- 'bash tools/gpu_final.sh r04_z'
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
bash tools/gpu_counters.sh ${TAG} > $O/${TAG}_counters.log 2>&1
bash tools/gpu_bench_lines.sh ${TAG} > $O/${TAG}_bench_lines.txt 2>&1
tail -3 $O/${TAG}_pytest.txt; tail -2 $O/${TAG}_smoke.txt; cat $O/${TAG}_bench_lines.txt
