#!/bin/bash
# Closing tail: the -m gpu suite and smoke() on the final tests / bench.py, then the bench lines (counters were taken by
# tools/gpu_final.sh on the same kernel sources).
TAG=${1:-r06_zu}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
unset GSR_REQUIRE_REF
bash tools/gpu_bench_lines.sh ${TAG} > $O/${TAG}_bench_lines.txt 2>&1
tail -3 $O/${TAG}_pytest.txt; tail -1 $O/${TAG}_smoke.txt; grep "bench.json\|views8.json" $O/${TAG}_bench_lines.txt
