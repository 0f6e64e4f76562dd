#!/bin/bash
# round 4, call e: the new host-side pieces on the GPU (round-4 tests, the default bench line incl. extra_configs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r04_e}
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -25 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
timeout 600 python bench.py --steps 50 --warmup 10 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 3000 $O/${TAG}_bench.json; tail -5 $O/${TAG}_bench.err
timeout 300 python bench.py --views 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_views8.json 2>> $O/${TAG}_bench.err
tail -c 1500 $O/${TAG}_bench_views8.json
