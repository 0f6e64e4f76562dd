#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/gpu_r04_u.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_near.py -q -x 2>&1 | tail -15 > $O/r04_u_pytest.txt
timeout 300 python tools/bench_near.py > $O/r04_u_near.txt 2>&1
timeout 300 python tools/bench_near.py --points 6000000 --reps 2 >> $O/r04_u_near.txt 2>&1
cat $O/r04_u_pytest.txt; cat $O/r04_u_near.txt
