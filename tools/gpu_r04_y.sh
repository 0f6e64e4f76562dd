#!/bin/bash
# K1 leaves the first depth pass's histogram (Geom::hist0): parity of the lists, then same-box A/B against GSR_K1_HIST=0
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
timeout 120 python tools/bench_binning.py --oracle
timeout 120 python tools/bench_binning.py --s0 0.05
timeout 120 python tools/bench_binning.py --width 512 --height 512
timeout 300 python tools/ab_variants.py --smoke --no-extra-configs new old=GSR_K1_HIST=0 new2 old2=GSR_K1_HIST=0
timeout 300 python tools/ab_variants.py --no-extra-configs --steps 50 --gaussians 6000000 new old=GSR_K1_HIST=0 new2 old2=GSR_K1_HIST=0
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -5
} > $O/r04_y.txt 2>&1
cat $O/r04_y.txt
