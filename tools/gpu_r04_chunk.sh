#!/bin/bash
# group_chunk_kernel: two entries of each stream per round in the list-append loop
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
timeout 120 python tools/bench_binning.py --oracle
timeout 120 python tools/bench_binning.py --s0 0.05
timeout 120 python tools/bench_binning.py --gaussians 6000000
timeout 300 python tools/ab_variants.py --smoke --no-extra-configs new old@binold new2 old2@binold
timeout 300 python tools/ab_variants.py --no-extra-configs --s0 0.05 new old@binold
timeout 300 python tools/ab_variants.py --no-extra-configs --steps 50 --gaussians 6000000 new old@binold
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_round3.py -q -x 2>&1 | tail -3
} > $O/r04_chunk.txt 2>&1
cat $O/r04_chunk.txt
