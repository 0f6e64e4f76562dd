#!/bin/bash
# round 4, call g: same-box A/B of K1's wave-cooperative SH load against one row per thread (variant library k1row)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r04_g}
timeout 900 python tools/ab_variants.py --smoke --steps 100 coop rows@k1row coop_b rows_b@k1row > $O/${TAG}_ab.txt 2>&1
timeout 900 python tools/ab_variants.py --steps 30 --gaussians 6000000 coop6 rows6@k1row >> $O/${TAG}_ab.txt 2>&1
cat $O/${TAG}_ab.txt | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_all_stages or sh_degrees or colors_precomp or culling or nonfinite or edge_geometries or random_configuration or backward_vs_oracle or headline" 2>&1 | tail -4
