#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r04_o}
timeout 200 python tools/bench_binning.py --oracle > $O/${TAG}_binning.txt 2>&1
grep -q "point list == oracle: True" $O/${TAG}_binning.txt || { echo "BINNING MISMATCH"; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -x -q -k "backward or three_way or segments or headline or needle or apply_weights or aux or two_streams" 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
timeout 900 python tools/ab_variants.py --steps 100 auto forced=GSR_CK_MIN_LIST=0 auto_b forced_b=GSR_CK_MIN_LIST=0 > $O/${TAG}_ab.txt 2>&1
timeout 600 python tools/ab_variants.py --steps 50 --s0 0.05 autod offd=GSR_CK_CHUNKS=0 >> $O/${TAG}_ab.txt 2>&1
timeout 600 python tools/ab_variants.py --steps 30 --gaussians 6000000 auto6 off6=GSR_CK_CHUNKS=0 >> $O/${TAG}_ab.txt 2>&1
timeout 600 python tools/ab_variants.py --steps 50 --s0 0.03 auto3 off3=GSR_CK_CHUNKS=0 on3=GSR_CK_MIN_LIST=0 >> $O/${TAG}_ab.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_ab.txt
