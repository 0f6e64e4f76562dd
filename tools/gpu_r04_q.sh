#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r04_q}
timeout 200 python tools/bench_binning.py --oracle > $O/${TAG}_binning.txt 2>&1
grep -q "point list == oracle: True" $O/${TAG}_binning.txt || { echo "BINNING MISMATCH"; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -x -q -k "backward or three_way or segments or headline or needle or apply_weights or aux or two_streams or empty or 4k" 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
timeout 900 python tools/ab_variants.py --steps 100 auto3 off=GSR_CK_CHUNKS=0 ck2=GSR_CK_CHUNKS=2 ck4=GSR_CK_CHUNKS=4 auto3_b off_b=GSR_CK_CHUNKS=0 seg3=GSR_BWD_SEG=3 seg8=GSR_BWD_SEG=8 > $O/${TAG}_ab.txt 2>&1
timeout 600 python tools/ab_variants.py --steps 50 --s0 0.05 autod offd=GSR_CK_CHUNKS=0 ck4d=GSR_CK_CHUNKS=4 ck12d=GSR_CK_CHUNKS=12 >> $O/${TAG}_ab.txt 2>&1
timeout 600 python tools/ab_variants.py --steps 30 --gaussians 6000000 auto6 off6=GSR_CK_CHUNKS=0 ck4_6=GSR_CK_CHUNKS=4 >> $O/${TAG}_ab.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_ab.txt
