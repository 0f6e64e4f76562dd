#!/bin/bash
# round 4, call k: the backward's list segments -- binning check, parity (default knobs and forced segmentation), same-box A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r04_k}
timeout 200 python tools/bench_binning.py --oracle > $O/${TAG}_binning.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_binning.txt
grep -q "point list == oracle: True" $O/${TAG}_binning.txt || { echo "BINNING MISMATCH"; exit 1; }
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -x -q -k "backward or three_way or segments or headline or needle or apply_weights or aux or two_streams" 2>&1 | tail -15 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
timeout 900 python tools/ab_variants.py --steps 100 seg noseg=GSR_BWD_SEG=0 seg_b noseg_b=GSR_BWD_SEG=0 seg3=GSR_BWD_SEG=3 ck4=GSR_CK_CHUNKS=4 ck0=GSR_CK_CHUNKS=0 > $O/${TAG}_ab.txt 2>&1
timeout 600 python tools/ab_variants.py --steps 50 --s0 0.05 segd nosegd=GSR_BWD_SEG=0 ck4d=GSR_CK_CHUNKS=4 ck16d=GSR_CK_CHUNKS=16 >> $O/${TAG}_ab.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_ab.txt
