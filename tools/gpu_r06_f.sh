#!/bin/bash
# Round 6: K7's scheduling knobs re-measured now that its atomics are cheap (same box, 200 steps each).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r06_f}
timeout 2400 python tools/ab_variants.py --steps 200 base h6=GSR_BWD_HALVES=6 h8=GSR_BWD_HALVES=8 h12=GSR_BWD_HALVES=12 h16=GSR_BWD_HALVES=16 \
  ck8=GSR_CK_CHUNKS=8 ck8s3=GSR_CK_CHUNKS=8,GSR_BWD_SEG=3 ck4=GSR_CK_CHUNKS=4 ck4s3=GSR_CK_CHUNKS=4,GSR_BWD_SEG=3 w3=GSR_BLEND_WAVES_PER_SIMD=3 baseb > $O/${T}_ab.txt 2>&1
cat $O/${T}_ab.txt
timeout 900 python tools/ab_variants.py --steps 100 --scene v2 v2base v2h8=GSR_BWD_HALVES=8 v2s3=GSR_BWD_SEG=3 v2s8=GSR_BWD_SEG=8 v2ck4=GSR_CK_CHUNKS=4 > $O/${T}_ab_v2.txt 2>&1
cat $O/${T}_ab_v2.txt
