#!/bin/bash
# Checkpoints fine in front, coarse behind (the small-image class): parity of the segment paths, then against the uniform table.
TAG=${1:-r06_o}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -k "list_segments or deep_translucent or edit_loop_workload or three_way_parity or deep" 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
unset GSR_REQUIRE_REF
for cfg in "--gaussians 500000" "--gaussians 1000000" "--gaussians 2000000" "--gaussians 3000000" "--gaussians 6000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" "--scene v2 --gaussians 3000000" \
           "--width 400 --height 400 --gaussians 1000000" "--width 400 --height 400 --gaussians 2000000" "--width 256 --height 256 --gaussians 300000" "--width 256 --height 256 --gaussians 1000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_geom.txt
  echo "uniform  $(GSR_CK_GEOM=0 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_geom.txt
  echo "geom     $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_geom.txt
done
