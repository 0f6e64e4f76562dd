#!/bin/bash
# Fine checkpoints + merged segment items (round 6, second half): parity of the segment paths, then the sweep of the item
# size (GSR_BWD_SEG_ITEM, sixteenths of a workgroup's fair share) over the workloads of profiles/r06_k, next to the library of
# the previous commit on the same box (build_variants/libgsr_head.so).
#   gpurun --timeout 2400 -- 'bash tools/gpu_r06_merge.sh r06_m'
TAG=${1:-r06_m}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
if [ "${2:-}" != "sweeponly" ]; then
timeout 1500 python -m pytest tests -m gpu -q -x -k "list_segments or deep_translucent or edit_loop_workload or three_way_parity or deep" 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
fi
unset GSR_REQUIRE_REF
HEAD_LIB=$GRAFT_REPO_ROOT/build_variants/libgsr_head.so
for cfg in "--gaussians 1000000" "--gaussians 500000" "--gaussians 2000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" \
           "--width 400 --height 400 --gaussians 1000000" "--width 640 --height 640 --gaussians 1000000" "--width 800 --height 800 --gaussians 3000000" \
           "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000" "--width 1920 --height 1080 --scene v2 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_sweep.txt
  echo "head       $(GSR_LIBRARY_PATH=$HEAD_LIB python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_sweep.txt
  echo "head ck8   $(GSR_LIBRARY_PATH=$HEAD_LIB GSR_CK_CHUNKS=8 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_sweep.txt
  for it in 0 2 3 4 6 8 12; do
    echo "new it=$it   $(GSR_CK_MIN_LIST=1 GSR_BWD_SEG_ITEM=$it python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_sweep.txt
  done
done
