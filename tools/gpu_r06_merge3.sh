#!/bin/bash
# Fine checkpoints, third session: the image-size classes (fine stride + 16 slots up to 1 024 tiles, 512 positions + 8 slots
# above) against the previous commit's library on the same box; merged items on the two workloads the fine stride lost.
TAG=${1:-r06_m3}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
HEAD_LIB=$GRAFT_REPO_ROOT/build_variants/libgsr_head.so
for cfg in "--gaussians 1000000" "--gaussians 500000" "--gaussians 2000000" "--gaussians 3000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" "--scene v2 --gaussians 3000000" \
           "--width 400 --height 400 --gaussians 1000000" "--width 640 --height 640 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 800 --height 800 --gaussians 3000000" \
           "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_defaults.txt
  echo "head   $(GSR_LIBRARY_PATH=$HEAD_LIB python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
  echo "new    $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
done
for cfg in "--scene v2 --gaussians 2000000" "--gaussians 2000000"; do
  echo "== merged items: $cfg" | tee -a $O/${TAG}_defaults.txt
  for it in 2 3 4 6; do
    echo "new it=$it $(GSR_BWD_SEG_ITEM=$it python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
  done
  echo "new ck8 $(GSR_CK_CHUNKS=8 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
  echo "new ck6 slots12 $(GSR_CK_CHUNKS=6 GSR_CK_SLOTS=12 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
done
