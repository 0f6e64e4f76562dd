#!/bin/bash
# Fine checkpoints, fourth session: the classes that keep 512 positions + 8 slots must cost nothing against the previous
# commit's library (three alternations per workload), then the segment parity tests on the compact slot layout.
TAG=${1:-r06_m4}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
HEAD_LIB=$GRAFT_REPO_ROOT/build_variants/libgsr_head.so
for cfg in "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000" "--width 640 --height 640 --gaussians 2000000" "--gaussians 1000000" "--scene v2 --gaussians 1000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_defaults.txt
  for rep in 1 2 3; do
    echo "head   $(GSR_LIBRARY_PATH=$HEAD_LIB python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
    echo "new    $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
  done
done
export GSR_REQUIRE_REF=1
timeout 1500 python -m pytest tests -m gpu -q -x -k "list_segments or deep_translucent or edit_loop_workload or three_way_parity or deep" 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
