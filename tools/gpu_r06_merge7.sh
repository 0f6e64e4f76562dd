#!/bin/bash
# Fine checkpoints, closing table: the shipped defaults against the previous commit's library over the fifteen workloads
# (one process per row, same box), seed 619 of the synth-v2 fuzz five times on each library, then the whole -m gpu suite.
TAG=${1:-r06_m9}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
HEAD_LIB=$GRAFT_REPO_ROOT/build_variants/libgsr_head.so
for cfg in "--gaussians 500000" "--gaussians 1000000" "--gaussians 2000000" "--gaussians 3000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" "--scene v2 --gaussians 3000000" \
           "--width 400 --height 400 --gaussians 1000000" "--width 640 --height 640 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 800 --height 800 --gaussians 3000000" \
           "--width 1920 --height 1080 --gaussians 1000000" "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000" "--width 1920 --height 1080 --scene v2 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_defaults.txt
  echo "head   $(GSR_LIBRARY_PATH=$HEAD_LIB python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
  echo "new    $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
done
export GSR_REQUIRE_REF=1
for i in 1 2 3 4 5; do
  echo "head: $(GSR_LIBRARY_PATH=$HEAD_LIB timeout 120 python tools/fuzz_v2.py --only 619 --judge 2>&1 | grep 'dL_drotations  vs\|(619')" | tee -a $O/${TAG}_seed619.txt
  echo "new:  $(timeout 120 python tools/fuzz_v2.py --only 619 --judge 2>&1 | grep 'dL_drotations  vs\|(619')" | tee -a $O/${TAG}_seed619.txt
done
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
