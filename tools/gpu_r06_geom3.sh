#!/bin/bash
# The fine-in-front checkpoint table as the one default: the whole -m gpu suite, the closing table against the library of
# commit 75be983 (build_variants/libgsr_head.so: 512 x 8 from a mean list of 2 048) over twenty views, a fuzz sweep.
TAG=${1:-r06_o3}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
unset GSR_REQUIRE_REF
HEAD_LIB=$GRAFT_REPO_ROOT/build_variants/libgsr_head.so
for cfg in "--gaussians 500000" "--gaussians 1000000" "--gaussians 2000000" "--gaussians 3000000" "--gaussians 6000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" "--scene v2 --gaussians 3000000" \
           "--width 256 --height 256 --gaussians 300000" "--width 256 --height 256 --gaussians 1000000" "--width 400 --height 400 --gaussians 1000000" "--width 400 --height 400 --gaussians 2000000" \
           "--width 640 --height 640 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 800 --height 800 --gaussians 3000000" "--width 800 --height 800 --gaussians 6000000" \
           "--width 1920 --height 1080 --gaussians 1000000" "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000" "--width 1920 --height 1080 --scene v2 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_defaults.txt
  echo "head   $(GSR_LIBRARY_PATH=$HEAD_LIB python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
  echo "new    $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
done
export GSR_REQUIRE_REF=1
timeout 500 python tools/fuzz_v2.py --first 0 --count 700 --seconds 300 --judge > $O/${TAG}_fuzz_v2.txt 2>&1
tail -2 $O/${TAG}_fuzz_v2.txt
