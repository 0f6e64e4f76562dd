#!/bin/bash
# Fine checkpoints, second session: the whole -m gpu suite on the new defaults, the defaults against the previous commit's library
# on the same box over the sweep's workloads, the bench line (C3 entries), and a second look at the deep-tile view.
TAG=${1:-r06_m2}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
unset GSR_REQUIRE_REF
HEAD_LIB=$GRAFT_REPO_ROOT/build_variants/libgsr_head.so
for cfg in "--gaussians 1000000" "--gaussians 500000" "--gaussians 2000000" "--gaussians 3000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" "--scene v2 --gaussians 3000000" \
           "--width 400 --height 400 --gaussians 1000000" "--width 640 --height 640 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 800 --height 800 --gaussians 3000000" \
           "--width 1920 --height 1080 --gaussians 1000000" "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000" "--width 1920 --height 1080 --scene v2 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_defaults.txt
  echo "head   $(GSR_LIBRARY_PATH=$HEAD_LIB python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
  echo "new    $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
done
cfg="--width 1920 --height 1080 --gaussians 1000000 --s0 0.05"
echo "== deep again: $cfg" | tee -a $O/${TAG}_defaults.txt
for it in 0 2 0 2; do
  echo "new ck4 it=$it $(GSR_CK_CHUNKS=4 GSR_BWD_SEG_ITEM=$it python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_defaults.txt
done
python bench.py --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
GSR_LIBRARY_PATH=$HEAD_LIB python bench.py --no-cpu-baseline > $O/${TAG}_bench_head.json 2>> $O/${TAG}_bench.err
python - <<'PY'
import json
for n in ("gpurun_out/%s_bench.json" % "$TAG", "gpurun_out/%s_bench_head.json" % "TAGX"):
    pass
PY
