#!/bin/bash
# The forward's cut for renders with a backward: the new rule (default) against the old one (the cut that is best for the
# forward alone: GSR_FWD_SPLIT_FORCE = what the old rule chose), then the whole -m gpu suite.
TAG=${1:-r06_r5}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
old_rule() { # width -> the old rule's cut
  T=$(( (($1 + 15) / 16) * (($1 + 15) / 16) )); q=$((4 * T))
  if [ $q -ge 8192 ]; then echo 1; elif [ $q -ge 4096 ]; then echo 2; else echo 4; fi
}
for wh in 256 320 400 448 512 576 640 720 800; do
  for sc in "--gaussians 500000" "--gaussians 1000000" "--scene v2 --gaussians 1000000" "--gaussians 3000000"; do
    cfg="--width $wh --height $wh $sc"
    echo "== $cfg" | tee -a $O/${TAG}_est.txt
    echo "old $(GSR_FWD_SPLIT_FORCE=$(old_rule $wh) python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_est.txt
    echo "new $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_est.txt
  done
done
export GSR_REQUIRE_REF=1
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
