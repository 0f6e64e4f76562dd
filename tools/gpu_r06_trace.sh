#!/bin/bash
# K12 (apply_weights): parity of the grouped inner loop, then same-box A/B of the loop variants and their ablations.
TAG=${1:-r06_t}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "weights or trace or apply" 2>&1 | tail -5 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
python tools/trace_probe.py 2>/dev/null | tee $O/${TAG}_probe.txt
for v in "$@"; do
  [ "$v" = "$TAG" ] && continue
  GSR_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_variants/libgsr_$v.so python tools/trace_probe.py 2>/dev/null | tee -a $O/${TAG}_probe.txt
done
python tools/trace_probe.py 2>/dev/null | tee -a $O/${TAG}_probe.txt
