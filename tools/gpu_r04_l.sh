#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for ck in 1 2 4; do
  echo "== GSR_CK_CHUNKS=$ck GSR_BWD_SEG=1"
  GSR_CK_CHUNKS=$ck GSR_BWD_SEG=1 timeout 900 python -m pytest -q -m gpu tests/test_gpu_round2.py -k "three_way_parity" 2>&1 | grep -E "^E  |passed|failed|Error" | head -20
done
