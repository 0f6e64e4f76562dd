#!/bin/bash
# Round 6: the suite on the HEAD after the sparse-backward revert, streaming stores for all of K8+K9's outputs (A/B).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1
O=gpurun_out
T=${1:-r06_h}
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/${T}_pytest.txt
tail -6 $O/${T}_pytest.txt
timeout 1200 python tools/ab_variants.py --steps 200 base k9nt@k9nt baseb k9ntb@k9nt > $O/${T}_ab.txt 2>&1
cat $O/${T}_ab.txt
timeout 600 python tools/ab_variants.py --steps 100 --scene v2 v2base v2k9nt@k9nt > $O/${T}_ab_v2.txt 2>&1
cat $O/${T}_ab_v2.txt
