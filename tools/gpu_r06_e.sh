#!/bin/bash
# Round 6: self-cleaning accumulator table + streaming loads of its rows -- suite, then same-box A/B.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1
O=gpurun_out
T=${1:-r06_e}
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/${T}_pytest.txt
tail -6 $O/${T}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python tools/ab_variants.py --steps 200 base nopersist=GSR_ACC_PERSIST=0 plain@k9plain baseb nopersistb=GSR_ACC_PERSIST=0 plainb@k9plain > $O/${T}_ab.txt 2>&1
cat $O/${T}_ab.txt
timeout 600 python tools/ab_variants.py --steps 100 --scene v2 v2base v2plain@k9plain > $O/${T}_ab_v2.txt 2>&1
cat $O/${T}_ab_v2.txt
