#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/r04_k1b_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/r04_k1b_pytest.txt
cat $O/r04_k1b_pytest.txt
