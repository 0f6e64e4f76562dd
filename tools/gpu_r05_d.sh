#!/bin/bash
# round 5, call D: the forward's heavy-tile split (correctness + sweep of its two knobs, same box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05d_smoke.txt 2>&1; tail -2 $O/r05d_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/r05d_pytest.txt; tail -5 $O/r05d_pytest.txt
timeout 900 python tools/ab_variants.py --steps 200 r4@r4 off=GSR_FWD_HEAVY_SPLIT=0 s2c16 s2c12=GSR_FWD_HEAVY_CHUNKS=12 s2c20=GSR_FWD_HEAVY_CHUNKS=20 s2c8=GSR_FWD_HEAVY_CHUNKS=8 s4c16=GSR_FWD_HEAVY_SPLIT=4 s4c12=GSR_FWD_HEAVY_SPLIT=4,GSR_FWD_HEAVY_CHUNKS=12 s4c20=GSR_FWD_HEAVY_SPLIT=4,GSR_FWD_HEAVY_CHUNKS=20 offb=GSR_FWD_HEAVY_SPLIT=0 > $O/r05d_ab.txt 2>&1; cat $O/r05d_ab.txt
