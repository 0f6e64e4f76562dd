#!/bin/bash
# round 5, call C: which of the forward's item changes costs / pays (same-box A/B, smoke-checked)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
timeout 1200 python tools/ab_variants.py --smoke --steps 200 r4@r4 base bg0@bg0 est0@est0 r4like@r4like stream@stream r4b@r4 baseb bg0b@bg0 est0b@est0 r4likeb@r4like streamb@stream > $O/r05c_ab.txt 2>&1; cat $O/r05c_ab.txt
