#!/bin/bash
# Fuzz sweeps on the fine-checkpoint build (small images with long lists are the class that changed).
TAG=${1:-r06_f}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 700 python tools/fuzz_v2.py --first 0 --count 700 --seconds 420 --judge > $O/${TAG}_fuzz_v2.txt 2>&1
tail -3 $O/${TAG}_fuzz_v2.txt
timeout 500 python tools/fuzz_parity.py --first 12 --count 600 --seconds 300 --judge > $O/${TAG}_fuzz_parity.txt 2>&1
tail -3 $O/${TAG}_fuzz_parity.txt
