#!/bin/bash
# Fuzz sweeps on the final build (checkpoint table): synth-v2 seeds 698 .. , cube seeds 612 ..
TAG=${1:-r06_f3}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 800 python tools/fuzz_v2.py --first 698 --count 900 --seconds 540 --judge > $O/${TAG}_fuzz_v2.txt 2>&1
tail -1 $O/${TAG}_fuzz_v2.txt
timeout 400 python tools/fuzz_parity.py --first 612 --count 2000 --seconds 240 --judge > $O/${TAG}_fuzz_parity.txt 2>&1
tail -1 $O/${TAG}_fuzz_parity.txt
