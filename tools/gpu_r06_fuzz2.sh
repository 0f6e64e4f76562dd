#!/bin/bash
TAG=${1:-r06_f2}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
O=gpurun_out
timeout 900 python tools/fuzz_v2.py --first 620 --count 900 --seconds 600 --judge > $O/${TAG}_fuzz_v2.txt 2>&1
tail -2 $O/${TAG}_fuzz_v2.txt
