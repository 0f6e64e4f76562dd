#!/bin/bash
# Candidate checkpoint tables (GSR_CK_TABLE) against the shipped one.
TAG=${1:-r06_q}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
A="4,8,12,16,20,24,32,40,48,64,80,96,128,192,256"
B="2,4,6,8,10,12,16,20,24,32,48,64,96,128,256"
C="2,4,6,8,12,16,20,24,32,40,48,64,96,128,256"
D="3,6,9,12,15,18,24,30,36,48,64,80,112,160,256"
E="4,8,12,16,20,24,28,32,40,48,64,80,112,160,256"
for cfg in "--gaussians 500000" "--gaussians 1000000" "--gaussians 2000000" "--gaussians 3000000" "--gaussians 6000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" \
           "--width 400 --height 400 --gaussians 1000000" "--width 256 --height 256 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 800 --height 800 --gaussians 6000000" \
           "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_table.txt
  for t in A B C D E; do
    echo "$t  $(GSR_CK_TABLE=${!t} python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_table.txt
  done
done
