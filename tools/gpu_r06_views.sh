#!/bin/bash
# Where does the two-stream view pipelining lose its overlap?  Kernel trace of the 8-view batch, pipelined and serial.
TAG=${1:-r06_v}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
for mode in pipe serial; do
  extra=""; [ $mode = serial ] && extra="--no-view-pipeline"
  mkdir -p $R/$O/${TAG}_$mode
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/$O/${TAG}_$mode -o p -- python $R/bench.py --views 8 --steps 10 --warmup 5 --no-cpu-baseline --no-extra-configs $extra > $R/$O/${TAG}_${mode}_line.json 2> /dev/null)
  DB=$(find $R/$O/${TAG}_$mode -name "*.db" | head -1)
  python tools/rocpd_overlap.py $DB 4500 > $O/${TAG}_${mode}_overlap.md 2> $O/${TAG}_${mode}_overlap.err
  rm -rf $R/$O/${TAG}_$mode
done
tail -40 $O/${TAG}_pipe_overlap.md
