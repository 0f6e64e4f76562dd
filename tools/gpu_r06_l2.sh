#!/bin/bash
TAG=${1:-r06_l2}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
python tools/l2_trace.py --steps 200
python tools/l2_trace.py --steps 200 --l0
python tools/l2_trace.py --steps 200 --c3
python tools/l2_trace.py --steps 200 --c3 --profile $O/${TAG}_c3_hostprof.txt
for mode in l2 c3; do
  extra=""; [ $mode = c3 ] && extra="--c3"
  mkdir -p $R/$O/${TAG}_$mode
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$O/${TAG}_$mode -o p -- python $R/tools/l2_trace.py --steps 20 $extra > /dev/null 2>&1)
  DB=$(find $R/$O/${TAG}_$mode -name "*.db" | head -1)
  python tools/rocpd_timeline.py $DB 3 > $O/${TAG}_${mode}_timeline.md 2>&1
  rm -rf $R/$O/${TAG}_$mode
done
cat $O/${TAG}_c3_timeline.md
