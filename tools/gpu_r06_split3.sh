#!/bin/bash
# Split K1, third look: a throttled colour kernel (few persistent blocks) under the WHOLE sort + bin chain, joined in front of K6.
TAG=${1:-r06_s3}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
line() { python bench.py --train-only --steps 40 --warmup 5 --prewarm 50 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])"; }
export GSR_K1_JOIN_LATE=1
for rep in 1 2; do
  echo "fused            $(GSR_K1_SPLIT=0 line)"  | tee -a $O/${TAG}_ab.txt
  for b in 64 128 192 256 384 512; do
    echo "split late $b   $(GSR_K1_SPLIT=1 GSR_K1_SIDE_BLOCKS=$b line)"  | tee -a $O/${TAG}_ab.txt
  done
done
prof() { # name, rocprof args ... -- bench args
  name=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1)
  find $R/$O/$name -name "*.db" | head -1
}
export GSR_K1_SPLIT=1
for b in 128 256; do
  export GSR_K1_SIDE_BLOCKS=$b
  DB=$(prof ${TAG}_kt$b --kernel-trace --stats -d $R/$O/${TAG}_kt$b -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --prewarm 50)
  python tools/rocpd_timeline.py $DB 2 > $O/${TAG}_late_blocks${b}_timeline.md 2>&1
  rm -rf $O/${TAG}_kt$b
  head -26 $O/${TAG}_late_blocks${b}_timeline.md
done
