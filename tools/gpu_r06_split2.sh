#!/bin/bash
# Split K1, second look: the colour kernel as a few persistent waves per SIMD (GSR_K1_SIDE_BLOCKS) and on a low-priority
# stream (GSR_K1_SIDE_PRIO), same-box A/B against the fused kernel + kernel timelines.
TAG=${1:-r06_s2}
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
for rep in 1 2; do
  echo "fused            $(GSR_K1_SPLIT=0 line)"  | tee -a $O/${TAG}_ab.txt
  echo "split full       $(GSR_K1_SPLIT=1 line)"  | tee -a $O/${TAG}_ab.txt
  echo "split prio       $(GSR_K1_SPLIT=1 GSR_K1_SIDE_PRIO=1 line)"  | tee -a $O/${TAG}_ab.txt
  echo "split 512        $(GSR_K1_SPLIT=1 GSR_K1_SIDE_BLOCKS=512 line)"  | tee -a $O/${TAG}_ab.txt
  echo "split 1024       $(GSR_K1_SPLIT=1 GSR_K1_SIDE_BLOCKS=1024 line)"  | tee -a $O/${TAG}_ab.txt
  echo "split 1024 prio  $(GSR_K1_SPLIT=1 GSR_K1_SIDE_BLOCKS=1024 GSR_K1_SIDE_PRIO=1 line)"  | tee -a $O/${TAG}_ab.txt
  echo "split 2048 prio  $(GSR_K1_SPLIT=1 GSR_K1_SIDE_BLOCKS=2048 GSR_K1_SIDE_PRIO=1 line)"  | tee -a $O/${TAG}_ab.txt
done
prof() { # name, rocprof args ... -- bench args
  name=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1)
  find $R/$O/$name -name "*.db" | head -1
}
export GSR_K1_SPLIT=1
for m in "512 0" "1024 1"; do
  set -- $m
  export GSR_K1_SIDE_BLOCKS=$1 GSR_K1_SIDE_PRIO=$2
  DB=$(prof ${TAG}_kt$1 --kernel-trace --stats -d $R/$O/${TAG}_kt$1 -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --prewarm 50)
  python tools/rocpd_timeline.py $DB 2 > $O/${TAG}_blocks$1_prio$2_timeline.md 2>&1
  rm -rf $O/${TAG}_kt$1
  head -16 $O/${TAG}_blocks$1_prio$2_timeline.md
done
