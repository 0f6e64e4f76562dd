#!/bin/bash
# The split K1 (geometry on the caller's stream, SH colours on a second stream under the depth sort): parity subset, same-box
# A/B of the train iteration against the fused kernel (GSR_K1_SPLIT=0), kernel timeline of both.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r06_split.sh r06_s'
TAG=${1:-r06_s}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GSR_REQUIRE_REF=1
R=$GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x tests/test_gpu_parity.py tests/test_gpu_round2.py 2>&1 | tail -6 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
line() { python bench.py --train-only --steps 40 --warmup 5 --prewarm 50 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d.get('stage_ms'))"; }
for rep in 1 2 3; do
  echo "fused  $(GSR_K1_SPLIT=0 line)"  | tee -a $O/${TAG}_ab.txt
  echo "split  $(GSR_K1_SPLIT=1 line)"  | tee -a $O/${TAG}_ab.txt
done
echo "fused v2 $(GSR_K1_SPLIT=0 line --scene v2)" | tee -a $O/${TAG}_ab.txt
echo "split v2 $(GSR_K1_SPLIT=1 line --scene v2)" | tee -a $O/${TAG}_ab.txt
echo "fused 6M $(GSR_K1_SPLIT=0 line --gaussians 6000000 --prewarm 10 --steps 20)" | tee -a $O/${TAG}_ab.txt
echo "split 6M $(GSR_K1_SPLIT=1 line --gaussians 6000000 --prewarm 10 --steps 20)" | tee -a $O/${TAG}_ab.txt
prof() { # name, rocprof args ... -- bench args
  name=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1)
  find $R/$O/$name -name "*.db" | head -1
}
for m in 1 0; do
  export GSR_K1_SPLIT=$m
  DB=$(prof ${TAG}_kt$m --kernel-trace --stats -d $R/$O/${TAG}_kt$m -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --prewarm 50)
  python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_split${m}_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_split${m}_kernel_stats.md 2>&1
  rm -rf $O/${TAG}_kt$m
done
unset GSR_K1_SPLIT
tail -30 $O/${TAG}_split1_kernel_stats.md
