#!/bin/bash
# The round's closing session: the -m gpu suite, smoke(), the PMC passes (the csrc hash in profiles/traffic_latest.json must
# match the shipped sources), the bench lines that quote them, and kernel tables + timelines of the headline and the 6 M view.
#   gpurun --timeout 2400 -- 'bash tools/gpu_final.sh r04_zz'
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1  # a missing oracle/_ref build FAILS the reference-backed tests instead of skipping them
R=$GRAFT_REPO_ROOT
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
bash tools/gpu_counters.sh ${TAG} > $O/${TAG}_counters.log 2>&1
bash tools/gpu_bench_lines.sh ${TAG} > $O/${TAG}_bench_lines.txt 2>&1
prof() { # name, rocprof args ... -- bench args
  name=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1)
  find $R/$O/$name -name "*.db" | head -1
}
DB=$(prof ${TAG}_kt --kernel-trace --stats -d $R/$O/${TAG}_kt -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --prewarm 50)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_kernel_stats.md 2>&1
DB=$(prof ${TAG}_kt6 --kernel-trace --stats -d $R/$O/${TAG}_kt6 -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --prewarm 20 --gaussians 6000000)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_6m_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_6m_kernel_stats.md 2>&1
DB=$(prof ${TAG}_ktv --kernel-trace --stats -d $R/$O/${TAG}_ktv -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --prewarm 30 --scene v2)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_v2_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_v2_kernel_stats.md 2>&1
# the default bench command itself under the kernel trace (the dominant kernel's average must agree with the line's stage event)
DB=$(prof ${TAG}_ktb --kernel-trace --stats -d $R/$O/${TAG}_ktb -o p -- python $R/bench.py --no-cpu-baseline --no-extra-configs)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_default_bench_kernel_stats.md 2>&1
rm -rf $O/${TAG}_kt $O/${TAG}_kt6 $O/${TAG}_ktv $O/${TAG}_ktb
timeout 200 python tools/bwd_items_profile.py > $O/${TAG}_bwd_items.txt 2>&1
tail -3 $O/${TAG}_pytest.txt; tail -2 $O/${TAG}_smoke.txt; cat $O/${TAG}_bench_lines.txt; head -12 $O/${TAG}_kernel_stats.md
