#!/bin/bash
# Development / measurement session on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2700 -- 'bash tools/gpu_session.sh r04_s'
# Writes everything under gpurun_out/<tag>_*: the -m gpu suite, smoke(), the bench lines (headline with cpu_baseline and
# extra_configs, deep tiles, 6 M Gaussians, the fixed 8-view batch with and without view pipelining, the two opt-in flags, the
# forced one-rank exchange), rocprofv3 kernel statistics + timelines for the three workloads, the PMC passes
# (tools/gpu_counters.sh: FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU in separate runs, never combined with another trace
# domain) BEFORE the bench lines that quote them, the other BASELINE configurations and the densification surgery.
TAG=${1:-session}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1  # a missing oracle/_ref build FAILS the reference-backed tests instead of skipping them
R=$GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
bash tools/gpu_counters.sh ${TAG} > $O/${TAG}_counters.log 2>&1
bash tools/gpu_bench_lines.sh ${TAG} > $O/${TAG}_bench_lines.txt 2>&1
prof() { # name, rocprof args ... -- bench args
  name=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1)
  find $R/$O/$name -name "*.db" | head -1
}
DB=$(prof ${TAG}_kt --kernel-trace --stats -d $R/$O/${TAG}_kt -o p -- python $R/bench.py --train-only --steps 10 --warmup 2)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_kernel_stats.md 2>&1
DB=$(prof ${TAG}_ktd --kernel-trace --stats -d $R/$O/${TAG}_ktd -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --s0 0.05)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_deep_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_deep_kernel_stats.md 2>&1
DB=$(prof ${TAG}_kt6 --kernel-trace --stats -d $R/$O/${TAG}_kt6 -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --gaussians 6000000)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_6m_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_6m_kernel_stats.md 2>&1
DB=$(prof ${TAG}_ktx --kernel-trace --stats -d $R/$O/${TAG}_ktx -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --force-exchange)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_forced_exchange_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_forced_exchange_kernel_stats.md 2>&1
rm -rf $O/${TAG}_kt $O/${TAG}_ktd $O/${TAG}_kt6 $O/${TAG}_ktx
timeout 300 python tools/touched_fraction.py > $O/${TAG}_touched_fraction.txt 2>&1
timeout 600 python tools/bench_configs.py > $O/${TAG}_other_configs.txt 2>&1
timeout 300 python tools/bench_densify.py > $O/${TAG}_densify.txt 2>&1
timeout 120 python tools/bench_binning.py --oracle > $O/${TAG}_binning.txt 2>&1
timeout 120 python tools/bench_binning.py --s0 0.05 >> $O/${TAG}_binning.txt 2>&1
timeout 120 python tools/bench_binning.py --width 512 --height 512 >> $O/${TAG}_binning.txt 2>&1
timeout 120 python tools/bench_binning.py --gaussians 6000000 >> $O/${TAG}_binning.txt 2>&1
timeout 300 python tools/pipeline_probe.py > $O/${TAG}_pipeline.txt 2>&1
GSR_BLEND_WAVES_PER_SIMD=2 timeout 300 python tools/pipeline_probe.py >> $O/${TAG}_pipeline.txt 2>&1
tail -3 $O/${TAG}_pytest.txt; cat $O/${TAG}_smoke.txt | tail -2; tail -c 600 $O/${TAG}_bench.json
