set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/debug_binning.py 2>&1 | tail -14; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a_pytest.txt
cat gpurun_out/r3a_pytest.txt
timeout 300 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
tail -c 1500 gpurun_out/r3a_bench.json
timeout 300 python bench.py --steps 50 --s0 0.05 --no-cpu-baseline > gpurun_out/r3a_bench_deep.json 2>> gpurun_out/r3a_bench.err
tail -c 1200 gpurun_out/r3a_bench_deep.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3a_prof -o r3a -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && ls gpurun_out/r3a_prof | head
DB=$(find gpurun_out/r3a_prof -name "*.db" | head -1); echo DB=$DB; python tools/rocpd_kernel_stats.py $DB > gpurun_out/r3a_kernel_stats.md 2>&1 || true
python tools/rocpd_timeline.py $DB -6 >> gpurun_out/r3a_kernel_stats.md 2>&1 || true
rm -rf gpurun_out/r3a_prof
cat gpurun_out/r3a_kernel_stats.md | head -70
