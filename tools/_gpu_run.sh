set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3b_pytest.txt
cat gpurun_out/r3b_pytest.txt
timeout 300 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err
tail -c 1500 gpurun_out/r3b_bench.json
timeout 300 python bench.py --steps 50 --s0 0.05 --no-cpu-baseline > gpurun_out/r3b_bench_deep.json 2>> gpurun_out/r3b_bench.err
tail -c 1200 gpurun_out/r3b_bench_deep.json
timeout 300 python bench.py --steps 50 --force-exchange --no-cpu-baseline > gpurun_out/r3b_bench_forced_exchange.json 2>> gpurun_out/r3b_bench.err
tail -c 1500 gpurun_out/r3b_bench_forced_exchange.json
prof() { # name, bench args
  name=$1; shift
  mkdir -p $R/gpurun_out/$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$name -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1)
  DB=$(find $R/gpurun_out/$name -name "*.db" | head -1)
  python tools/rocpd_kernel_stats.py $DB > gpurun_out/${name}_kernel_stats.md 2>&1
  python tools/rocpd_timeline.py $DB -6 >> gpurun_out/${name}_kernel_stats.md 2>&1
  rm -rf $R/gpurun_out/$name
}
prof r3b
prof r3b_forced --force-exchange
cat gpurun_out/r3b_kernel_stats.md | head -64
tail -45 gpurun_out/r3b_forced_kernel_stats.md
