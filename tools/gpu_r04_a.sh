#!/bin/bash
# Round 4, call a: baseline of the sources as round 3 left them -- 6 M Gaussians kernel table + timeline,
# the headline bench line of this box, and the PyTorch-CPU leg measured in full once (all tiles; item 8).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
TAG=r04_a
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
mkdir -p $R/$O/${TAG}_kt6
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${TAG}_kt6 -o p -- python $R/bench.py --steps 10 --warmup 2 --gaussians 6000000 --no-cpu-baseline > $R/$O/${TAG}_bench_6m_prof.json 2>/dev/null)
DB=$(find $R/$O/${TAG}_kt6 -name "*.db" | head -1)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_6m_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB -6 >> $O/${TAG}_6m_kernel_stats.md 2>&1
rm -rf $O/${TAG}_kt6
nproc > $O/${TAG}_torch_cpu_full.txt; grep -m1 "model name" /proc/cpuinfo >> $O/${TAG}_torch_cpu_full.txt
for th in 32 128; do
  OMP_NUM_THREADS=$th MKL_NUM_THREADS=$th timeout 500 python -m oracle.torch_cpu 1000000 1920 1080 0.01 0 8 1 $th >> $O/${TAG}_torch_cpu_full.txt 2>&1
done
tail -c 400 $O/${TAG}_bench.json; tail -30 $O/${TAG}_6m_kernel_stats.md; cat $O/${TAG}_torch_cpu_full.txt
