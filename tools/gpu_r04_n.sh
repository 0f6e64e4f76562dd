#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
prof() { name=$1; shift; mkdir -p $R/$O/$name; (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1); find $R/$O/$name -name "*.db" | head -1; }
for v in 5 0; do
  export GSR_BWD_SEG=$v
  DB=$(prof r04_n_kt$v --kernel-trace --stats -d $R/$O/r04_n_kt$v -o p -- python $R/bench.py --train-only --steps 10 --warmup 2)
  echo "== GSR_BWD_SEG=$v"; python tools/rocpd_kernel_stats.py $DB | grep -E "blend_backward|backward_worklist|blend_forward|calls"
  rm -rf $O/r04_n_kt$v
done
