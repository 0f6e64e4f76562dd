#!/bin/bash
# Where do the 3-6 % of K7's stage go on the views that keep 512 positions + 8 slots?  Kernel tables of one view, the previous
# commit's library and this one.
TAG=${1:-r06_m7}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
HEAD_LIB=$GRAFT_REPO_ROOT/build_variants/libgsr_head.so
prof() { # name -- command
  name=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o p -- "$@" > /dev/null 2>&1)
  find $R/$O/$name -name "*.db" | head -1
}
for cfg in ${CFGS:-"--width 1920 --height 1080 --gaussians 6000000" "--gaussians 1000000"}; do
  echo "== $cfg" | tee -a $O/${TAG}_kernels.txt
  DB=$(GSR_LIBRARY_PATH=$HEAD_LIB prof ${TAG}_h python $R/tools/c3_knobs.py $cfg)
  echo "head" | tee -a $O/${TAG}_kernels.txt
  python tools/rocpd_kernel_stats.py $DB 2>&1 | grep "blend\|worklist" | tee -a $O/${TAG}_kernels.txt
  rm -rf $O/${TAG}_h
  DB=$(prof ${TAG}_n python $R/tools/c3_knobs.py $cfg)
  echo "new" | tee -a $O/${TAG}_kernels.txt
  python tools/rocpd_kernel_stats.py $DB 2>&1 | grep "blend\|worklist" | tee -a $O/${TAG}_kernels.txt
  rm -rf $O/${TAG}_n
done
