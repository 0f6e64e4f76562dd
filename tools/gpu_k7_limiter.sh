#!/bin/bash
# What bounds K7: SQ counter passes over the blend kernels of `bench.py --train-only`, one counter
# group per rocprofv3 pass (kernel trace only, no other trace domain), per-kernel averages per dispatch.
#   gpurun -- 'bash tools/gpu_k7_limiter.sh r06_a [--scene v2]'
TAG=${1:-k7}; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
MD=$R/$O/${TAG}_k7_limiter.md
echo "# ${TAG}: SQ counters of the blend kernels (rocprofv3 --kernel-trace --pmc, one group per pass; bench.py --train-only --steps 4 --warmup 1 --prewarm 0 $*)" > $MD
GROUPS_=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_LDS"
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC"
  "SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"
)
i=0
for g in "${GROUPS_[@]}"; do
  d=$R/$O/${TAG}_pmc$i
  mkdir -p $d
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $g -d $d -o p -- python $R/bench.py --train-only --steps 4 --warmup 1 --prewarm 0 "$@" > /dev/null 2>&1)
  DB=$(find $d -name "*.db" | head -1)
  echo >> $MD; echo "## pass $i: $g" >> $MD; echo '```' >> $MD
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB blend_ >> $MD 2>&1; else echo "(no database: the pass failed)" >> $MD; fi
  echo '```' >> $MD
  rm -rf $d
  i=$((i+1))
done
cat $MD
