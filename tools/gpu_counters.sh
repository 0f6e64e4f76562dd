#!/bin/bash
# PMC passes for the bench line's `roofline.traffic` / `frac_counter` / `valu_issue_frac` on the four bench workloads
# (headline, deep tiles, 6 M Gaussians, synth-v2): rocprofv3 --kernel-trace --pmc <group> over `bench.py --train-only`, one group per
# pass (FETCH_SIZE and WRITE_SIZE never together, never with another trace domain), then tools/collect_counters.py.
#   gpurun --timeout 1500 -- 'bash tools/gpu_counters.sh r04_x'
TAG=${1:-counters}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
MD=$R/$O/${TAG}_counters.md
echo "# ${TAG}: HBM traffic and VALU instructions per stage (rocprofv3 PMC passes over bench.py --train-only --steps 4 --warmup 1 --prewarm 0)" > $MD
echo >> $MD
pass() { # dir, counters..., -- bench args
  d=$1; shift
  mkdir -p $d
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc "$@" > /dev/null 2>&1)
  find $d -name "*.db" | head -1
}
run() { # key, bench args...
  key=$1; shift
  F=$(pass $R/$O/${TAG}_f FETCH_SIZE -d $R/$O/${TAG}_f -o p -- python $R/bench.py --train-only --steps 4 --warmup 1 --prewarm 0 "$@")
  W=$(pass $R/$O/${TAG}_w WRITE_SIZE -d $R/$O/${TAG}_w -o p -- python $R/bench.py --train-only --steps 4 --warmup 1 --prewarm 0 "$@")
  S=$(pass $R/$O/${TAG}_s SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $R/$O/${TAG}_s -o p -- python $R/bench.py --train-only --steps 4 --warmup 1 --prewarm 0 "$@")
  python tools/collect_counters.py --key "$key" --fetch "$F" --write "$W" --sq "$S" --md $MD \
    --note "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes (separate runs) over bench.py --train-only, tools/gpu_counters.sh $TAG; FETCH x2 on the two streaming kernels" > /dev/null
  rm -rf $R/$O/${TAG}_f $R/$O/${TAG}_w $R/$O/${TAG}_s
}
run "synth-v1:1000000:1920x1080:s0=0.01"
run "synth-v1:1000000:1920x1080:s0=0.05" --s0 0.05
run "synth-v1:6000000:1920x1080:s0=0.01" --gaussians 6000000
run "synth-v2:1000000:1920x1080" --scene v2
cp profiles/traffic_latest.json $O/${TAG}_traffic_latest.json
cat $MD | head -60
