#!/bin/bash
# rocprofv3 --kernel-trace --stats over the DEFAULT bench command (the one the driver runs), summary for profiles/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $R/$O/r04_zz_def
(cd /tmp && timeout 800 rocprofv3 --kernel-trace --stats -d $R/$O/r04_zz_def -o p -- python $R/bench.py --no-cpu-baseline > $R/$O/r04_zz_default_bench.json 2> $R/$O/r04_zz_default_bench.err)
DB=$(find $R/$O/r04_zz_def -name "*.db" | head -1)
python tools/rocpd_kernel_stats.py $DB > $O/r04_zz_default_bench_kernel_stats.md 2>&1
find $R/$O/r04_zz_def -name "*kernel_stats.csv" | head -1 | xargs -r head -25 > $O/r04_zz_default_bench_rocprof_stats_head.csv
rm -rf $R/$O/r04_zz_def
head -14 $O/r04_zz_default_bench_kernel_stats.md
python -c "
import json; d=json.load(open('$O/r04_zz_default_bench.json')); print(round(d['value'],1), d['stage_ms'], d['roofline']['achieved'], d['roofline']['frac'])"
