#!/bin/bash
# Host side of the pipelined 8-view batch: which Python / native calls does the launch thread spend a step in?
TAG=${1:-r06_w}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--views', '8', '--steps', '100', '--warmup', '10', '--no-cpu-baseline', '--no-extra-configs']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats('cumulative').print_stats(60)
st.sort_stats('tottime').print_stats(45)
open('$O/${TAG}_hostprof.txt', 'w').write(s.getvalue())
" > $O/${TAG}_hostprof_line.json 2> $O/${TAG}_hostprof.err
grep -v "^$" $O/${TAG}_hostprof.txt | head -150
