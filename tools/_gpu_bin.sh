cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/bench_binning.py --oracle 2>&1 | tail -2
timeout 120 python tools/bench_binning.py --s0 0.05 2>&1 | tail -1
timeout 120 python tools/bench_binning.py --width 512 --height 512 2>&1 | tail -1
timeout 120 python tools/bench_binning.py --gaussians 6000000 2>&1 | tail -1
timeout 120 python tools/bench_binning.py --gaussians 20000 --width 512 --height 512 --s0 0.02 --oracle 2>&1 | tail -2
