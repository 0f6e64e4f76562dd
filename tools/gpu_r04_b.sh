#!/bin/bash
# Round 4, call b (and later calls with another TAG): binning-only check against the oracle FIRST (a wrong point list hangs
# the blend kernels), then the binning-centred GPU tests, bench lines and kernel timelines at 1 M and 6 M Gaussians.
TAG=${1:-r04_b}
FULL=${2:-0}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out
: > $O/${TAG}_binning.txt
timeout 200 python tools/bench_binning.py --oracle >> $O/${TAG}_binning.txt 2>&1
timeout 200 python tools/bench_binning.py --width 512 --height 512 --oracle >> $O/${TAG}_binning.txt 2>&1
timeout 300 python tools/bench_binning.py --s0 0.05 --oracle >> $O/${TAG}_binning.txt 2>&1
timeout 300 python tools/bench_binning.py --gaussians 6000000 --oracle >> $O/${TAG}_binning.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_binning.txt
if [ $(grep -c "point list == oracle: True" $O/${TAG}_binning.txt) -ne 4 ] || grep -q CHANGED $O/${TAG}_binning.txt; then
  echo "BINNING MISMATCH: stopping before any blend kernel runs"; exit 1
fi
if [ "$FULL" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/${TAG}_pytest.txt
else
  timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_round3.py tests/test_gpu_parity.py \
    -k "grouped_binning or legacy or forward_all_stages or edge_geometries or random_configuration or depth_key_ranges or empty or 4k or headline or degenerate or nonfinite or backward_vs_oracle or apply_weights or two_streams" 2>&1 | tail -12 > $O/${TAG}_pytest.txt
fi
cat $O/${TAG}_pytest.txt
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout 300 python bench.py --steps 50 --gaussians 6000000 --no-cpu-baseline > $O/${TAG}_bench_6m.json 2>> $O/${TAG}_bench.err
prof() { name=$1; shift; mkdir -p $R/$O/$name; (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1); find $R/$O/$name -name "*.db" | head -1; }
DB=$(prof ${TAG}_kt --kernel-trace --stats -d $R/$O/${TAG}_kt -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB -6 >> $O/${TAG}_kernel_stats.md 2>&1
DB=$(prof ${TAG}_kt6 --kernel-trace --stats -d $R/$O/${TAG}_kt6 -o p -- python $R/bench.py --steps 10 --warmup 2 --gaussians 6000000 --no-cpu-baseline)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_6m_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB -6 >> $O/${TAG}_6m_kernel_stats.md 2>&1
rm -rf $O/${TAG}_kt $O/${TAG}_kt6
python - <<PY
import json
for f in ("$O/${TAG}_bench.json", "$O/${TAG}_bench_6m.json"):
    try:
        d = json.load(open(f)); print(f, round(d["value"], 1), "it/s fwd_ms", round(d["forward_ms"], 4), {k: round(v * 1e3, 1) for k, v in d["stage_ms"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
sed -n '/t_start/,$p' $O/${TAG}_kernel_stats.md | head -40
sed -n '/t_start/,$p' $O/${TAG}_6m_kernel_stats.md | head -40
