#!/bin/bash
# A closing session in ~13 GPU-minutes (tools/gpu_final.sh is the long form): the -m gpu suite, smoke(), the PMC passes (kernel
# sources changed -> profiles/traffic_latest.json must be re-measured), six bench lines, the headline's kernel table + timeline
# and rocprofv3's statistics of the default bench command itself.
#   gpurun --timeout 1500 -- 'bash tools/gpu_close.sh r05_zzzz'          (COUNTERS=0: host-side changes only, ~9 minutes)
TAG=${1:-close}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1
R=$GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/${TAG}_smoke.txt
if [ "${COUNTERS:-1}" = "1" ]; then bash tools/gpu_counters.sh ${TAG} > $O/${TAG}_counters.log 2>&1; fi
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout 300 python bench.py --steps 100 --s0 0.05 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_deep_s005.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --steps 50 --gaussians 6000000 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_6m.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --scene v2 --steps 100 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_v2.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_views8.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --no-view-pipeline > $O/${TAG}_bench_views8_serial.json 2>> $O/${TAG}_bench.err
prof() { # name, rocprof args ... -- bench args
  name=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 400 rocprofv3 "$@" > /dev/null 2>&1)
  find $R/$O/$name -name "*.db" | head -1
}
DB=$(prof ${TAG}_kt --kernel-trace --stats -d $R/$O/${TAG}_kt -o p -- python $R/bench.py --train-only --steps 10 --warmup 2 --prewarm 50)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_kernel_stats.md 2>&1; python tools/rocpd_timeline.py $DB 2 >> $O/${TAG}_kernel_stats.md 2>&1
DB=$(prof ${TAG}_ktb --kernel-trace --stats -d $R/$O/${TAG}_ktb -o p -- python $R/bench.py --no-cpu-baseline --no-extra-configs)
python tools/rocpd_kernel_stats.py $DB > $O/${TAG}_default_bench_kernel_stats.md 2>&1
rm -rf $O/${TAG}_kt $O/${TAG}_ktb
tail -3 $O/${TAG}_pytest.txt; tail -2 $O/${TAG}_smoke.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/${TAG}_bench*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v * 1e3, 1) for k, v in d.get("stage_ms", {}).items()},
              {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "frac", "frac_counter", "valu_issue_frac", "traffic")})
        ex = d.get("extra_configs") or {}
        for k in ("views8_one_gpu", "persistent_rows_1M_1080p"):
            if k in ex:
                print("   ", k, {a: (b if not isinstance(b, dict) else {x: round(y, 1) for x, y in b.items() if isinstance(y, float)}) for a, b in ex[k].items() if a not in ("what",)})
    except Exception as e:
        print(f, "unreadable", e)
PY
head -8 $O/${TAG}_kernel_stats.md
