#!/bin/bash
# The bench lines of a measurement session alone (tools/gpu_session.sh runs them after the PMC passes they quote).
TAG=${1:-lines}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout 300 python bench.py --steps 100 --s0 0.05 --no-cpu-baseline > $O/${TAG}_bench_deep_s005.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --steps 50 --gaussians 6000000 --no-cpu-baseline > $O/${TAG}_bench_6m.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_views8.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-view-pipeline > $O/${TAG}_bench_views8_serial.json 2>> $O/${TAG}_bench.err
GSR_TILE_BOUNDS=alpha timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_alpha_bounds.json 2>> $O/${TAG}_bench.err
GSR_FAST_EXP=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_fast_exp.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --steps 100 --force-exchange --no-cpu-baseline > $O/${TAG}_bench_forced_exchange.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --scene v2 --steps 100 --no-cpu-baseline > $O/${TAG}_bench_v2.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --scene v2 --gaussians 6000000 --steps 30 --no-cpu-baseline > $O/${TAG}_bench_v2_6m.json 2>> $O/${TAG}_bench.err
grep -v amdgpu.ids $O/${TAG}_bench.err | tail -5
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/${TAG}_bench*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v * 1e3, 1) for k, v in d.get("stage_ms", {}).items()},
              {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "frac", "frac_8d", "frac_counter", "valu_issue_frac")})
    except Exception as e:
        print(f, "unreadable", e)
PY
