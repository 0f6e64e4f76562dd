#!/bin/bash
# A short closing session when only host-side code changed since the last full one (tools/gpu_final.sh): the -m gpu suite,
# smoke(), the default bench line and the two 8-view batch lines.   gpurun --timeout 1500 -- 'bash tools/gpu_close.sh r05_zzz'
TAG=${1:-close}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_views8.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --views 8 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --no-view-pipeline > $O/${TAG}_bench_views8_serial.json 2>> $O/${TAG}_bench.err
python tools/views8_probe.py --reps 1 --steps 15 --rounds 6 2>&1 | grep -E "pipelined|serial" > $O/${TAG}_views8_probe.txt
tail -3 $O/${TAG}_pytest.txt; tail -2 $O/${TAG}_smoke.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/${TAG}_bench*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v * 1e3, 1) for k, v in d.get("stage_ms", {}).items()})
        ex = d.get("extra_configs") or {}
        for k in ("views8_one_gpu", "persistent_rows_1M_1080p", "synth_v2_1M_1080p"):
            if k in ex:
                print("   ", k, {a: (b if not isinstance(b, dict) else {x: round(y, 1) for x, y in b.items() if isinstance(y, float)}) for a, b in ex[k].items() if a not in ("what", "roofline", "stage_ms")})
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/${TAG}_views8_probe.txt
