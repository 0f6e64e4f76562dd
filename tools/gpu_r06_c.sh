#!/bin/bash
# Round 6, third session: the accumulator-row backward (one memory-side request per (tile, Gaussian)) -- the suite, then the lines.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1
O=gpurun_out
T=${1:-r06_c}
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/${T}_pytest.txt
tail -6 $O/${T}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python tools/ab_variants.py --steps 200 base abl2=GSR_BWD_ABLATE=2 baseb > $O/${T}_ab.txt 2>&1
cat $O/${T}_ab.txt
timeout 600 python tools/ab_variants.py --steps 100 --scene v2 v2base v2abl2=GSR_BWD_ABLATE=2 > $O/${T}_ab_v2.txt 2>&1
cat $O/${T}_ab_v2.txt
timeout 300 python bench.py --steps 100 --s0 0.05 --no-cpu-baseline --no-extra-configs > $O/${T}_bench_deep.json 2>/dev/null
timeout 300 python bench.py --steps 50 --gaussians 6000000 --no-cpu-baseline --no-extra-configs > $O/${T}_bench_6m.json 2>/dev/null
python - <<PY
import json
for n in ("deep", "6m"):
    try:
        d = json.load(open("$O/${T}_bench_%s.json" % n)); print(n, round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v * 1e3, 1) for k, v in d["stage_ms"].items()})
    except Exception as e:
        print(n, "unreadable", e)
PY
