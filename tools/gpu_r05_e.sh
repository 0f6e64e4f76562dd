#!/bin/bash
# round 5, call E: the whole GPU suite with the new tests (GSR_REQUIRE_REF=1), synth-v2 bench lines and its heuristics sweep
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export GSR_REQUIRE_REF=1
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/r05e_pytest.txt; tail -8 $O/r05e_pytest.txt
unset GSR_REQUIRE_REF
timeout 300 python bench.py --scene v2 --no-cpu-baseline --steps 200 > $O/r05e_bench_v2.json 2> $O/r05e_bench_v2.err
timeout 300 python bench.py --scene v2 --gaussians 6000000 --no-cpu-baseline --steps 50 > $O/r05e_bench_v2_6m.json 2>> $O/r05e_bench_v2.err
timeout 900 python tools/ab_variants.py --steps 200 --scene=v2 base ck8=GSR_CK_CHUNKS=8 ck4=GSR_CK_CHUNKS=4 w3=GSR_BLEND_WAVES_PER_SIMD=3 w2=GSR_BLEND_WAVES_PER_SIMD=2 h0=GSR_BWD_HALVES=0 h8=GSR_BWD_HALVES=8 h14=GSR_BWD_HALVES=14 baseb > $O/r05e_ab_v2.txt 2>&1; cat $O/r05e_ab_v2.txt
python - <<PY
import json
for f in ("$O/r05e_bench_v2.json", "$O/r05e_bench_v2_6m.json"):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],4), round(d["forward_ms"],4), {k: round(v*1e3,1) for k,v in d["stage_ms"].items()}, d["config"]["num_rendered"], d["config"]["visible"], {k: round(v,3) if isinstance(v,float) else v for k,v in d["roofline"].items() if k in ("kernel","frac","frac_8d","valu_issue_frac")})
    except Exception as e:
        print(f, "unreadable", e)
PY
