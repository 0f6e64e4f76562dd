#!/bin/bash
# Round 6, second session: view-reuse tests, what the pieces of K7 cost (ablations on the round's base), C3 entries of the line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1
O=gpurun_out
T=r06_b
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x 2>&1 | tail -25 > $O/${T}_pytest6.txt
cat $O/${T}_pytest6.txt
timeout 1500 python tools/ab_variants.py --steps 150 base abl1=GSR_BWD_ABLATE=1 abl2=GSR_BWD_ABLATE=2 abl3=GSR_BWD_ABLATE=3 abl4=GSR_BWD_ABLATE=4 \
   abl5=GSR_BWD_ABLATE=5 abl6=GSR_BWD_ABLATE=6 baseb > $O/${T}_ablate.txt 2>&1
cat $O/${T}_ablate.txt
timeout 900 python tools/ab_variants.py --steps 60 --scene v2 v2base v2abl1=GSR_BWD_ABLATE=1 v2abl2=GSR_BWD_ABLATE=2 v2abl5=GSR_BWD_ABLATE=5 v2abl6=GSR_BWD_ABLATE=6 > $O/${T}_ablate_v2.txt 2>&1
cat $O/${T}_ablate_v2.txt
timeout 900 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
print(d["value"], d["ms_per_step"], d["stage_ms"])
for k, v in d["extra_configs"].items():
    if k.startswith("C3") or k.startswith("C5"):
        print(k, v)
PY
tail -3 $O/${T}_bench.err
