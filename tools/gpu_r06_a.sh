#!/bin/bash
# Round 6, first session: the suite on the round's first build, the headline line, K7's issue ceiling (replay of its own
# group body) and its SQ counters, and same-box A/Bs of the backward's exponential / slot variants.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GSR_REQUIRE_REF=1
O=gpurun_out
T=r06_a
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/${T}_pytest.txt
tail -4 $O/${T}_pytest.txt
timeout 300 build_variants/k7_group_replay > $O/${T}_k7_replay.txt 2>&1
timeout 300 build_variants/k7_group_replay_poly > $O/${T}_k7_replay_poly.txt 2>&1
timeout 300 build_variants/issue_rate > $O/${T}_issue_rate.txt 2>&1
cat $O/${T}_k7_replay.txt $O/${T}_k7_replay_poly.txt
timeout 1500 python tools/ab_variants.py --steps 200 base poly@polyexp slot@slotreg baseb polyb@polyexp slotb@slotreg > $O/${T}_ab.txt 2>&1
cat $O/${T}_ab.txt
timeout 600 python tools/ab_variants.py --steps 100 --scene v2 v2base v2poly@polyexp v2slot@slotreg > $O/${T}_ab_v2.txt 2>&1
cat $O/${T}_ab_v2.txt
bash tools/gpu_k7_limiter.sh ${T} > /dev/null 2>&1
bash tools/gpu_k7_limiter.sh ${T}_v2 --scene v2 > /dev/null 2>&1
tail -60 $O/${T}_k7_limiter.md
