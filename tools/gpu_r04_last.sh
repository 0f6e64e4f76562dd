#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -q -x -k "bench" 2>&1 | tail -3
timeout 300 python bench.py --steps 100 --force-exchange --no-cpu-baseline > $O/r04_last_fx.out 2> $O/r04_last_fx.err; echo "stdout lines: $(wc -l < $O/r04_last_fx.out)"; grep -c "RCCL version" $O/r04_last_fx.err
timeout 300 python bench.py --steps 100 --force-exchange --no-cpu-baseline > $O/r04_last_merged.out 2>&1; grep -n "RCCL version\|^{" $O/r04_last_merged.out | cut -c1-60
timeout 600 python bench.py --steps 20000 --no-cpu-baseline --no-extra-configs --prewarm 0 > $O/r04_last_soak.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r04_last_soak.json')); print('soak', d['value'], d['ms_per_step'], d['step_ms_gpu'])"
