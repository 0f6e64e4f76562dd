#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2; do
for mode in "" "--keep-gc"; do
timeout 300 python bench.py --steps 4000 --no-cpu-baseline --no-extra-configs --train-only $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gc ' + ('on ' if '$mode' else 'off'), round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['step_ms_gpu'].items() if k!='note'})"
done; done
for mode in "" "--keep-gc"; do
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-extra-configs --train-only $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('200 steps, gc ' + ('on ' if '$mode' else 'off'), round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['step_ms_gpu'].items() if k!='note'})"
done
