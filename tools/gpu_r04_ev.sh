#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python bench.py --steps 3000 --no-cpu-baseline --no-extra-configs --train-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['step_ms_gpu'].items() if k!='note'})"
done
timeout 120 python - <<'PY'
import time, torch
torch.cuda.init(); s=torch.cuda.current_stream()
worst=[]
for rep in range(3):
    evs=[]; mx=0; t_all=time.perf_counter()
    for i in range(6000):
        t=time.perf_counter(); e=torch.cuda.Event(enable_timing=True); e.record(s); dt=time.perf_counter()-t
        evs.append(e)
        if dt>mx: mx=dt; at=i
    print(f"6000 new events recorded: total {1e3*(time.perf_counter()-t_all):.1f} ms, slowest {1e3*mx:.2f} ms at #{at}")
PY
