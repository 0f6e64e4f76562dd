#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python tools/pipeline_probe.py > $O/r04_i_pipeline.txt 2>&1
GSR_BLEND_WAVES_PER_SIMD=3 timeout 300 python tools/pipeline_probe.py >> $O/r04_i_pipeline.txt 2>&1
GSR_BLEND_WAVES_PER_SIMD=2 timeout 300 python tools/pipeline_probe.py >> $O/r04_i_pipeline.txt 2>&1
timeout 300 python tools/pipeline_probe.py --s0 0.05 >> $O/r04_i_pipeline.txt 2>&1
grep -v amdgpu.ids $O/r04_i_pipeline.txt
