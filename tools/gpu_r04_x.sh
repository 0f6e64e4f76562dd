#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests/test_gpu_properties.py -q -x 2>&1 | tail -25 > gpurun_out/r04_x_pytest.txt
cat gpurun_out/r04_x_pytest.txt
