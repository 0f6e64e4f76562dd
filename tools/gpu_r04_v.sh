#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python tools/host_probe.py > $O/r04_v_host.txt 2>&1
cat $O/r04_v_host.txt
