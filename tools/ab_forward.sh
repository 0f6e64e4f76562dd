#!/bin/bash
# Development tool: same-box A/B of forward builds.  usage (on the GPU box): bash tools/ab_forward.sh "<probe args>" variant ...
# ("-" = the in-tree library, otherwise build_variants/libgsr_<variant>.so); every configuration runs twice, interleaved.
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset GSR_LIBRARY_PATH; else export GSR_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_variants/libgsr_$v.so; fi
  timeout 300 python tools/forward_probe.py $ARGS 2>&1 | tail -1
done; done
