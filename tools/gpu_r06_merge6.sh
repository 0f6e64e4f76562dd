#!/bin/bash
# Fine checkpoints, the rewritten work-list kernel: kernel tables against the previous commit's library, the A/B over the
# classes, then the segment parity tests.
TAG=${1:-r06_m8}
cd $GRAFT_REPO_ROOT
bash tools/gpu_r06_merge5.sh ${TAG}
bash tools/gpu_r06_merge4.sh ${TAG}
