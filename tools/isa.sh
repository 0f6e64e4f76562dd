#!/bin/bash
# Development tool: device ISA of one translation unit with the product's flags, and the register / LDS / scratch figures of
# every kernel whose mangled name contains PATTERN.   tools/isa.sh gsr_blend blend_backward_kernelILi0ELb0ELb0  [-> /tmp/isa/<file>.s]
set -euo pipefail
cd "$(dirname "$0")/.."
F=${1:?translation unit without .hip}; PAT=${2:-.}
EXTRA=""; [ "$F" = gsr_blend ] && EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-atomic-optimizer-strategy=None"
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize \
  $EXTRA ${ISA_FLAGS:-} -S --cuda-device-only gaussianeditor_amd/csrc/$F.hip -o /tmp/isa/$F.s 2>/dev/null
awk -v pat="$PAT" '/^_Z.*:/{name=$1} /; NumVgprs:|; ScratchSize:|; LDSByteSize:|; Occupancy:|; TotalNumSgprs:/{ if (name ~ pat) print name, $0 }' /tmp/isa/$F.s
