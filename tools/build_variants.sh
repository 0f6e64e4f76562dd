#!/bin/bash
# Development tool: build differently compiled copies of libgsr_hip.so into build_variants/ for same-box A/B timing
# (tools/ab_variants.py loads them through GSR_LIBRARY_PATH).  A variant is "name|file1.hip:extra flags;file2.hip:extra flags";
# files not named are compiled with the product's flags.  Usage: tools/build_variants.sh [variant ...]
set -euo pipefail
cd "$(dirname "$0")/.."
SRC=gaussianeditor_amd/csrc
OUT=build_variants
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize -fvisibility=hidden -Wall -Wextra -Wno-unused-parameter"
FILES="gsr_capi gsr_preprocess gsr_binning gsr_blend gsr_knn gsr_optim gsr_compact"
ILP="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-atomic-optimizer-strategy=None"
DEFAULT_VARIANTS=(
  "ilp_blend|gsr_blend:$ILP"
  "ilp_all|gsr_blend:$ILP;gsr_preprocess:$ILP;gsr_binning:$ILP"
)
if [ $# -gt 0 ]; then VARIANTS=("$@"); else VARIANTS=("${DEFAULT_VARIANTS[@]}"); fi
mkdir -p "$OUT"
for v in "${VARIANTS[@]}"; do
  name="${v%%|*}"; spec="${v#*|}"
  work="$OUT/obj_$name"; mkdir -p "$work"
  objs=""
  for f in $FILES; do
    extra=""
    IFS=';' read -ra parts <<< "$spec"
    for p in "${parts[@]}"; do
      if [ "${p%%:*}" = "$f" ]; then extra="${p#*:}"; fi
    done
    if [ -z "$extra" ]; then
      objs="$objs $SRC/$f.o"   # the product's own object (make -C $SRC first)
    else
      /opt/rocm/bin/hipcc $BASE $extra -c "$SRC/$f.hip" -o "$work/$f.o" &
      objs="$objs $work/$f.o"
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$SRC/gsr.map -o "$OUT/libgsr_$name.so" $objs
  echo "built $OUT/libgsr_$name.so"
done
