#!/bin/bash
# Development aid: links variants of libmsspe_hip.so that differ in the shape of the row kernel
# (threads per block x table slots) into _var/variants/<threads>_<slots>/ for A/B timing with
# tools/perf_probe.py (MSSPE_PROBE_LIB=<path>).  usage: tools/variant_build.sh "512 64" "512 56" ...
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
pkg="$root/open-msspe-design_amd"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function)
for cfg in "$@"; do
  set -- $cfg
  out="$root/_var/variants/$1_$2"
  mkdir -p "$out"
  /opt/rocm/bin/hipcc "${FLAGS[@]}" -DMSSPE_ROW_THREADS=$1 -DMSSPE_ROW_SLOTS=$2 ${3:+-D$3} -x hip -c "$pkg/csrc/thal_pairs_row.hip" -o "$out/thal_pairs_row.o"
  objs=()
  for o in "$pkg"/build/*.o; do [[ "$(basename $o)" == thal_pairs_row.o ]] || objs+=("$o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmsspe_hip.so" "${objs[@]}" "$out/thal_pairs_row.o" -ldl
  ln -sfn ../../../open-msspe-design_amd/data "$out/data"
  echo "built $out/libmsspe_hip.so"
done
