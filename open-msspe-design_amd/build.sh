#!/usr/bin/env bash
# Builds libmsspe_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# -ffp-contract=off: every double expression keeps Primer3's operation order (no FMA fusion).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
cd "$here/csrc"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math
       -Wall -Wno-unused-function)
mkdir -p "$here/build"
objs=()
for src in nn_params.cpp capi.cpp kernels_generic.hip thal_hairpin_wave.hip thal_pairs.hip thal_pairs_int.hip thal_pairs_row.hip thal_pairs_split.hip thal_pairs_wave.hip pool_sort.hip kmer_stage.hip group.hip; do
  obj="$here/build/${src%.*}.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || -n "$(find . -name '*.hpp' -newer "$obj" -print -quit)" \
        || ../../include/msspe_hip.h -nt "$obj" ]]; then
    echo "hipcc $src"
    "$HIPCC" "${FLAGS[@]}" -x hip -c "$src" -o "$obj" ${EXTRA_HIPCC_FLAGS:-}
  fi
  objs+=("$obj")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/libmsspe_hip.so" "${objs[@]}" -ldl
echo "built $here/libmsspe_hip.so"
# host side above the C ABI (C++17, no HIP): the reference-interface mirror, its CLI and test hooks
cd "$here/host"
CXX="${CXX:-g++}"
"$CXX" -O2 -std=c++17 -fPIC -Wall -pthread -shared -o "$here/libod_msspe_host.so" od_msspe.cpp c_hooks.cpp \
    -L"$here" -lmsspe_hip -Wl,-rpath,'$ORIGIN'
"$CXX" -O2 -std=c++17 -Wall -pthread -o "$here/od-msspe-hip" main.cpp od_msspe.cpp -L"$here" -lmsspe_hip -Wl,-rpath,'$ORIGIN'
mkdir -p "$here/bin"
"$CXX" -O2 -std=c++17 -Wall -o "$here/bin/ntthal-hip" ntthal_shim.cpp -L"$here" -lmsspe_hip -Wl,-rpath,'$ORIGIN/..'
"$CXX" -O2 -std=c++17 -Wall -o "$here/bin/primer3_core-hip" primer3_shim.cpp -L"$here" -lmsspe_hip -Wl,-rpath,'$ORIGIN/..'
echo "built $here/libod_msspe_host.so, $here/od-msspe-hip, bin/ntthal-hip, bin/primer3_core-hip"
