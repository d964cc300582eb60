#!/bin/bash
# Development aid: links a variant of libmsspe_hip.so in which ONE source file is compiled with extra -D flags,
# into _var/variants/<name>/ (A/B timing with MSSPE_PROBE_LIB=<path>).
# usage: tools/variant_build_file.sh <name> <source file under csrc/> [-DFOO=1 ...]
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
pkg="$root/open-msspe-design_amd"
name="$1"; src="$2"; shift 2
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function)
out="$root/_var/variants/$name"
mkdir -p "$out"
stem="$(basename "${src%.*}")"
/opt/rocm/bin/hipcc "${FLAGS[@]}" "$@" -x hip -c "$pkg/csrc/$src" -o "$out/$stem.o"
objs=()
for o in "$pkg"/build/*.o; do [[ "$(basename $o)" == "$stem.o" ]] || objs+=("$o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmsspe_hip.so" "${objs[@]}" "$out/$stem.o" -ldl
ln -sfn ../../../open-msspe-design_amd/data "$out/data"
echo "built $out/libmsspe_hip.so"
