#!/bin/bash
# Development aid: registers, spills and LDS of every kernel in one of the package's objects.
# usage: tools/kernel_regs.sh open-msspe-design_amd/build/thal_pairs_row.o
set -euo pipefail
obj="$1"
tmp="$(mktemp -d)"
bin=/opt/rocm/lib/llvm/bin
"$bin/llvm-objcopy" --dump-section .hip_fatbin="$tmp/fat.bin" "$obj"
"$bin/clang-offload-bundler" --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat.bin" --output="$tmp/dev.o"
"$bin/llvm-readelf" --notes "$tmp/dev.o" | \
  awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.vgpr_spill_count:/{vs=$2; printf "%-80s vgpr %s (spilled %s) sgpr %s (spilled %s) lds %s scratch %s\n", n, v, vs, s, ss, l, p} /\.sgpr_spill_count:/{ss=$2} /\.group_segment_fixed_size:/{l=$2} /\.private_segment_fixed_size:/{p=$2}'
[[ "${2:-}" == "--asm" ]] && "$bin/llvm-objdump" -d "$tmp/dev.o" > "${3:-/tmp/dev.s}"
rm -rf "$tmp"
