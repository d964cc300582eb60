#!/bin/bash
# kernel statistics of stage A at 10,000 x 30 kb -> gpurun_out/r03/sa_kernel_stats.csv (run on the GPU box)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
rm -rf /tmp/prof_sa
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sa -o sa -- python3 tools/perf_stage_a.py ${1:-10000} ${2:-30000} > gpurun_out/r03/sa_prof.log 2>&1 || exit 1
find /tmp/prof_sa -name '*kernel_stats.csv' -exec cp {} gpurun_out/r03/sa_kernel_stats.csv \;
find /tmp/prof_sa -name '*kernel_trace.csv' -exec cp {} gpurun_out/r03/sa_kernel_trace.csv \;
grep "^dir" gpurun_out/r03/sa_prof.log
