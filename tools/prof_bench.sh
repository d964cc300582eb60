#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
rm -rf /tmp/prof_b
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 bench.py --steps 2 --warmup 1 --no-stage-a --no-stage-b --no-cpu-baseline > gpurun_out/r03/b_prof.log 2>&1 || exit 1
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} gpurun_out/r03/b_kernel_stats.csv \;
tail -1 gpurun_out/r03/b_prof.log | cut -c1-300
