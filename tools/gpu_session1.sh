#!/bin/bash
# round 4, first GPU session: the new configs[2] tests + campaign hashes, small-pool baseline with its kernel trace
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_campaigns.py tests/test_gpu_group.py -x -q -m gpu > $O/s1_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/s1_tests.log
timeout -k 10 200 python tools/perf_small_pool.py 2000 1225 4096 > $O/s1_small.log 2>&1; echo "small rc $?"; cat $O/s1_small.log
rm -rf /tmp/prof_s
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python3 tools/perf_small_pool.py 2000 --reps 20 > $O/s1_small_prof.log 2>&1; echo "prof rc $?"
find /tmp/prof_s -name '*kernel_stats.csv' -exec cp {} $O/s1_small_kernel_stats.csv \;
head -30 $O/s1_small_kernel_stats.csv | cut -c1-160
