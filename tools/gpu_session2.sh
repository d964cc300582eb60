#!/bin/bash
# round 4, session 2: stage B one-fill + lane path parity, list kernels with dynamic depth, small-pool and stage-B timings
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_thermo_parity.py tests/test_gpu_campaigns.py tests/test_gpu_shims.py -x -q -m gpu > $O/s2_tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -15 $O/s2_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/perf_small_pool.py 2000 1225 4096 > $O/s2_small.log 2>&1; echo "small rc $?"; cat $O/s2_small.log
for n in 2000 32768 131072 1048576; do timeout -k 10 200 python tools/perf_stage_b.py $n >> $O/s2_stage_b.log 2>&1 || exit 1; done; cat $O/s2_stage_b.log
MSSPE_PROBE_OPTIONS=self_lane_from=0 timeout -k 10 100 python tools/perf_stage_b.py 8192 > $O/s2_stage_b_8k_lane.log 2>&1; MSSPE_PROBE_OPTIONS=self_lane_from=1000000000 timeout -k 10 100 python tools/perf_stage_b.py 8192 > $O/s2_stage_b_8k_wave.log 2>&1; MSSPE_PROBE_OPTIONS=self_lane_from=1000000000 timeout -k 10 100 python tools/perf_stage_b.py 65536 > $O/s2_stage_b_64k_wave.log 2>&1; MSSPE_PROBE_OPTIONS=self_lane_from=0 timeout -k 10 100 python tools/perf_stage_b.py 65536 > $O/s2_stage_b_64k_lane.log 2>&1
grep self_dimers $O/s2_stage_b_*k_*.log
