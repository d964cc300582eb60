#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_thermo_parity.py tests/test_gpu_campaigns.py tests/test_gpu_shims.py -x -q -m gpu > $O/sc_tests.log 2>&1; rc=$?; tail -3 $O/sc_tests.log; [ $rc -eq 0 ] || exit 1
for opt in "" "short_chain=0"; do echo "== options: $opt"; MSSPE_PROBE_OPTIONS=$opt timeout -k 10 200 python tools/perf_small_pool.py 1225 2000 2800 2>&1 | grep "counts+bitmap"; done
