#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_thermo_parity.py -x -q -m gpu -k "not long_oligo and not hairpin_wave" > $O/small_tests.log 2>&1; rc=$?; tail -3 $O/small_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/perf_small_pool.py 1225 2000 4096 16384 > $O/small2.log 2>&1; grep -v amdgpu.ids $O/small2.log
