#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_thermo_parity.py tests/test_gpu_campaigns.py -x -q -m gpu > $O/k1415_tests.log 2>&1; rc=$?; tail -3 $O/k1415_tests.log; [ $rc -eq 0 ] || exit 1
for k in 14 15; do MSSPE_PROBE_K=$k MSSPE_PROBE_PROFILE=1 timeout -k 10 200 python tools/perf_probe.py 16384 65536 2>&1 | grep -v amdgpu.ids | grep "ms/pass\|first-stage" | tr '\n' ' '; echo; done
for a in "10 29903" "1000 30000" "10000 30000 nocpu"; do timeout -k 10 300 python tools/perf_pipeline.py $a 2>&1 | grep "GPU pipeline\|CPU restatement"; done
