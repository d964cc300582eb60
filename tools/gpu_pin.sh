#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_thermo_parity.py -x -q -m gpu -k "not long_oligo and not hairpin_wave" > $O/pin_tests.log 2>&1; rc=$?; tail -3 $O/pin_tests.log; [ $rc -eq 0 ] || exit 1
bash tools/gpu_ab.sh "$@"
