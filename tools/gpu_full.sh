#!/bin/bash
# the whole -m gpu suite, then the bench as the driver runs it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; rc=$?; tail -4 $O/full_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/full_bench.json 2> $O/full_bench.err; rc=$?; tail -c 1500 $O/full_bench.json; exit $rc
