#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_stage_a_parity.py -x -q -m gpu -k "config2 or config1 or stage_a or candidate or posting" > $O/sa_tests.log 2>&1; rc=$?; tail -3 $O/sa_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-stage-b --no-cpu-baseline --no-small-pool > $O/sa_bench.json 2>$O/sa_bench.err; python - <<PY
import json
d=json.loads(open("$O/sa_bench.json").read().strip().splitlines()[-1])["stage_a"]
print({k:d[k] for k in ("ms_per_direction","both_directions_one_call_ms","both_directions_one_call_equal")}, d["roofline"]["event_ms_per_direction"])
PY
