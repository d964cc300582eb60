#!/bin/bash
# quick headline check: parity subset + bench without the extra legs
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
TAG=${1:-q}
timeout -k 10 600 python -m pytest tests/test_gpu_thermo_parity.py -x -q -m gpu -k "not long_oligo and not hairpin_wave" > $O/${TAG}_tests.log 2>&1; rc=$?; tail -3 $O/${TAG}_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-stage-a --no-stage-b --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; rc=$?; python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value %.4g ms/step %.1f frac %.4f launch_ms %.2f launches %d retried %.4f" % (d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["launches"], r["retried_in_list_mode"]))
PY
exit $rc
