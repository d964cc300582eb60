#!/bin/bash
# the strong-scaling configuration as the driver would run it at N = 1 (1,048,576 primers, budgeted run)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
start=$(date +%s)
timeout -k 10 700 python bench.py --config pool1m > $O/pool1m.json 2> $O/pool1m.err; rc=$?
echo "rc $rc wall $(( $(date +%s) - start )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/pool1m.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "timed_region_s", "scaling")})
print(d["config"])
PY
