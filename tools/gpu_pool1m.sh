#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python bench.py --config pool1m --pool 131072 --steps 3 --warmup 1 --max-seconds 60 > $O/pool1m_small.json 2> $O/pool1m_small.err; echo "rc $?"; python - <<PY
import json
d=json.loads(open("$O/pool1m_small.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","steps","warmup","steps_requested","warmup_requested","ms_per_step","timed_region_s","scaling","note")})
PY
