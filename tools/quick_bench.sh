#!/bin/bash
# Development aid (GPU box): the headline bench without the side legs; prints value, ms/step, first-stage launch ms, frac
timeout -k 10 200 python bench.py --steps ${1:-3} --warmup 1 --no-stage-a --no-stage-b --no-cpu-baseline > /tmp/qb.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("/tmp/qb.json"))
r = d["roofline"]
print(f"checks/s {d['value']:.4g}  ms/step {d['ms_per_step']:.1f}  launch {r['avg_launch_ms']:.2f} ms  frac {r['frac']:.4f}  handed on {r['retried_in_list_mode']:.4f}")
PY
