#!/bin/bash
# A/B timing of library variants (_var/variants/<name>/libmsspe_hip.so): 65,536-primer screen, first-stage launch time
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
for v in "$@"; do
  MSSPE_PROBE_PROFILE=1 MSSPE_PROBE_LIB=_var/variants/$v/libmsspe_hip.so timeout -k 10 120 python tools/perf_probe.py 65536 2>&1 | grep -v amdgpu.ids | grep "first-stage\|ms/pass" | tr '\n' ' ' | sed "s/^/$v: /" | tee -a $O/ab.log; echo | tee -a $O/ab.log
done
