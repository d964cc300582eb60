#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/s2_stage_b.log
for n in 2000 32768 131072 1048576; do timeout -k 10 200 python tools/perf_stage_b.py $n >> $O/s2_stage_b.log 2>&1 || exit 1; done; grep -v amdgpu.ids $O/s2_stage_b.log
MSSPE_PROBE_OPTIONS=self_lane_from=0 timeout -k 10 100 python tools/perf_stage_b.py 8192 > $O/s2_stage_b_8k_lane.log 2>&1; MSSPE_PROBE_OPTIONS=self_lane_from=1000000000 timeout -k 10 100 python tools/perf_stage_b.py 8192 > $O/s2_stage_b_8k_wave.log 2>&1; MSSPE_PROBE_OPTIONS=self_lane_from=1000000000 timeout -k 10 100 python tools/perf_stage_b.py 65536 > $O/s2_stage_b_64k_wave.log 2>&1; MSSPE_PROBE_OPTIONS=self_lane_from=0 timeout -k 10 100 python tools/perf_stage_b.py 65536 > $O/s2_stage_b_64k_lane.log 2>&1
grep self_dimers $O/s2_stage_b_*k_*.log
rm -rf /tmp/prof_sb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sb -o sb -- python3 tools/perf_stage_b.py 1048576 > $O/s2_sb_prof.log 2>&1; echo "prof rc $?"
find /tmp/prof_sb -name '*kernel_stats.csv' -exec cp {} $O/s2_stage_b_1m_kernel_stats.csv \;
head -12 $O/s2_stage_b_1m_kernel_stats.csv | cut -c1-200
