#!/bin/bash
# PMC passes for the dominant kernel on a 16,384-primer probe (counters in their own runs, no tracing
# domains besides the kernel trace).  usage: tools/collect_pmc.sh <outdir-under-gpurun_out>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1
mkdir -p $OUT
N=${2:-16384}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- python3 tools/perf_probe.py $N > $OUT/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 tools/perf_probe.py $N > $OUT/pmc2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc3 -- python3 tools/perf_probe.py $N > $OUT/pmc3.log 2>&1 || true
for p in pmc1 pmc2 pmc3; do python3 tools/pmc_summary.py "$OUT/$p/**/*counter_collection.csv" k_pairs_ > $OUT/$p.txt 2>&1; done
cat $OUT/pmc1.txt $OUT/pmc2.txt $OUT/pmc3.txt
