#!/bin/bash
# PMC passes on the 65,536-primer screen (one pass of perf_probe = 3 screens) for the library given (default: the product's)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/pmc_${1:-main}; rm -rf $O; mkdir -p $O
[ -n "$1" ] && export MSSPE_PROBE_LIB=_var/variants/$1/libmsspe_hip.so
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 tools/perf_probe.py 65536 > $O/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python3 tools/perf_probe.py 65536 > $O/pmc2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc3 -- python3 tools/perf_probe.py 65536 > $O/pmc3.log 2>&1 || true
for p in pmc1 pmc2 pmc3; do python3 tools/pmc_summary.py "$O/$p/**/*counter_collection.csv" k_pairs_row > $O/$p.txt 2>&1; cat $O/$p.txt; done
