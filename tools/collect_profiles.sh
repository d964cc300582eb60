#!/bin/bash
# Collects the round's profile artefacts on the GPU box into gpurun_out/profiles_raw/ (copy the
# summaries into profiles/ afterwards with tools/summarise_profiles.py <round>).  Every rocprofv3 run
# starts the python interpreter itself (no env/bash wrapper), counters in their own passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_raw   # (delete the local copy of this directory before the call: gpurun merges, it does not mirror)
rm -rf $OUT; mkdir -p $OUT
# (--no-small-pool: that leg launches the same first-stage kernel 66 more times on 4e6 and 1.5e6 checks, which would
#  dilute the kernel's average in the summary; the default command's own trace is kept beside it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python3 bench.py --steps 2 --warmup 1 --no-small-pool > $OUT/bench_stdout.log 2>&1 || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_default_stats -o benchdef -- python3 bench.py > $OUT/bench_default_stdout.log 2>&1 || exit 1
echo "default bench done"
# the counters on the BENCH ITSELF: one step of the 65,536-primer pool (4.29e9 checks, nine launches of the first stage)
PMCRUN="python3 bench.py --steps 1 --warmup 0 --no-stage-a --no-stage-b --no-cpu-baseline --no-small-pool"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- $PMCRUN > $OUT/pmc1.log 2>&1 || exit 1
echo "pmc1 done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $PMCRUN > $OUT/pmc2.log 2>&1 || exit 1
echo "pmc2 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMCRUN > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMCRUN > $OUT/pmc_write.log 2>&1 || exit 1
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stage_a_stats -o stagea -- python3 tools/perf_stage_a.py 10000 30000 > $OUT/stage_a_stdout.log 2>&1 || exit 1
echo "stage a done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stage_b_stats -o stageb -- python3 tools/perf_stage_b.py 1048576 > $OUT/stage_b_stdout.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stage_b2k_stats -o stageb2k -- python3 tools/perf_stage_b.py 2000 > $OUT/stage_b2k_stdout.log 2>&1 || exit 1
echo "stage b done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small_pool_stats -o smallpool -- python3 tools/perf_small_pool.py 2000 --reps 20 > $OUT/small_pool_stdout.log 2>&1 || exit 1
echo "small pool done"
timeout -k 10 120 tools/valu_peak > $OUT/valu_peak.jsonl 2>&1; echo "valu_peak done"
echo done
