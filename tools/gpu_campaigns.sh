#!/bin/bash
# round 4's one-off randomised campaigns on the final build (each prints BAD n and exits non-zero on a difference)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
run() { name=$1; shift; timeout -k 10 $1 python "${@:2}" > $O/camp_$name.log 2>&1; echo "$name rc $? : $(tail -1 $O/camp_$name.log)"; }
run pairs_${PAIRS_SEED:-401} 420 tools/random_campaign.py ${PAIRS_SEED:-401} 120
run stage_b_7 150 tools/random_campaign_stage_b.py 7 100
run stage_a_big_9 240 tools/random_campaign_stage_a_big.py 9 60
run stage_a_23 150 tools/random_campaign_stage_a.py 23 150
run cli_31 200 tools/random_campaign_cli.py 31 40
run blocks_12 100 tools/random_campaign_blocks.py 12 60
