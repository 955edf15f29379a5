#!/bin/bash
# wbcqp_rollout against the loop of wbcqp_tick over the grid of profiles/r03/rollout_bench.log and three more shapes: the library
# chooses the number of sub-batches from its own measurements (include/wbcqp.h); WBCQP_ROLLOUT_DEBUG shows the choice per call.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export WBCQP_ROLLOUT_DEBUG=1
run() { echo "# tools/rollout_bench.py $*"; python3 tools/rollout_bench.py "$@" 2> /tmp/rb.err; grep "wbcqp_rollout" /tmp/rb.err | tail -2; }
run
run --noise 0.05 --phase 7
run --batch 4096 --ticks 32
run --batch 2048 --ticks 32
run --batch 512
run --batch 8192 --ticks 16
unset WBCQP_ROLLOUT_DEBUG
for s in 1 2 3 4; do echo "# WBCQP_ROLLOUT_STREAMS=$s tools/rollout_bench.py"; WBCQP_ROLLOUT_STREAMS=$s python3 tools/rollout_bench.py 2> /dev/null; done
