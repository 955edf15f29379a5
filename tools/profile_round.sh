#!/bin/bash
# One measurement pass on the GPU box: bench line (with CPU baseline), rocprofv3 kernel-trace stats, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE separately), in-kernel phase shares of both kernels, whole-tick latency table.  Usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $ROOT
python3 -c "from inria_wbc_amd import build; build.build(); build.build_stamps()" > $OUT/build.log 2>&1
python3 bench.py --steps 20 --warmup 5 --sweep $OUT/sweep.json > $OUT/bench.json 2> $OUT/bench.err
python3 tools/phase_profile.py --out $OUT/phase.json > $OUT/phase.txt 2>&1
python3 tools/phase_profile.py --model > $OUT/phase_model.txt 2>&1
python3 tools/terms_profile.py > $OUT/terms_phase.txt 2>&1
python3 tools/tick_latency.py --reps 100 --out $OUT/tick_latency.json > $OUT/tick_latency.txt 2>&1
python3 tools/straggler_time.py --stamps > $OUT/straggler.txt 2>&1
python3 tools/straggler_time.py --out $OUT/straggler_product.json > /dev/null 2>&1
python3 bench.py --robot franka --batch 8192 --steps 50 --warmup 10 --no-sweep > $OUT/bench_franka_b8192.json 2> /dev/null
python3 tools/rollout_bench.py > $OUT/rollout_bench.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 200 --warmup 20 --headline-only > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 20 --warmup 4 --headline-only > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 20 --warmup 4 --headline-only > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --steps 20 --warmup 4 --headline-only > $OUT/sq.log 2>&1
cd $OUT
find ./sq -name "*counter_collection.csv" -exec cp {} $OUT/pmc_sq.csv \;
find . -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find ./fetch -name "*counter_collection.csv" -exec cp {} $OUT/pmc_fetch_size.csv \;
find ./write -name "*counter_collection.csv" -exec cp {} $OUT/pmc_write_size.csv \;
rm -rf trace fetch write sq
ls -la $OUT
tail -1 $OUT/bench.json | cut -c1-300
head -3 $OUT/kernel_stats.csv
