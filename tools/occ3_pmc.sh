#!/bin/bash
# Two workgroups per CU against three, on iCub (BASELINE config 3's stack, B = 8192, f64): the SQ counters and the kernel time of the product's two-per-CU
# kernel (hardware dispatch of solve_kernel<double, true, 2>: 206 VGPRs, no scratch) and of its three-per-CU kernel (solve_queue3_kernel<double, 2>: 168 VGPRs,
# 204 B/lane of scratch).  What does the third QP buy per wave cycle, what do the spills cost?  Usage: tools/occ3_pmc.sh <tag>; tools/occ3_pmc_summary.py <tag> <round>.
set -u
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/occ3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in three two; do
  FLAG=""; [ $MODE = two ] && FLAG="--two"
  HEAD="python3 $ROOT/tools/occ3_probe.py --stack icub --batch 8192 --reps 12 $FLAG"
  pmc() { # name, counters...
      local name=$1; shift
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${name}_$MODE -- $HEAD > $OUT/${name}_$MODE.log 2>&1
      find $OUT/${name}_$MODE -name "*counter_collection.csv" -exec cp {} $OUT/pmc_${name}_$MODE.csv \;
      rm -rf $OUT/${name}_$MODE
  }
  pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
  pmc issue SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pmc lds SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$MODE -- $HEAD > $OUT/trace_$MODE.log 2>&1
  find $OUT/trace_$MODE -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$MODE.csv \;
  rm -rf $OUT/trace_$MODE
  grep "^{" $OUT/trace_$MODE.log | tail -1 | cut -c1-300
done
ls $OUT
