#!/bin/bash
# One QP per CU against two: the SQ counters of the solve kernel on the tick stream at B = 256 (every QP alone on its CU) and at B = 8192
# (512 persistent workgroups, two per CU, sixteen QPs each), three PMC passes each.  What does a QP pay for its neighbour, and in which
# counter does it show?  Usage: tools/coresidency_pmc.sh <tag>; tools/coresidency_summary.py <tag> prints the per-QP table.
set -u
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/cores_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 256 8192; do
  HEAD="python3 $ROOT/bench.py --batch $B --steps 20 --warmup 4 --headline-only"
  pmc() { # name, counters...
      local name=$1; shift
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${name}_b$B -- $HEAD > $OUT/${name}_b$B.log 2>&1
      find $OUT/${name}_b$B -name "*counter_collection.csv" -exec cp {} $OUT/pmc_${name}_b$B.csv \;
      rm -rf $OUT/${name}_b$B
  }
  pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
  pmc lds SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
  pmc issue SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_b$B -- $HEAD > $OUT/trace_b$B.log 2>&1
  find $OUT/trace_b$B -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_b$B.csv \;
  rm -rf $OUT/trace_b$B
  tail -1 $OUT/trace_b$B.log | cut -c1-200
done
ls $OUT
