#!/bin/bash
# What tools/profile_round.sh does not cover: the counters that name the solve kernel's limiter (LDS conflicts and array cycles, f64 instruction
# mix) and a rocprofv3 kernel trace for every OTHER kernel a number is quoted for (one wavefront per QP on Franka B = 8192, the dense seam's
# kernel at batch 1 and 256, the rows kernel, the roll-out).  Usage: tools/profile_kernels.sh <tag>; tools/pmc_summary.py folds it in.
set -u
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="python3 $ROOT/bench.py --steps 20 --warmup 4 --headline-only"
pmc() { # name, counters...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $HEAD > $OUT/$name.log 2>&1
    find $OUT/$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
    rm -rf $OUT/$name
}
pmc lds SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pmc ldsbw SQ_WAVES SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_LDS_UNALIGNED_STALL SQ_BUSY_CU_CYCLES
pmc f64 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
pmc issue SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU_CVT
trace() { # name, command...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/trace_$name.log 2>&1
    find $OUT/$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
    rm -rf $OUT/$name
}
trace franka_b8192 python3 $ROOT/bench.py --robot franka --batch 8192 --steps 200 --warmup 20 --headline-only
trace dense_b1 python3 $ROOT/tools/dense_time.py --batch 1
trace dense_b256 python3 $ROOT/tools/dense_time.py --batch 256
trace rows python3 $ROOT/tools/terms_profile.py --time $ROOT/inria_wbc_amd/lib/libwbcqp.so
trace rollout python3 $ROOT/tools/rollout_bench.py
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/rows_sq -- python3 $ROOT/tools/terms_profile.py --time $ROOT/inria_wbc_amd/lib/libwbcqp.so > $OUT/rows_sq.log 2>&1
find $OUT/rows_sq -name "*counter_collection.csv" -exec cp {} $OUT/pmc_rows_sq.csv \;
rm -rf $OUT/rows_sq
ls -la $OUT | head -40
for f in $OUT/kernel_stats_*.csv; do echo $f; head -4 $f | cut -c1-200; done
grep -h "us per launch" $OUT/trace_dense_b1.log $OUT/trace_dense_b256.log $OUT/trace_rows.log
