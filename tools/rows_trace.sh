#!/bin/bash
# rocprofv3 kernel trace of the rows kernel alone (tools/terms_profile.py --time <lib>): its average duration without the launch gaps
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rows_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $GRAFT_REPO_ROOT/tools/terms_profile.py --time $GRAFT_REPO_ROOT/inria_wbc_amd/lib/$1 > $OUT/log.txt 2>&1
find $OUT/t -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/t
tail -1 $OUT/log.txt
head -4 $OUT/kernel_stats.csv | cut -c1-220
