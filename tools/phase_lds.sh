#!/bin/bash
# LDS work per phase of one Talos QP: libraries built with -DWBCQP_X_STOP=<stamp> (tools/variants.sh) end the QP at that stamp; the LDS
# counters of consecutive ones differ by one phase (instructions, cycles the LDS index unit is active, bank-conflict cycles).
# usage (GPU box): tools/phase_lds.sh <qp index> tag1 tag2 ...
cd /tmp && export TMPDIR=/tmp
QP=$1; shift
for t in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/phase_lds_$t
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tools/one_qp.py --lib $GRAFT_REPO_ROOT/inria_wbc_amd/lib/libwbcqp_$t.so --qp $QP > $OUT/log.txt 2>&1
  find $OUT/sq -name "*counter_collection.csv" -exec cp {} $OUT/pmc.csv \;
  rm -rf $OUT/sq
  python3 - "$t" "$OUT/pmc.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[2])))
by = collections.OrderedDict()
for r in rows:
    if "solve" not in r["Kernel_Name"]: continue
    d = by.setdefault(r["Dispatch_Id"], {})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
d = list(by.values())[-1]
print(sys.argv[1], {c[3:]: round(v) for c, v in d.items()})
PY
done
