#!/bin/bash
# PMC pass over batch-1 launches of the straggler and the median QP
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tools/straggler_time.py --reps 4 > $OUT/log.txt 2>&1
find $OUT/sq -name "*counter_collection.csv" -exec cp {} $OUT/pmc.csv \;
rm -rf $OUT/sq
python3 - <<'PY'
import csv, os, collections
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_one/pmc.csv")
rows = list(csv.DictReader(open(p)))
# group by dispatch id
by = collections.OrderedDict()
for r in rows:
    if "solve" not in r["Kernel_Name"]: continue
    d = by.setdefault(r["Dispatch_Id"], {"grid": r["Grid_Size"], "k": r["Kernel_Name"][:40]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
seen = collections.OrderedDict()
for k, d in by.items():
    if d["grid"] != "256": continue
    key = (round(d.get("SQ_INSTS_VALU", 0)),)
    seen.setdefault(key, d)
for key, d in list(seen.items())[:12]:
    w = d.get("SQ_WAVES", 4)
    print({c: round(v / w) for c, v in d.items() if c.startswith("SQ_")})
PY
