cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rows_pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tools/terms_profile.py --time $GRAFT_REPO_ROOT/inria_wbc_amd/lib/libwbcqp_$1.so > $OUT/log.txt 2>&1
find $OUT/sq -name "*counter_collection.csv" -exec cp {} $OUT/pmc.csv \;
rm -rf $OUT/sq
python3 - "$OUT/pmc.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if "terms" not in r["Kernel_Name"]: continue
    d = by.setdefault(r["Dispatch_Id"], {})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
d = list(by.values())[-1]
w = d.get("SQ_WAVES", 4)
print({c[3:]: round(v / w) for c, v in d.items() if c != "SQ_WAVES"}, "waves", int(w))
PY
