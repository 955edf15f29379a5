#!/usr/bin/env python3
"""Per-QP SQ counters of the solve kernel with one QP per CU (B = 256) and two (B = 8192) from tools/coresidency_pmc.sh's CSVs:
copies them to profiles/<round>/<tag>_cores_* and prints / writes the table.   python tools/coresidency_summary.py v28 r04"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path, kernel="solve_queue_kernel"):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if kernel not in r["Kernel_Name"]:
            continue
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(by.values())


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "cores_" + tag)
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    out = {}
    for B in (256, 8192):
        acc = {}
        for name in ("sq", "lds", "issue"):
            f = os.path.join(src, "pmc_%s_b%d.csv" % (name, B))
            shutil.copy(f, os.path.join(dst, "%s_cores_pmc_%s_b%d.csv" % (tag, name, B)))
            rows = per_dispatch(f)
            for k in rows[0]:
                v = [r[k] for r in rows if k in r]
                acc.setdefault(k, sum(v) / len(v))
            acc["launches_" + name] = len(rows)
        ks = os.path.join(src, "kernel_stats_b%d.csv" % B)
        shutil.copy(ks, os.path.join(dst, "%s_cores_kernel_stats_b%d.csv" % (tag, B)))
        for r in csv.DictReader(open(ks)):
            if "solve_queue_kernel" in r["Name"]:
                acc["kernel_us"] = float(r["AverageNs"]) / 1e3
        q = float(B)
        cyc = acc["SQ_WAVE_CYCLES"]
        out["b%d" % B] = {
            "qps_per_cu_at_once": 1 if B == 256 else 2, "kernel_us": acc["kernel_us"], "waves": acc["SQ_WAVES"],
            "wave_cycles_per_qp_wave": cyc / q / 4.0,  # SQ_WAVE_CYCLES counts in units of 4 cycles per wave on this part: compare the two columns, not the unit
            "valu_per_qp_wave": acc["SQ_INSTS_VALU"] / q / 4.0, "salu_per_qp_wave": acc["SQ_INSTS_SALU"] / q / 4.0, "lds_insts_per_qp_wave": acc["SQ_INSTS_LDS"] / q / 4.0,
            "share_issuing": acc["SQ_ACTIVE_INST_ANY"] / cyc, "share_waiting_any": acc["SQ_WAIT_ANY"] / cyc, "share_ready_not_picked": acc["SQ_WAIT_INST_ANY"] / cyc,
            "share_issuing_valu": acc["SQ_ACTIVE_INST_VALU"] / cyc, "share_issuing_scalar": acc["SQ_ACTIVE_INST_SCA"] / cyc, "share_issuing_lds": acc["SQ_ACTIVE_INST_LDS"] / cyc,
            "share_waiting_for_lds_issue": acc["SQ_WAIT_INST_LDS"] / cyc,
            "lds_bank_conflict_share_of_active": acc["SQ_LDS_BANK_CONFLICT"] / acc["SQ_LDS_IDX_ACTIVE"], "lds_idx_active_per_qp": acc["SQ_LDS_IDX_ACTIVE"] / q,
        }
    a, b = out["b256"], out["b8192"]
    out["two_over_one"] = {k: (b[k] / a[k] if a[k] else None) for k in a if isinstance(a[k], float) and k not in ("kernel_us", "waves")}
    json.dump(out, open(os.path.join(dst, "%s_cores_summary.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
