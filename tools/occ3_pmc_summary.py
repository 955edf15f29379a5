#!/usr/bin/env python3
"""Table of tools/occ3_pmc.sh: iCub at two and at three workgroups per CU, per launch of 8192 QPs -- kernel time, what a wave does with its cycles, instruction
counts.  Copies the CSVs to profiles/<round>/<tag>_occ3_* and writes <tag>_occ3_summary.json.   python tools/occ3_pmc_summary.py v33 r05"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "solve_" not in r["Kernel_Name"] or "small" in r["Kernel_Name"]:
            continue
        by.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(by.values())


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "occ3_" + tag)
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    out = {"stack": "icub (n 62, nEq 18)", "batch": 8192, "commands": ["tools/occ3_pmc.sh " + tag, "python tools/occ3_pmc_summary.py %s %s" % (tag, rnd)]}
    for mode in ("two", "three"):
        acc = {}
        for name in ("sq", "issue", "lds"):
            f = os.path.join(src, "pmc_%s_%s.csv" % (name, mode))
            shutil.copy(f, os.path.join(dst, "%s_occ3_pmc_%s_%s.csv" % (tag, name, mode)))
            rows = per_dispatch(f)
            acc["kernel"] = rows[0]["kernel"].split("(")[0]
            for k in rows[0]:
                if k == "kernel":
                    continue
                v = [r[k] for r in rows if k in r]
                acc.setdefault(k, sum(v) / len(v))
            acc["launches_" + name] = len(rows)
        ks = os.path.join(src, "kernel_stats_%s.csv" % mode)
        shutil.copy(ks, os.path.join(dst, "%s_occ3_kernel_stats_%s.csv" % (tag, mode)))
        for r in csv.DictReader(open(ks)):
            if "solve_" in r["Name"] and "small" not in r["Name"]:
                acc["kernel_us"] = float(r["AverageNs"]) / 1e3
                acc["kernel_calls"] = int(r["Calls"])
        wc = acc["SQ_WAVE_CYCLES"]
        out[mode] = {
            "kernel": acc["kernel"], "kernel_us_per_launch": acc.get("kernel_us"), "qps": 8192 / (acc["kernel_us"] * 1e-6) if acc.get("kernel_us") else None,
            "waves_per_launch": acc["SQ_WAVES"], "wave_cycles_per_qp": wc / 8192,
            "issuing_any": acc["SQ_ACTIVE_INST_ANY"] / wc, "waiting_any": acc["SQ_WAIT_ANY"] / wc, "waiting_on_issue": acc["SQ_WAIT_INST_ANY"] / wc,
            "issue_valu": acc["SQ_ACTIVE_INST_VALU"] / wc, "issue_scalar": acc["SQ_ACTIVE_INST_SCA"] / wc, "issue_lds": acc["SQ_ACTIVE_INST_LDS"] / wc,
            "issue_vmem": acc["SQ_ACTIVE_INST_VMEM"] / wc, "lds_wait": acc["SQ_WAIT_INST_LDS"] / wc,
            "valu_per_qp": acc["SQ_INSTS_VALU"] / 8192, "salu_per_qp": acc["SQ_INSTS_SALU"] / 8192, "lds_per_qp": acc["SQ_INSTS_LDS"] / 8192,
            "vmem_rd_per_qp": acc["SQ_INSTS_VMEM_RD"] / 8192, "vmem_wr_per_qp": acc["SQ_INSTS_VMEM_WR"] / 8192,
            "lds_bank_conflict_share": acc["SQ_LDS_BANK_CONFLICT"] / max(acc["SQ_LDS_IDX_ACTIVE"], 1.0),
        }
    a, b = out["two"], out["three"]
    if a["kernel_us_per_launch"] and b["kernel_us_per_launch"]:
        out["three_over_two"] = a["kernel_us_per_launch"] / b["kernel_us_per_launch"]
    path = os.path.join(dst, "%s_occ3_summary.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    for mode in ("two", "three"):
        print(mode, json.dumps(out[mode]))
    print("three over two:", out.get("three_over_two"))


if __name__ == "__main__":
    main()
