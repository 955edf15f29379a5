#!/usr/bin/env python3
"""Copies one tools/profile_round.sh pass from gpurun_out/prof_<tag>/ into profiles/<round>/ (prefix <tag>_) and writes the PMC summary
FROM THOSE VERY CSVs: <tag>_pmc_summary.json and profiles/pmc_latest.json (what bench.py scales `roofline.traffic` from).
    python tools/pmc_summary.py v20 r03"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path, kernel):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if kernel not in r["Kernel_Name"]:
            continue
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(by.values())


def mean(rows, key):
    v = [r[key] for r in rows if key in r]
    return sum(v) / len(v), len(v)


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    keep = ["bench.json", "sweep.json", "kernel_stats.csv", "pmc_fetch_size.csv", "pmc_write_size.csv", "pmc_sq.csv", "phase.txt", "phase_model.txt",
            "terms_phase.txt", "tick_latency.json", "straggler.txt", "straggler_product.json", "bench_franka_b8192.json", "rollout_bench.json"]
    for f in keep:
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f)))
    kern = "solve_queue_kernel"
    fe = per_dispatch(os.path.join(src, "pmc_fetch_size.csv"), kern)
    wr = per_dispatch(os.path.join(src, "pmc_write_size.csv"), kern)
    sq = per_dispatch(os.path.join(src, "pmc_sq.csv"), kern)
    f_kb, nf = mean(fe, "FETCH_SIZE")
    w_kb, nw = mean(wr, "WRITE_SIZE")
    waves, _ = mean(sq, "SQ_WAVES")
    cyc, _ = mean(sq, "SQ_WAVE_CYCLES")
    out = {"round": int(rnd.lstrip("r")), "kernel": "wbcqp::solve_queue_kernel<double, true>", "workload": "talos_pos_tracker_b1024_fp64_squat_tick_stream",
           "batch": 1024, "fetch_size_kb": round(f_kb, 2), "write_size_kb": round(w_kb, 2), "launches_fetch_pass": nf, "launches_write_pass": nw,
           "traffic_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0, "traffic_over_algorithmic": (2.0 * f_kb + w_kb) * 1024.0 / (35152.0 * 1024),
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, then the SQ counters) over `bench.py --steps 20 --warmup 4 --headline-only` "
                   "(the tick stream's launches and nothing else); every number here is computed from profiles/%s/%s_pmc_*.csv by tools/pmc_summary.py. "
                   "gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request, so read bytes = 2 x FETCH_SIZE; "
                   "WRITE_SIZE reads exactly." % (rnd, tag),
           "commands": ["tools/profile_round.sh %s" % tag, "python tools/pmc_summary.py %s %s" % (tag, rnd)],
           "sq_counters": {"launches": len(sq), "waves_per_launch": waves,
                           "active_inst_any": mean(sq, "SQ_ACTIVE_INST_ANY")[0] / cyc, "wait_any": mean(sq, "SQ_WAIT_ANY")[0] / cyc,
                           "wait_inst_any": mean(sq, "SQ_WAIT_INST_ANY")[0] / cyc,
                           "valu_per_wave": mean(sq, "SQ_INSTS_VALU")[0] / waves, "salu_per_wave": mean(sq, "SQ_INSTS_SALU")[0] / waves,
                           "lds_per_wave": mean(sq, "SQ_INSTS_LDS")[0] / waves,
                           "note": "per wave of a launch; a wave slot of the 512 resident workgroups solves 2 QPs of a 1024-QP launch: halve for per-QP figures"}}
    with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
