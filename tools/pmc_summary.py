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
            "terms_phase.txt", "tick_latency.json", "straggler.txt", "straggler_product.json", "bench_franka_b8192.json", "rollout_bench.json",
            # tools/profile_kernels.sh: the limiter's counters and a kernel trace for every other kernel a number is quoted for
            "pmc_lds.csv", "pmc_ldsbw.csv", "pmc_f64.csv", "pmc_issue.csv", "pmc_rows_sq.csv", "kernel_stats_franka_b8192.csv", "kernel_stats_dense_b1.csv",
            "kernel_stats_dense_b256.csv", "kernel_stats_rows.csv", "kernel_stats_rollout.csv", "trace_dense_b1.log", "trace_dense_b256.log", "trace_rows.log",
            "rollout_grid.log", "throughput.json", "determinism_probe.txt"]
    for f in keep:
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f)))
    # the dominant kernel is whatever the pass's bench line names (round 6: B = 1024 goes to the dispatcher, solve_kernel; before: the queue kernel)
    kernel_full = "wbcqp::solve_queue_kernel<double, true, 1>"
    try:
        with open(os.path.join(src, "bench.json")) as fh:
            kernel_full = json.loads(fh.read().strip().splitlines()[-1])["roofline"]["kernel"]
    except Exception:  # noqa: BLE001
        pass
    kern = kernel_full.split("::", 1)[1]  # (as rocprofv3 prints it after "wbcqp::"; "solve_kernel<" does not match "solve_queue_kernel<")
    fe = per_dispatch(os.path.join(src, "pmc_fetch_size.csv"), kern)
    wr = per_dispatch(os.path.join(src, "pmc_write_size.csv"), kern)
    sq = per_dispatch(os.path.join(src, "pmc_sq.csv"), kern)
    f_kb, nf = mean(fe, "FETCH_SIZE")
    w_kb, nw = mean(wr, "WRITE_SIZE")
    waves, _ = mean(sq, "SQ_WAVES")
    cyc, _ = mean(sq, "SQ_WAVE_CYCLES")
    out = {"round": int(rnd.lstrip("r")), "kernel": kernel_full, "workload": "talos_pos_tracker_b1024_fp64_squat_tick_stream",
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
                           "note": "per wave of a launch: 4096 waves = one workgroup (four waves) per QP; 2048 = the queue's 512 resident workgroups, each wave slot solving 2 QPs of "
                                   "a 1024-QP launch (halve for per-QP figures)"}}
    # ---- what names the limiter (tools/profile_kernels.sh passes; absent files leave the keys out) ----
    kern_ms = None
    try:
        for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
            if kern in r["Name"]:
                kern_ms = float(r["AverageNs"]) * 1e-6
    except OSError:
        pass
    out["kernel_ms_traced"] = kern_ms

    def have(name):
        return os.path.exists(os.path.join(src, name))

    if have("pmc_f64.csv") and kern_ms:
        f = per_dispatch(os.path.join(src, "pmc_f64.csv"), kern)
        fma, add, mul, tr = (mean(f, "SQ_INSTS_VALU_%s_F64" % k)[0] for k in ("FMA", "ADD", "MUL", "TRANS"))
        valu = mean(f, "SQ_INSTS_VALU")[0]
        flops = 64.0 * (2.0 * fma + add + mul + tr)
        out["fp64"] = {"bound": "vector f64", "achieved": flops / (kern_ms * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                       "frac": flops / (kern_ms * 1e-3) / 1e12 / 78.6, "flops_per_launch_issued": flops,
                       "insts_per_launch": {"VALU": valu, "FMA_F64": fma, "ADD_F64": add, "MUL_F64": mul, "TRANS_F64": tr,
                                            "INT32": mean(f, "SQ_INSTS_VALU_INT32")[0], "INT64": mean(f, "SQ_INSTS_VALU_INT64")[0]},
                       "f64_share_of_valu": (fma + add + mul + tr) / valu,
                       "note": "counted instructions x 64 lanes (2 flops per FMA): what the SIMDs ISSUED, masked lanes and redundant per-thread copies "
                               "of wave-uniform arithmetic included -- an upper bound on useful flops (bench.py's analytic figure: Structure.flops_estimate)"}
    if have("pmc_lds.csv") and kern_ms:
        l = per_dispatch(os.path.join(src, "pmc_lds.csv"), kern)
        idx, conf = mean(l, "SQ_LDS_IDX_ACTIVE")[0], mean(l, "SQ_LDS_BANK_CONFLICT")[0]
        lcyc = mean(l, "SQ_WAVE_CYCLES")[0]
        out["lds"] = {"idx_active": idx, "bank_conflict": conf, "bank_conflict_share_of_active": conf / idx, "addr_conflict": mean(l, "SQ_LDS_ADDR_CONFLICT")[0],
                      "insts_per_wave": mean(l, "SQ_INSTS_LDS")[0] / mean(l, "SQ_WAVES")[0],
                      "wave_cycles_issuing_lds": mean(l, "SQ_ACTIVE_INST_LDS")[0] / lcyc, "wave_cycles_stalled_on_lds_issue": mean(l, "SQ_WAIT_INST_LDS")[0] / lcyc}
        if have("pmc_ldsbw.csv"):
            b = per_dispatch(os.path.join(src, "pmc_ldsbw.csv"), kern)
            ld, stv = mean(b, "SQ_INSTS_LDS_LOAD_BANDWIDTH")[0], mean(b, "SQ_INSTS_LDS_STORE_BANDWIDTH")[0]
            nbytes = 64.0 * (ld + stv)
            out["lds"].update({"bound": "lds", "bytes_per_launch": nbytes, "achieved": nbytes / (kern_ms * 1e-3) / 1e12, "peak": 150.0, "unit": "TB/s",
                               "frac": nbytes / (kern_ms * 1e-3) / 1e12 / 150.0,
                               "load_insts": mean(b, "SQ_INSTS_LDS_LOAD")[0], "store_insts": mean(b, "SQ_INSTS_LDS_STORE")[0],
                               "unaligned_stall": mean(b, "SQ_LDS_UNALIGNED_STALL")[0],
                               "note": "bytes = (SQ_INSTS_LDS_LOAD_BANDWIDTH + SQ_INSTS_LDS_STORE_BANDWIDTH) x 64 B (the counters step once per 64 bytes "
                                       "moved: 12.7 per load instruction = the mix of 8-byte and 16-byte wave64 reads this kernel issues); peak = the "
                                       "guide's aggregate for ds_read_b64 / b128 with every CU streaming (MI355X_MICROARCH.md, LDS section)"})
    if have("pmc_issue.csv"):
        q = per_dispatch(os.path.join(src, "pmc_issue.csv"), kern)
        qc = mean(q, "SQ_WAVE_CYCLES")[0]
        out["issue"] = {"valu": mean(q, "SQ_ACTIVE_INST_VALU")[0] / qc, "scalar": mean(q, "SQ_ACTIVE_INST_SCA")[0] / qc, "lds": mean(q, "SQ_ACTIVE_INST_LDS")[0] / qc,
                        "vmem": mean(q, "SQ_ACTIVE_INST_VMEM")[0] / qc, "note": "share of a wave's cycles in which it has an instruction of that kind in flight"}
    with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
