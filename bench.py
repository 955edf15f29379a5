#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the batched whole-body-QP tick on MI355X.

One "step" = one pass of the hot path (assemble H,g -> GI active-set solve -> torque decode, i.e.
controller.cpp:244-251 of the reference for every instance) over one batch of synthetic Talos QPs
that is already resident in HBM.  Default workload = BASELINE.json configs[1]:
Talos pos-tracker, batch 1024, fp64, one wavefront per QP, 1 x MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the batch shards embarrassingly -- every rank owns `--batch` QPs of the same seeded stream
(weak scaling), no collective on the solve path (the QPs of a batch are independent).  `--allgather`
adds the optional exchange step of BASELINE config 4 (all-gather of joint torques over RCCL/xGMI)
after the solve inside the timed step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# compact-boundary bytes per Talos fp64 QP (inputs 34 200 B + outputs 952 B)
ALGORITHMIC_BYTES = {"talos": 35152, "icub": 23872, "franka": 1152}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--batch", type=int, default=1024, help="QPs per GPU per step")
    p.add_argument("--robot", default="talos", choices=["talos", "icub", "franka"])
    p.add_argument("--squat", action="store_true", help="CoM reference follows etc/talos/squat.yaml (BASELINE config 4)")
    p.add_argument("--allgather", action="store_true",
                   help="N > 1: also all-gather the joint torques over RCCL inside every step (BASELINE config 4's optional\n"
                        "exchange; the path itself has none -- the QPs of a batch are independent)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-compare", action="store_true", help="skip the extra index-order and hardware-dispatch runs reported beside `value`")
    p.add_argument("--index-order", action="store_true",
                   help="launch the QPs in index order instead of longest-first (WBCQP_FLAG_INDEX_ORDER)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU-baseline sample")
    p.add_argument("--traffic", type=float, default=None,
                   help="HBM bytes per launch from rocprofv3 PMC passes (default: scaled from profiles/pmc_latest.json)")
    p.add_argument("--sweep", default=None, help="also time batch 1..8192 and write the table to this JSON file")
    return p.parse_args()



def before_path(h, st, dev, B, torch, cpu_baseline=True):
    """Before the path, outside `value`: the rows kernel (q, v, references -> QP record) on a
    Talos-like tree, alone and chained with the solve and the state integration (one whole control tick on the device)."""
    from inria_wbc_amd import capi
    from inria_wbc_amd import model as mdl
    m = mdl.talos_like()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    h.set_structure(1, st)
    h.set_model(1, m, tm)
    s = mdl.sample_states(m, tm, B, 9_000_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
    L = st.field_lengths()
    state = {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
    rows["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
    rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
               status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    qn, vn = torch.zeros_like(state["q"]), torch.zeros_like(state["v"])
    sp = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        h.problem_data(1, B, state, rows, stream=sp)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        h.problem_data(1, B, state, rows, stream=sp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nbytes = B * tm.algorithmic_bytes(m, st)
    res = {"kernel": "wbcqp::terms_kernel<double>", "us_per_launch": us, "bytes_per_launch": nbytes,
           "achieved_GBps": nbytes / (us * 1e-6) / 1e9, "frac_of_hbm_peak": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "instances_per_s": B / (us * 1e-6), "model": "talos_like (45 bodies, nv 50), etc/talos/tasks.yaml stack"}

    def tick():
        h.problem_data(1, B, state, rows, stream=sp)
        h.solve_batch(1, B, rows, out, stream=sp)
        h.integrate(B, st.nv, True, tm.dt, state["q"], state["v"], out["x"], st.n, out["status"], qn, vn, None, stream=sp)

    for _ in range(4):
        tick()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        tick()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    it = out["iters"].cpu().numpy()
    res["whole_tick"] = {"ticks_per_s": B / (ms * 1e-3), "ms_per_batch": ms, "iters_mean": float(it.mean()),
                         "status_optimal": int((out["status"] == 0).sum().item()),
                         "note": "rows kernel + solve kernel + integrate kernel on the same stream, states around the reference posture"}
    if cpu_baseline:
        from oracle import rbd
        ns = min(B, 128)
        t1 = time.perf_counter()
        ora = rbd.task_rows(m, tm, st, s["q"][:ns], s["v"][:ns], s["ref"][:ns], n_threads=1)
        dt1 = time.perf_counter() - t1
        got = {k: rows[k][:ns].cpu().numpy() for k in capi.ROW_FIELDS}
        res["cpu_port"] = {"instances_per_s": ns / dt1, "cores": 1, "sample": "%d instances, oracle/rbd_oracle.c" % ns}
        res["parity"] = {k: float(np.abs(got[k] - ora[k]).max() / max(1.0, np.abs(ora[k]).max())) for k in capi.ROW_FIELDS}
    return res

def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from inria_wbc_amd import build, capi, structure, synth
    if rank == 0:
        build.build()
    if distributed:
        dist.barrier()

    st = structure.STRUCTURES[args.robot]()
    B = args.batch
    seed_key = "talos_squat" if (args.squat and args.robot == "talos") else args.robot
    # rank r owns QPs [r*B, (r+1)*B) of the stream
    inputs = synth.generate(st, B, synth.SEED_BASE[seed_key], first=rank * B, squat=args.squat)
    dev = torch.device("cuda", local_rank)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev),
                 tau=torch.zeros(B, max(st.na, 1), dtype=torch.float64, device=dev),
                 status=torch.full((B,), -99, dtype=torch.int32, device=dev),
                 iters=torch.zeros(B, dtype=torch.int32, device=dev))
    tau_all = torch.zeros(world * B, max(st.na, 1), dtype=torch.float64, device=dev) if distributed else None

    h = capi.Handle(device=local_rank, dtype=capi.F64, flags=capi.FLAG_INDEX_ORDER if args.index_order else 0)
    h.set_structure(0, st)
    layout = capi.layout_of(st)
    do_gather = distributed and args.allgather
    gather_state = {"ok": do_gather, "err": None}

    def step():
        h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
        if gather_state["ok"]:
            try:
                dist.all_gather_into_tensor(tau_all, d_out["tau"])
            except Exception as e:  # keep the scaling run alive; the exchange step is optional
                gather_state["ok"] = False
                gather_state["err"] = repr(e)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # one HIP event pair over the timed region, on the stream the kernels are launched on: per-launch duration = region / K
    # (it includes the two order kernels of every 4th step).  Not one pair per step: measured with rocprofv3, every event
    # record left a 10 us bubble in front of the next solve kernel -- 3.5 % of a step spent on the measurement itself.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
        if gather_state["ok"]:
            dist.all_gather_into_tensor(tau_all, d_out["tau"])
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_avg_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps

    status = d_out["status"].cpu().numpy()
    iters = d_out["iters"].cpu().numpy()
    x_gpu = d_out["x"].cpu().numpy()
    tau_gpu = d_out["tau"].cpu().numpy()[:, :st.na]

    traffic = args.traffic
    if traffic is None and args.robot == "talos":
        try:  # committed PMC measurement (separate --pmc passes), scaled to this launch's batch
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as fh:
                pmc = json.load(fh)
            traffic = (2.0 * pmc["fetch_size_kb"] + pmc["write_size_kb"]) * 1024.0 * B / pmc["batch"]
        except Exception:
            traffic = None

    result = None
    if rank == 0:
        total_qps = world * B * args.steps
        value = total_qps / elapsed
        abytes = ALGORITHMIC_BYTES.get(args.robot, st.algorithmic_bytes())
        achieved = abytes * B / kern_avg_s / 1e9
        result = {
            "metric": "QP solves/sec (Talos ~50-var WBC tick)" if args.robot == "talos" else "QP solves/sec (%s)" % args.robot,
            "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s_pos_tracker%s_b%d_fp64_one_workgroup_per_qp" % (args.robot, "_squat" if args.squat else "", B),
                       "batch_per_gpu": B, "n": st.n, "neq": st.neq, "nin": st.nin, "level1_rows": st.r1,
                       "parallelism": "batch-shard x%d" % world,
                       "schedule": "index-order" if args.index_order else
                       ("queue of resident workgroups; order bin-packed from the iteration counts of an earlier step, renewed every 4th launch"
                        if layout["waves_per_cu"] == 1 else
                        "hardware dispatch (several workgroups per CU), longest-first from the iteration counts of an earlier step"),
                       "allgather_tau": bool(gather_state["ok"]), "lds_bytes_per_qp": layout["lds_bytes"],
                       "qps_resident_per_cu": layout["waves_per_cu"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "wbcqp::solve_queue_kernel<double>", "kernel_ms": kern_avg_s * 1e3,
                         "algorithmic_bytes_per_qp": abytes},
            "active_set": {"iters_mean": float(iters.mean()), "iters_max": int(iters.max()),
                           "iters_hist": np.bincount(np.minimum(iters, 15), minlength=16).tolist(),
                           "status_optimal": int((status == 0).sum()), "batch": int(B)},
        }
        flops = st.flops_estimate(float(iters.mean()))
        result["fp64"] = {"flops_per_qp": flops, "achieved_TFLOPs": flops * B / kern_avg_s / 1e12,
                          "note": "analytic useful flops (Structure.flops_estimate) at the batch's mean iteration count; the path is a "
                                  "chain of dependent operations, see DESIGN.md section 4"}
        if gather_state["err"]:
            result["config"]["allgather_error"] = gather_state["err"]
        if world == 1 and args.robot == "talos":
            # after the path: state integration kernel on the solver's own output, outside `value`
            nvv, nq = st.nv, st.nv + 1
            gq = torch.zeros(B, nq, dtype=torch.float64, device=dev)
            gq[:, 6] = 1.0
            gdq = torch.zeros(B, nvv, dtype=torch.float64, device=dev)
            gqn, gvn, gqs = torch.zeros_like(gq), torch.zeros_like(gdq), torch.zeros_like(gdq)
            sp = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                h.integrate(B, nvv, True, 1e-3, gq, gdq, d_out["x"], st.n, d_out["status"], gqn, gvn, gqs, stream=sp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                h.integrate(B, nvv, True, 1e-3, gq, gdq, d_out["x"], st.n, d_out["status"], gqn, gvn, gqs, stream=sp)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            ibytes = B * (8 * (2 * nq + 4 * nvv) + 4)  # q, dq, dv in; q_next, v_next, q_solver out; status
            result["after_path"] = {"kernel": "wbcqp::integrate_kernel<double>", "us_per_launch": us, "bytes_per_launch": ibytes,
                                    "achieved_GBps": ibytes / (us * 1e-6) / 1e9, "bound": "hbm (launch-latency sized at this batch)"}
        if world == 1 and args.robot == "talos":
            try:  # secondary section: never allowed to take the headline line down with it
                result["before_path"] = before_path(h, st, dev, B, torch, not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                result["before_path"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.index_order and not args.no_compare:
            # the same K steps (a) in plain index order, (b) longest-first but one workgroup per QP through the hardware's
            # dispatcher instead of resident workgroups and a queue -- both reported beside `value`
            for key, flags, note in (
                    ("index_order", capi.FLAG_INDEX_ORDER, "same batch, QPs taken in index order (no schedule from the previous step)"),
                    ("hw_dispatch", capi.FLAG_HW_DISPATCH, "same batch, longest-first, one workgroup per QP dealt out by the hardware "
                                                           "(XCD i % 8, shader engine (i / 8) % 4, in order) instead of the queue"),
                    ("queue_longest_first", capi.FLAG_NO_PACKING, "same batch, queue, plain longest-first order (no bin packing)"),
                    ("refresh_every_launch", capi.flag_refresh(1), "same batch, queue, packed order renewed after every launch "
                                                                    "(default: every 4th, WBCQP_FLAG_REFRESH)")):
                h2 = capi.Handle(device=local_rank, dtype=capi.F64, flags=flags)
                h2.set_structure(0, st)
                for _ in range(args.warmup):
                    h2.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    h2.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                result[key] = {"value": B * args.steps / (time.perf_counter() - t1), "unit": "QP/s", "note": note}
                h2.close()

        if not args.no_cpu_baseline and world == 1:
            # the oracle is the checker here and the reported CPU baseline -- never the thing shipped
            from oracle import oracle
            oracle.build()
            cores = os.cpu_count() or 1
            nsamp = min(B, 256)
            sub = {k: v[:nsamp] for k, v in inputs.items()}
            t1 = time.perf_counter()
            ref = oracle.tick_batch(st, sub, nthreads=1)
            single = nsamp / (time.perf_counter() - t1)
            # all-core run, repeated until the sample is ~cpu_seconds of CPU wall time (bounded)
            reps, done, tcpu = 0, 0, 0.0
            budget = max(1.0, args.cpu_seconds - nsamp / single)
            while tcpu < budget and reps < 64:
                t1 = time.perf_counter()
                oracle.tick_batch(st, inputs, nthreads=cores)
                tcpu += time.perf_counter() - t1
                done += B
                reps += 1
            multi = done / tcpu
            ok = ref["status"] == 0
            xs = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
            result["cpu_baseline"] = {
                "value": multi, "unit": "QP/s", "cores": cores, "kind": "port",
                "sample": "%d x %d Talos QPs of the same batch, %d pthreads; single-thread %.0f QP/s on %d QPs" % (reps, B, cores, single, nsamp)
                          if args.robot == "talos" else "%d x %d QPs, %d pthreads" % (reps, B, cores),
                "single_thread": single,
            }
            result["parity"] = {
                "sample": nsamp,
                "max_rel_dx": float((np.abs(x_gpu[:nsamp] - ref["x"]).max(axis=1) / xs)[ok].max()),
                "max_abs_dtau": float(np.abs(tau_gpu[:nsamp] - ref["tau"])[ok].max()) if st.na else 0.0,
                "status_equal": bool(np.array_equal(status[:nsamp], ref["status"])),
                "iters_equal_frac": float((iters[:nsamp] == ref["iters"]).mean()),
            }

        if args.sweep and world == 1:
            table = []
            for b in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
                si = synth.generate(st, min(b, 1024), synth.SEED_BASE[seed_key])
                reps_in = (b + 1023) // 1024
                di = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (reps_in, 1))[:b])).to(dev) for k, v in si.items() if v.size}
                do = dict(x=torch.zeros(b, st.n, dtype=torch.float64, device=dev),
                          tau=torch.zeros(b, max(st.na, 1), dtype=torch.float64, device=dev),
                          status=torch.zeros(b, dtype=torch.int32, device=dev), iters=torch.zeros(b, dtype=torch.int32, device=dev))
                for _ in range(4):
                    h.solve_batch(0, b, di, do, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nrep = 20
                for _ in range(nrep):
                    h.solve_batch(0, b, di, do, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / nrep
                table.append({"batch": b, "ms": dt * 1e3, "qps": b / dt})
            os.makedirs(os.path.dirname(os.path.abspath(args.sweep)), exist_ok=True)
            with open(args.sweep, "w") as fh:
                json.dump(table, fh, indent=1)
        print(json.dumps(result), flush=True)

    h.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
