#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the batched whole-body-QP tick on MI355X.

One "step" = one pass of the hot path (assemble H,g -> GI active-set solve -> torque decode, i.e.
controller.cpp:244-251 of the reference for every instance) over one batch of synthetic QPs that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: Talos pos-tracker, batch 1024, fp64, 1 x MI355X.

The timed steps are a STREAM, not a replay: instance i at step t solves tick (i + t) of the squat reference stream
(etc/talos/squat.yaml through move_com.cpp:22-45, BASELINE config 4's own definition), so the launch order the library
derives from earlier launches never comes from the batch it schedules.  The replayed-batch figure (perfect foresight),
the index-order figure (no schedule) and a run over unrelated batches stand beside `value`.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --robot icub --batch 4096 --dtype f32          (BASELINE config 3)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the batch shards embarrassingly -- every rank owns `--batch` QPs of the same seeded stream (weak scaling), no
collective on the solve path (the QPs of a batch are independent).  `--allgather` adds the optional exchange step of
BASELINE config 4 (all-gather of joint torques over RCCL/xGMI through the library's own wbcqp_allgather_tau) after the
solve inside the timed step.  Rank 0 prints ONE JSON line.
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
# compact-boundary bytes per fp64 QP (Talos: inputs 34 200 B + outputs 952 B)
ALGORITHMIC_BYTES = {"talos": 35152, "icub": 23872, "franka": 1152}
N_TICKS = 128  # distinct consecutive ticks of the reference stream kept resident (the timed steps cycle through them)
SWEEP = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--batch", type=int, default=1024, help="QPs per GPU per step")
    p.add_argument("--robot", default="talos", choices=["talos", "icub", "franka"])
    p.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                   help="boundary type of every input / output array (the solve itself runs in f64, DESIGN.md section 3)")
    p.add_argument("--replay", action="store_true", help="solve the same batch every step (the round-1 headline) instead of the tick stream")
    p.add_argument("--squat", action="store_true", help="accepted for compatibility: the tick stream IS the squat reference stream")
    p.add_argument("--allgather", action="store_true",
                   help="N > 1: also all-gather the joint torques over RCCL inside every step (BASELINE config 4's optional\n"
                        "exchange; the path itself has none -- the QPs of a batch are independent)")
    p.add_argument("--no-exchange", action="store_true", help="N > 1: skip the all-gather leg that is timed apart from the solve")
    p.add_argument("--exchange-timeout", type=float, default=120.0, help="seconds after which the all-gather leg is abandoned")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-compare", action="store_true", help="skip the extra runs reported beside `value`")
    p.add_argument("--no-sweep", action="store_true", help="skip the batch 1..8192 table")
    p.add_argument("--headline-only", action="store_true", help="the warm-up and the K timed steps and nothing else on the device (no comparison lines, no\n"
                   "before/after-path sections, no CPU baseline): the command the rocprofv3 passes of tools/profile_round.sh trace, so that\n"
                   "the kernel's average duration in the trace is the duration of the timed launches")
    p.add_argument("--index-order", action="store_true",
                   help="launch the QPs in index order instead of longest-first (WBCQP_FLAG_INDEX_ORDER)")
    p.add_argument("--flags", type=int, default=0, help="extra wbcqp_desc.flags for the headline handle")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU-baseline sample")
    p.add_argument("--traffic", type=float, default=None,
                   help="HBM bytes per launch from rocprofv3 PMC passes (default: scaled from profiles/pmc_latest.json)")
    p.add_argument("--sweep", default=None, help="also write the batch 1..8192 table to this JSON file")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to\n"
                   "exercise the N > 1 code path where RCCL cannot run, e.g. several ranks on one GPU)")
    p.add_argument("--single-device", action="store_true", help="testing only: every rank uses cuda:0 (with --backend gloo)")
    return p.parse_args()


def before_path(h, st, dev, B, torch, cpu_baseline=True):
    """Before the path, outside `value`: the rows kernel (q, v, references -> QP record) on a
    Talos-like tree, alone and chained with the solve and the state integration (one whole control tick on the device),
    open loop on fixed states and CLOSED loop (q_next, v_next fed back on the device, the squat reference advancing)."""
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
    # closed loop: the integrated state is the next tick's state (ping-pong buffers, nothing leaves the device)
    try:
        sa = {"q": state["q"].clone(), "v": state["v"].clone(), "ref": state["ref"]}
        sb = {"q": torch.zeros_like(sa["q"]), "v": torch.zeros_like(sa["v"]), "ref": state["ref"]}

        def closed(n_ticks):
            a, b = sa, sb
            for _ in range(n_ticks):
                h.problem_data(1, B, a, rows, stream=sp)
                h.solve_batch(1, B, rows, out, stream=sp)
                h.integrate(B, st.nv, True, tm.dt, a["q"], a["v"], out["x"], st.n, out["status"], b["q"], b["v"], None, stream=sp)
                a, b = b, a

        closed(10)
        torch.cuda.synchronize()
        it0 = out["iters"].cpu().numpy().astype(np.float64)
        closed(1)
        torch.cuda.synchronize()
        it1 = out["iters"].cpu().numpy().astype(np.float64)
        e0.record()
        closed(200)
        e1.record()
        torch.cuda.synchronize()
        msc = e0.elapsed_time(e1) / 200
        res["closed_loop"] = {"ticks_per_s": B / (msc * 1e-3), "ms_per_batch": msc, "ticks": 200,
                              "status_optimal_last": int((out["status"] == 0).sum().item()),
                              "iters_mean_last": float(out["iters"].float().mean().item()),
                              "iters_tick_to_tick_corr": _corr(it0, it1), "iters_tick_to_tick_equal": float((it0 == it1).mean()),
                              "note": "q_next, v_next fed back on the device for 200 ticks: every launch solves QPs no earlier launch has seen"}
    except Exception as e:  # noqa: BLE001
        res["closed_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if cpu_baseline:
        from oracle import rbd
        ns = min(B, 128)
        t1 = time.perf_counter()
        ora = rbd.task_rows(m, tm, st, s["q"][:ns], s["v"][:ns], s["ref"][:ns], n_threads=1)
        dt1 = time.perf_counter() - t1
        h.problem_data(1, B, state, rows, stream=sp)
        torch.cuda.synchronize()
        got = {k: rows[k][:ns].cpu().numpy() for k in capi.ROW_FIELDS}
        res["cpu_port"] = {"instances_per_s": ns / dt1, "cores": 1, "sample": "%d instances, oracle/rbd_oracle.c" % ns}
        res["parity"] = {k: float(np.abs(got[k] - ora[k]).max() / max(1.0, np.abs(ora[k]).max())) for k in capi.ROW_FIELDS if ora[k].size}  # (Acop: empty without a cop task)
    return res


def _corr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.std() == 0.0 or b.std() == 0.0:
        return 1.0 if np.array_equal(a, b) else 0.0
    return float(np.corrcoef(a, b)[0, 1])


def usable_cores():
    """(threads to use, how that number was found): the CPUs this process may run on, capped by the container's CPU quota
    (cgroup cpu.max) -- os.cpu_count() alone reports the machine, not what the container is given."""
    n = os.cpu_count() or 1
    note = "os.cpu_count() = %d" % n
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, note = aff, note + ", affinity mask %d" % aff
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota = txt[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                    period = float(fh2.read().split()[0])
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    n, note = q, note + ", cgroup CPU quota %d" % q
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, note


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def other_configs(local_rank, torch, parity=True):
    """BASELINE configs 3 (iCub, foot contacts, B = 4096, fp32 boundary) and 5 (ragged mix, B = 8192) in the default line, each with
    its own fraction of the HBM roofline and a parity sample against the oracle.  Config 3: 1024 generated QPs tiled four times,
    instance i at step t on tick (i + 17 t) of the squat CoM stream like the headline; inputs rounded to f32 for both sides."""
    from inria_wbc_amd import capi, structure, synth
    dev = torch.device("cuda", local_rank)
    out = {}
    st = structure.STRUCTURES["icub"]()
    Bg, Bt, nt = 1024, 4096, 8
    inp = synth.generate(st, Bg, synth.SEED_BASE["icub"])
    inp = {k: v.astype(np.float32).astype(np.float64) for k, v in inp.items()}
    com_rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    table = np.stack([synth.squat_com_rhs(st, t, st.kp.get("com", 30.0)) for t in range(4000)])
    base = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (Bt // Bg, 1)).astype(np.float32))).to(dev) for k, v in inp.items() if v.size}
    b1s = []
    for t in range(nt):
        b1 = np.tile(inp["b1"], (Bt // Bg, 1))
        b1[:, com_rows] += table[(np.arange(Bt) + 17 * t) % 4000][:, :com_rows.size]
        b1s.append(b1.astype(np.float32))
    dicts = []
    for t in range(nt):
        d = dict(base)
        d["b1"] = torch.from_numpy(np.ascontiguousarray(b1s[t])).to(dev)
        dicts.append(d)
    o = dict(x=torch.zeros(Bt, st.n, dtype=torch.float32, device=dev), tau=torch.zeros(Bt, st.na, dtype=torch.float32, device=dev),
             status=torch.full((Bt,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(Bt, dtype=torch.int32, device=dev),
             active_mask=torch.zeros(Bt, 8, dtype=torch.int32, device=dev))
    h = capi.Handle(device=local_rank, dtype=capi.F32)
    h.set_structure(0, st)
    sp = torch.cuda.current_stream().cuda_stream
    for t in range(8):
        h.solve_batch(0, Bt, dicts[t % nt], o, stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nrep = 48
    t0 = time.perf_counter()
    e0.record()
    for t in range(nrep):
        h.solve_batch(0, Bt, dicts[t % nt], o, stream=sp)
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nrep
    kms = e0.elapsed_time(e1) / nrep
    ab = (ALGORITHMIC_BYTES["icub"] - 8) // 2 + 8
    it = o["iters"].cpu().numpy()
    c3 = {"workload": "icub_pos_tracker_b4096_fp32_boundary_squat_tick_stream", "value": Bt / dt, "unit": "QP/s", "ms_per_step": dt * 1e3, "kernel_ms": kms,
          "algorithmic_bytes_per_qp": ab, "frac_hbm": ab * Bt / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "dtype": "f64 arithmetic, f32 arrays in HBM",
          "iters_mean": float(it.mean()), "iters_max": int(it.max()), "status_optimal": int((o["status"] == 0).sum().item())}
    if parity:
        from oracle import oracle
        last = (nrep - 1) % nt
        ns = 64
        sub = {k: v[:ns].copy() for k, v in inp.items()}
        sub["b1"] = b1s[last][:ns].astype(np.float64)
        ref = oracle.tick_batch(st, sub)
        xg = o["x"][:ns].cpu().numpy().astype(np.float64)
        okr = ref["status"] == 0
        c3["parity"] = {"sample": ns, "max_rel_dx": float((np.abs(xg - ref["x"]).max(axis=1) / np.maximum(1.0, np.abs(ref["x"]).max(axis=1)))[okr].max()),
                        "status_equal": bool(np.array_equal(o["status"][:ns].cpu().numpy(), ref["status"])),
                        "iters_equal_frac": float((o["iters"][:ns].cpu().numpy() == ref["iters"]).mean()),
                        "active_set_equal_frac": (float((o["active_mask"][:ns].cpu().numpy().view(np.uint32) == ref["active_mask"]).all(axis=1)[okr].mean())
                                                  if "active_mask" in o else None),
                        "tolerance": "1e-3 relative (SURVEY 8(d): fp32 boundary; outputs are rounded to f32)"}
    h.close()
    out["config3_icub_b4096_f32"] = c3
    from tools import ragged_bench
    ns5 = argparse.Namespace(batch=8192, steps=30, no_parity=not parity)
    out["config5_ragged_b8192"] = ragged_bench.run(ns5)
    out["icub_single_support_b4096_f32"] = three_per_cu(local_rank, torch)
    return out


def three_per_cu(local_rank, torch):
    """Config 3's robot on ONE foot (walk_on_spot's single-support stack: n 50, nEq 12), the humanoid stack whose workgroup leaves room for a
    third one on a CU (52.9 KB of LDS): the queue takes solve_queue3_kernel for it (csrc/wbcqp_device.hpp).  Same batch through the two-per-CU
    kernel (hardware dispatch) beside it, and whether the two gave the same bits."""
    from inria_wbc_amd import capi, structure, synth
    dev = torch.device("cuda", local_rank)
    st = structure.icub_structure(single_support=True)
    Bg, Bt = 1024, 4096
    inp = synth.generate(st, Bg, synth.SEED_BASE["icub"] + 500_000)
    base = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (Bt // Bg, 1)).astype(np.float32))).to(dev) for k, v in inp.items() if v.size}
    sp = torch.cuda.current_stream().cuda_stream
    res, xs = {}, []
    for name, flags in (("value", 0), ("two_per_cu_value", capi.FLAG_HW_DISPATCH)):
        o = dict(x=torch.zeros(Bt, st.n, dtype=torch.float32, device=dev), tau=torch.zeros(Bt, st.na, dtype=torch.float32, device=dev),
                 status=torch.full((Bt,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(Bt, dtype=torch.int32, device=dev))
        h = capi.Handle(device=local_rank, dtype=capi.F32, flags=flags)
        h.set_structure(0, st)
        for _ in range(6):
            h.solve_batch(0, Bt, base, o, stream=sp)
        torch.cuda.synchronize()
        nrep = 40
        t0 = time.perf_counter()
        for _ in range(nrep):
            h.solve_batch(0, Bt, base, o, stream=sp)
        torch.cuda.synchronize()
        res[name] = Bt * nrep / (time.perf_counter() - t0)
        xs.append(o["x"].cpu().numpy())
        if not flags:
            res["status_optimal"] = int((o["status"] == 0).sum().item())
            res["iters_mean"] = float(o["iters"].float().mean().item())
        h.close()
    lay = capi.layout_of(st)
    return {"workload": "icub_single_support_b4096_fp32_boundary", "unit": "QP/s", "waves_per_cu": lay["waves_per_cu"], "lds_bytes": lay["lds_bytes"],
            "same_bits_as_two_per_cu": bool(np.array_equal(xs[0], xs[1])), **res}


def warm_start(local_rank, torch, st, tick_dicts, new_out, B, cdt, base_flags, steps, inputs, b1_ticks, parity=True):
    """The same tick stream with WBCQP_FLAG_WARM_START: the active set of an instance's previous tick (a 256-bit mask per instance that
    stays on the device, in / out of every launch) goes first among the violated rows.  Not what the reference does -- reported
    beside `value`.  The QP is strictly convex, so the solution is the cold start's up to rounding, EXCEPT where eiquadprog's own
    stopping rule (|sum min(s, 0)| <= nIneq eps tr(H) tr(J) 100, about 0.3 for these stacks) ends the two pick orders on different
    iterates: `parity` counts those QPs instead of hiding them."""
    from inria_wbc_amd import capi
    dev = torch.device("cuda", local_rank)
    sp = torch.cuda.current_stream().cuda_stream
    h = capi.Handle(device=local_rank, dtype=cdt, flags=base_flags | capi.FLAG_WARM_START)
    h.set_structure(0, st)
    out = new_out()
    out["active_mask"] = torch.zeros(B, 8, dtype=torch.int32, device=dev)
    nt = len(tick_dicts)
    for t in range(8):
        h.solve_batch(0, B, tick_dicts[t % nt], out, stream=sp)
    torch.cuda.synchronize()
    n2 = max(steps, 100)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for t in range(n2):
        h.solve_batch(0, B, tick_dicts[(8 + t) % nt], out, stream=sp)
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n2
    it = out["iters"].cpu().numpy()
    res = {"value": B / dt, "unit": "QP/s", "steps": n2, "kernel_ms": e0.elapsed_time(e1) / n2, "iters_mean": float(it.mean()), "iters_max": int(it.max()),
           "iters_hist": np.bincount(np.minimum(it, 15), minlength=16).tolist(), "status_optimal": int((out["status"] == 0).sum().item()),
           "note": "opt-in pick priority for the previous tick's active set (WBCQP_FLAG_WARM_START); eiquadprog-fast has no such rule: never the headline"}
    if parity:
        from oracle import oracle
        last = (8 + n2 - 1) % nt
        ns = min(B, 256)
        cpu_in = {k: v[:ns].copy() for k, v in inputs.items()}
        cpu_in["b1"] = b1_ticks[last][:ns].cpu().numpy().astype(np.float64)
        ref = oracle.tick_batch(st, cpu_in, nthreads=8)
        x = out["x"][:ns].cpu().numpy().astype(np.float64)
        e = np.abs(x[:, :st.nv] - ref["x"][:, :st.nv]).max(axis=1) / np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
        res["parity"] = {"sample": ns, "tick": int(last), "status_equal": bool(np.array_equal(out["status"][:ns].cpu().numpy(), ref["status"])),
                         "frac_within_1e-8_of_cold_oracle": float((e <= 1e-8).mean()), "max_rel_dx": float(e.max()),
                         "iters_cold_mean": float(ref["iters"].mean()), "iters_cold_max": int(ref["iters"].max()),
                         "iters_warm_mean": float(it[:ns].mean()), "iters_warm_max": int(it[:ns].max())}
    h.close()
    return res


def dense_seam(local_rank, st, inputs, cpu=True):
    """Latency of ONE Talos QP through the two host-pointer seams: `wbcqp_solve_dense_host` -- what stands behind
    solver_->solve(HQPData) (controller.cpp:247), the reference's own use case at n_qp = 1 -- and `wbcqp_solve_batch_host`
    (structured record in, batch 1), next to the CPU restatement on one thread.  Both sides are timed FROM C: the GPU side by the
    facade's harness (qp_timer_test --dense: the reference's Timer around each call, no ctypes between two calls), the CPU side by
    loops inside the oracle library (wbco_tick_batch_timed on one thread, wbco_eiquadprog_timed), each warm and >= 200 repetitions.
    The figures through the Python binding are kept beside them under `through_ctypes` (round 3 quoted those: the CPU ones were
    about twice the C clock's).  Wall time per call, PCIe staging included."""
    import subprocess
    import tempfile
    from inria_wbc_amd import build, capi
    from oracle import oracle
    from tools import dump_batch, emit_configs
    one = {k: v[:1] for k, v in inputs.items()}
    H, g, CE, ce0, CI, ci0 = [np.ascontiguousarray(a) for a in oracle.assemble(st, inputs, 0)]
    res = {"qp": "QP 0 of the headline batch (Talos, n %d, neq %d, %d one-sided rows)" % (st.n, st.neq, st.nin2),
           "clock": "C on both sides: qp_timer_test --dense (utils::Timer around each call) and loops inside the oracle library"}
    # ---- GPU side, from C ----
    try:
        host = build.build_host()
        emit_configs.main()
        with tempfile.TemporaryDirectory() as td:
            dq, db = os.path.join(td, "dense.bin"), os.path.join(td, "batch.bin")
            with open(dq, "wb") as fh:
                np.array([st.n, st.neq, st.nin2], dtype=np.int64).tofile(fh)
                for a in (H, g, CE, ce0, CI, ci0):
                    np.ascontiguousarray(a, dtype=np.float64).tofile(fh)
            dump_batch.dump(db, st, one)
            r = subprocess.run([host["qp_timer_test"], "--dense", dq, db, os.path.join(ROOT, "configs", "talos", "tasks.yaml"), str(st.nv), str(st.na), "500"],
                               capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError((r.stdout + r.stderr)[-400:])
        for ln in r.stdout.splitlines():
            w = ln.split()
            if w and w[0] in ("dense_host_us:", "batch_host_us:"):
                key = "solve_dense_host" if w[0].startswith("dense") else "solve_batch_host"
                res[key + "_us"] = float(w[1])
                res[key + "_min_us"] = float(w[3])
            elif w and w[0] == "status":
                res["iters"] = int(w[5])
                res["agree_max_dx"] = float(w[-1])
    except Exception as e:  # noqa: BLE001
        res["from_c_error"] = "%s: %s" % (type(e).__name__, e)
    # ---- the same two calls through the Python binding (what round 3 reported) ----
    h = capi.Handle(device=local_rank, dtype=capi.F64)
    h.set_structure(0, st)

    def wall(fn, reps=200):
        for _ in range(10):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        return (time.perf_counter() - t0) / reps * 1e6, r

    us_d, rd = wall(lambda: h.solve_dense_host(H[None], g[None], CE[None], ce0[None], CI[None], ci0[None]))
    us_b, rb = wall(lambda: h.solve_batch_host(0, one))
    res["through_ctypes"] = {"solve_dense_host_us": us_d, "solve_batch_host_us": us_b}
    res.setdefault("iters", int(rb["iters"][0]))
    res.setdefault("agree_max_dx", float(np.abs(rd["x"][0] - rb["x"][0]).max()))
    if cpu:
        s1, ref = oracle.tick_batch_timed(st, one, nthreads=1, reps=1)  # warm: page in, first malloc
        s1, ref = oracle.tick_batch_timed(st, one, nthreads=1, reps=400)
        res["cpu_port_single_thread_us"] = s1 / 400 * 1e6  # assembly + eiquadprog + decode, in C
        se, est, eit = oracle.eiquadprog_timed(H, g, CE, ce0, CI, ci0, reps=400)
        res["cpu_port_eiquadprog_only_us"] = se * 1e6
        try:
            sn, _ = oracle.tick_batch_timed(st, one, nthreads=1, reps=400, native=True)
            res["cpu_port_single_thread_native_us"] = sn / 400 * 1e6
        except Exception:  # noqa: BLE001
            pass
        t0 = time.perf_counter()
        for _ in range(50):
            oracle.tick_batch(st, one)
        res["through_ctypes"]["cpu_port_single_thread_us"] = (time.perf_counter() - t0) / 50 * 1e6
        res["max_rel_dx_vs_oracle"] = float(np.abs(rb["x"][0] - ref["x"][0]).max() / max(1.0, np.abs(ref["x"][0]).max()))
        if "solve_batch_host_us" in res:
            res["gpu_over_cpu"] = {"structured_seam": res["cpu_port_single_thread_us"] / res["solve_batch_host_us"],
                                   "dense_seam_vs_eiquadprog_alone": res["cpu_port_eiquadprog_only_us"] / res["solve_dense_host_us"],
                                   "note": "> 1: one robot's tick is shorter through the GPU seam than on one host core (C clock on both sides)"}
    h.close()
    return res


def franka_single_tick(local_rank, torch, cpu=True):
    """The reference's one timing fixture: Franka pos-tracker, 12.6 us per tick INCLUDING kinematics on its authors' machine
    (tests/ref_test_franka.yaml:11, solver = whole behavior->update(), test_franka.cpp:175-179).  Here: one Franka instance through
    wbcqp_tick_host (rows from the model, QP, integration; wall time per call) and the device-side tick (event time), the CPU
    restatement of the same tick on one core beside them."""
    from inria_wbc_amd import capi, structure
    from inria_wbc_amd import model as mdl
    m = mdl.franka_like()
    st = structure.franka_structure()
    tm = mdl.build_taskmap(m, st, mdl.franka_stack())
    h = capi.Handle(device=local_rank, dtype=capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    s = mdl.sample_states(m, tm, 1, 77_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
    w = st.default_weights[None]
    for _ in range(20):
        h.tick_host(0, s["q"], s["v"], s["ref"], None, None, w, tm.dt)
    t0 = time.perf_counter()
    for _ in range(300):
        got = h.tick_host(0, s["q"], s["v"], s["ref"], None, None, w, tm.dt)
    host_us = (time.perf_counter() - t0) / 300 * 1e6
    dev = torch.device("cuda", local_rank)
    L = st.field_lengths()
    state = {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}
    rows = {k: torch.zeros(1, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows["w"] = torch.from_numpy(w.copy()).to(dev)
    out = dict(x=torch.zeros(1, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(1, st.na, dtype=torch.float64, device=dev),
               status=torch.zeros(1, dtype=torch.int32, device=dev), iters=torch.zeros(1, dtype=torch.int32, device=dev))
    qn, vn = torch.zeros_like(state["q"]), torch.zeros_like(state["v"])
    sp = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        h.tick(0, 1, state, rows, out, qn, vn, tm.dt, stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        h.tick(0, 1, state, rows, out, qn, vn, tm.dt, stream=sp)
    e1.record()
    torch.cuda.synchronize()
    res = {"reference_anchor_us": 12.6, "reference_anchor": "tests/ref_test_franka.yaml:11 (2021, unknown machine, incl. kinematics, one thread)",
           "tick_host_us": host_us, "tick_device_us": e0.elapsed_time(e1) / 200 * 1e3, "status": int(got["status"][0]),
           "note": "one Franka instance (n = 9, no constraints): a single small QP is launch- and PCIe-latency on a GPU -- the CPU restatement of "
                   "the same tick (cpu_port_tick_us: kinematics + rows + QP on one core, C clock) is what compares with the anchor; the batch "
                   "is what the device is for (Franka B = 8192: 373 M QP/s)"}
    if cpu:
        from oracle import oracle, rbd
        nrep = 4000  # the same state 4000 times in ONE call of the C batch driver (one thread): the per-instance cost without ctypes
        many = {k: np.repeat(s[k], nrep, axis=0) for k in ("q", "v", "ref")}
        rbd.task_rows(m, tm, st, many["q"][:64], many["v"][:64], many["ref"][:64])
        t0 = time.perf_counter()
        rws = rbd.task_rows(m, tm, st, many["q"], many["v"], many["ref"], n_threads=1)
        res["cpu_port_rows_us"] = (time.perf_counter() - t0) / nrep * 1e6
        one = {k: v[:1] for k, v in rws.items()}
        s1, _ = oracle.tick_batch_timed(st, dict(one, tlb=np.zeros((1, 0)), tub=np.zeros((1, 0)), w=w), nthreads=1, reps=4000)
        res["cpu_port_qp_us"] = s1 / 4000 * 1e6
        res["cpu_port_tick_us"] = res["cpu_port_rows_us"] + res["cpu_port_qp_us"]
    h.close()
    return res


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def child_command(n, port, argv):
    """The command line `python bench.py --gpus N ...` turns itself into when it is typed without a launcher: the driver's own
    (one process per GPU under torch.distributed.run), same flags."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` typed as is (no WORLD_SIZE): this process never touches the GPU -- it starts the N ranks as a
    CHILD process (no exec), passes their stderr through, prints rank 0's JSON line and returns the child's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = child_command(args.gpus, free_port(), argv)
    print("bench.py: --gpus %d without WORLD_SIZE: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT)
    line, other = None, []
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            try:
                json.loads(t)
                line = t
                continue
            except ValueError:
                pass
        other.append(ln)
    rc = proc.wait()
    for ln in other:  # whatever else the ranks wrote to stdout is not the result line
        sys.stderr.write(ln)
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        print("bench.py: the %d-rank child printed no JSON line" % args.gpus, file=sys.stderr)
    return rc


def exchange_leg(torch, dist, h, gather, rank, world, dev, d_out, st, B, tdt, sp, step, fence, timeout_s):
    """All-gather of the joint torques on every rank: `reps` calls of wbcqp_allgather_tau alone, then the same number
    of solve + all-gather steps; returns the figures (rank 0 reports them) or {"error": ...}.  Runs in a worker thread so that a
    collective that never returns is abandoned after `timeout_s` (the process then leaves through os._exit at the end of main)."""
    import threading
    res = {}

    def work():
        try:
            from inria_wbc_amd import rccl
            torch.cuda.set_device(dev)  # the HIP device is a per-thread setting
            comm = gather["comm"]
            if comm is None:
                comm = rccl.comm_from_torch(dist, rank, world, dev)
            na = max(st.na, 1)
            tau_all = torch.zeros(world * B, na, dtype=tdt, device=dev)
            reps = 50
            for _ in range(5):
                h.allgather_tau(comm, d_out["tau"].data_ptr(), tau_all.data_ptr(), B * na, stream=sp)
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                h.allgather_tau(comm, d_out["tau"].data_ptr(), tau_all.data_ptr(), B * na, stream=sp)
            e1.record()
            fence()
            us = e0.elapsed_time(e1) / reps * 1e3
            # the gathered block of every rank must be that rank's tau: rank r's own block against what it sent
            own_ok = bool(torch.equal(tau_all[rank * B:(rank + 1) * B], d_out["tau"]))
            t0 = time.perf_counter()
            for t in range(reps):
                step(t)
                if not gather["ok"]:  # --allgather already gathers inside the step
                    h.allgather_tau(comm, d_out["tau"].data_ptr(), tau_all.data_ptr(), B * na, stream=sp)
            fence()
            both = (time.perf_counter() - t0) / reps
            esz = 4 if tdt == torch.float32 else 8
            res.update({"us_per_allgather": us, "bytes_per_rank": B * na * esz, "bytes_gathered": world * B * na * esz,
                        "GBps_per_rank_in": (world - 1) * B * na * esz / (us * 1e-6) / 1e9,
                        "rccl_nranks": rccl.comm_count(comm), "own_block_intact": own_ok,
                        "solve_plus_allgather_qps": world * B / both, "ms_per_step_with_allgather": both * 1e3,
                        "note": "wbcqp_allgather_tau (RCCL ncclAllGather over xGMI) of the joint torques, timed apart from the solve: `value` is "
                                "the path, which has no collective; solve_plus_allgather_qps is a step that also gathers"})
        except Exception as e:  # noqa: BLE001
            res["error"] = "%s: %s" % (type(e).__name__, e)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return {"error": "no answer from the exchange leg within %.0f s (abandoned)" % timeout_s, "hung": True}
    return res


def main():
    args = parse()
    if args.headline_only:
        args.no_compare = args.no_sweep = args.no_cpu_baseline = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d under a launcher that started %d ranks (WORLD_SIZE): one process per GPU -- launch it as\n"
                         "  python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port 29500 "
                         "bench.py --gpus %d ...\nor type `python bench.py --gpus %d ...` alone and it starts the ranks itself"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    from inria_wbc_amd import build, capi, structure, synth
    if rank == 0:
        build.build()
    if distributed:
        dist.barrier()

    f32 = args.dtype == "f32"
    tdt = torch.float32 if f32 else torch.float64
    ndt = np.float32 if f32 else np.float64
    cdt = capi.F32 if f32 else capi.F64
    st = structure.STRUCTURES[args.robot]()
    B = args.batch
    dev = torch.device("cuda", local_rank)
    # rank r owns QPs [r*B, (r+1)*B) of the stream
    inputs = synth.generate(st, B, synth.SEED_BASE[args.robot], first=rank * B)
    if f32:
        inputs = {k: v.astype(np.float32).astype(np.float64) for k, v in inputs.items()}  # the oracle sees what the device sees
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v.astype(ndt))).to(dev) for k, v in inputs.items() if v.size}

    # ---- the tick stream: CoM rows of b1 follow the squat reference, instance i at step t is on tick (rank B + i + t) ----
    stream = (not args.replay) and ("com" in st.task_names)
    com_rows = np.where(st.dense_row_task == st.task_names.index("com"))[0] if stream else None
    b1_ticks = None
    if stream:
        kp = st.kp.get("com", 30.0)
        table = np.stack([synth.squat_com_rhs(st, t, kp) for t in range(4000)])  # [4000, 3]
        idx = (rank * B + np.arange(B)[None, :] + np.arange(N_TICKS)[:, None]) % 4000  # [tick, instance]
        b1_all = np.repeat(inputs["b1"][None, :, :], N_TICKS, axis=0)
        b1_all[:, :, com_rows] += table[idx][:, :, :com_rows.size]
        if f32:
            b1_all = b1_all.astype(np.float32).astype(np.float64)
        b1_ticks = torch.from_numpy(np.ascontiguousarray(b1_all.astype(ndt))).to(dev)

    def tick_inputs(t):
        if not stream:
            return d_in
        d = dict(d_in)
        d["b1"] = b1_ticks[t % N_TICKS]
        return d

    tick_dicts = [tick_inputs(t) for t in range(N_TICKS if stream else 1)]

    def new_out(b=B):
        return dict(x=torch.zeros(b, st.n, dtype=tdt, device=dev), tau=torch.zeros(b, max(st.na, 1), dtype=tdt, device=dev),
                    status=torch.full((b,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(b, dtype=torch.int32, device=dev))

    d_out = new_out()
    # the rest of HQPOutput (objective, size of the active set, the active one-sided rows as a 256-bit mask): written by the timed launches too --
    # 44 bytes per QP beside the 35 152 algorithmic ones, not counted in `roofline.achieved` -- so that the parity sample below compares the
    # active set of the LAST TIMED launch with the oracle's (identical active set)
    d_out.update(objective=torch.zeros(B, dtype=tdt, device=dev), n_active=torch.zeros(B, dtype=torch.int32, device=dev),
                 active_mask=torch.zeros(B, 8, dtype=torch.int32, device=dev))
    base_flags = args.flags | (capi.FLAG_INDEX_ORDER if args.index_order else 0)
    h = capi.Handle(device=local_rank, dtype=cdt, flags=base_flags)
    h.set_structure(0, st)
    layout = capi.layout_of(st)
    sp = torch.cuda.current_stream().cuda_stream

    # optional exchange step: the library's own RCCL entry point on a communicator of this job's ranks
    gather = {"ok": False, "err": None, "comm": None}
    tau_all = None
    exchange_skipped = None  # why the line carries no all-gather figures although the run is distributed (never silently absent)
    if distributed and args.backend != "nccl":
        exchange_skipped = ("backend %s%s: wbcqp_allgather_tau is RCCL's ncclAllGather and needs one GPU per rank (backend nccl); the solve path itself has no "
                            "collective, so `value` is unaffected" % (args.backend, ", every rank on cuda:0 (--single-device)" if args.single_device else ""))
    elif distributed and args.no_exchange and not args.allgather:
        exchange_skipped = "--no-exchange"
    if distributed and args.allgather and exchange_skipped is None:
        try:
            from inria_wbc_amd import rccl
            gather["comm"] = rccl.comm_from_torch(dist, rank, world, dev)
            tau_all = torch.zeros(world * B, max(st.na, 1), dtype=tdt, device=dev)
            gather["ok"] = True
        except Exception as e:  # keep the scaling run alive; the exchange step is optional
            gather["err"] = repr(e)

    def step(t, hh=h, out=d_out):
        hh.solve_batch(0, B, tick_dicts[t % len(tick_dicts)], out, stream=sp)
        if gather["ok"]:
            hh.allgather_tau(gather["comm"], out["tau"].data_ptr(), tau_all.data_ptr(), B * max(st.na, 1), stream=sp)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, warm, hh=h, out=d_out, t_first=0):
        """(wall seconds, event seconds) of exactly n_steps steps after `warm` untimed ones"""
        for t in range(warm):
            step(t_first + t, hh, out)
        fence()
        # one HIP event pair over the timed region, on the stream the kernels are launched on (one pair per step left a
        # 10 us bubble in front of every solve kernel -- rocprofv3 trace, round 1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for t in range(n_steps):
            step(t_first + warm + t, hh, out)
        ev1.record()
        fence()
        el = time.perf_counter() - t0
        return el, ev0.elapsed_time(ev1) * 1e-3

    # ---- where in the stream the K timed steps lie.  The resident ticks are not equally heavy (the squat's later ticks ask for
    # more), so a short window is representative only where its mean launch time equals the cycle's: one untimed pass measures
    # every resident tick's launch (an event pair per launch), and the window starts at the phase whose mean over the K steps
    # behind the warm-up is closest to the mean over the whole cycle.  `value / long_run` is printed beside it.
    t_first, window = 0, None
    if stream and args.steps < len(tick_dicts) and not args.replay:
        nt = len(tick_dicts)
        for t in range(8):
            step(nt - 8 + t)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nt)]
        for t in range(nt):
            evs[t][0].record()
            step(t)
            evs[t][1].record()
        torch.cuda.synchronize()
        per = np.array([a.elapsed_time(b) for a, b in evs])
        means = np.array([per[[(s0 + args.warmup + j) % nt for j in range(args.steps)]].mean() for s0 in range(nt)])
        t_first = int(np.argmin(np.abs(means - per.mean())))
        if distributed:  # every rank times the same stretch
            tf = torch.tensor([t_first], dtype=torch.int64, device=dev)
            dist.broadcast(tf, src=0)
            t_first = int(tf.item())
        window = {"first_tick": t_first, "ms_window_mean_prepass": float(means[t_first]), "ms_cycle_mean_prepass": float(per.mean()),
                  "ms_cycle_min": float(per.min()), "ms_cycle_max": float(per.max()),
                  "note": "the timed steps start at the phase of the %d-tick cycle whose %d-step mean launch time is closest to the cycle's mean "
                          "(untimed pre-pass, one event pair per launch)" % (nt, args.steps)}
    elapsed, ev_s = timed(args.steps, args.warmup, t_first=t_first)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_avg_s = ev_s / args.steps

    # the last timed step's outputs, with the tick they belong to (parity sample below) -- on the host BEFORE the exchange leg: that leg
    # solves further ticks into the same buffers, and a collective that hangs on the device would block any later copy from its stream
    last_tick = (t_first + args.warmup + args.steps - 1) % len(tick_dicts)
    status = d_out["status"].cpu().numpy()
    iters = d_out["iters"].cpu().numpy()
    x_gpu = d_out["x"].cpu().numpy().astype(np.float64)
    tau_gpu = d_out["tau"].cpu().numpy()[:, :st.na].astype(np.float64)
    mask_gpu = d_out["active_mask"].cpu().numpy().view(np.uint32)
    nact_gpu = d_out["n_active"].cpu().numpy()
    obj_gpu = d_out["objective"].cpu().numpy().astype(np.float64)

    # ---- the optional exchange step of BASELINE config 4, timed APART from the solve (the path itself has no collective): the
    # library's own wbcqp_allgather_tau on a communicator of this job's ranks.  It has never run on more than one rank before the
    # driver's 8-GPU run, so it runs under a watchdog: a hang costs the exchange figures, never the line.
    exchange = None
    if distributed and args.backend == "nccl" and not args.no_exchange:
        exchange = exchange_leg(torch, dist, h, gather, rank, world, dev, d_out, st, B, tdt, sp, step, fence, args.exchange_timeout)


    traffic = args.traffic
    pmc = None
    if args.robot == "talos" and not f32:
        try:  # committed PMC measurement (separate --pmc passes, tools/profile_round.sh + tools/profile_kernels.sh -> tools/pmc_summary.py)
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as fh:
                pmc = json.load(fh)
        except Exception:
            pmc = None
    traffic_src = "--traffic" if traffic is not None else None
    if traffic is None and pmc is not None:
        traffic = (2.0 * pmc["fetch_size_kb"] + pmc["write_size_kb"]) * 1024.0 * B / pmc["batch"]  # scaled to this launch's batch
        traffic_src = "profiles/pmc_latest.json (committed rocprofv3 PMC passes of round %s, not measured by this run)" % pmc.get("round")

    result = None
    if rank == 0:
        total_qps = world * B * args.steps
        value = total_qps / elapsed
        ab64 = ALGORITHMIC_BYTES.get(args.robot, st.algorithmic_bytes())
        abytes = (ab64 - 8) // 2 + 8 if f32 else ab64
        achieved = abytes * B / kern_avg_s / 1e9
        per_cu = layout["waves_per_cu"]
        queued = (layout["lds_bytes"] >= 48 * 1024 or (base_flags & capi.FLAG_QUEUE)) and not (base_flags & capi.FLAG_HW_DISPATCH)
        compact = per_cu >= 2 or layout["lds_bytes"] <= 80 * 1024
        tname = "float" if f32 else "double"
        is_cp = compact and not (base_flags & capi.FLAG_FULL_LDS)
        # the name rocprofv3 prints: <boundary type, compact layout, specialisation index (0 = generic; wbcqp_layout.specialised)>
        spec = layout.get("specialised", 0) if is_cp and not (base_flags & capi.FLAG_GENERIC_KERNEL) else 0
        kernel = "wbcqp::%s<%s, %s, %d>" % ("solve_queue_kernel" if queued else "solve_kernel", tname, "true" if is_cp else "false", spec)
        if layout.get("wave_per_qp") and not (base_flags & capi.FLAG_WORKGROUP_PER_QP):
            kernel = "wbcqp::solve_small_kernel<%s>" % tname  # one wavefront per QP (csrc/wbcqp_small.hpp)
        result = {
            "metric": "QP solves/sec (Talos ~50-var WBC tick)" if args.robot == "talos" else "QP solves/sec (%s)" % args.robot,
            "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s_pos_tracker_b%d_%s_%s" % (args.robot, B, "fp32_boundary" if f32 else "fp64",
                                                                 "squat_tick_stream" if stream else "replayed_batch"),
                       "batch_per_gpu": B, "n": st.n, "neq": st.neq, "nin": st.nin, "level1_rows": st.r1,
                       "boundary_dtype": args.dtype,
                       "parallelism": "batch-shard x%d" % world,
                       "stream": ("instance i at step t solves tick (i + t) of the squat CoM reference (etc/talos/squat.yaml, move_com.cpp:22-45); "
                                  "%d consecutive ticks resident, cycled" % N_TICKS) if stream else "the same batch every step",
                       "schedule": "index-order" if args.index_order else
                       ("queue of resident workgroups (2 per CU), longest-first from the iteration counts of an earlier launch, renewed every 4th launch" if queued else
                        "hardware dispatch, longest-first from the iteration counts of an earlier launch, renewed every 4th launch"),
                       "allgather_tau": bool(gather["ok"]), "lds_bytes_per_qp": layout["lds_bytes"],
                       "qps_resident_per_cu": per_cu},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": kernel, "kernel_ms": kern_avg_s * 1e3,
                         "algorithmic_bytes_per_qp": abytes},
            "active_set": {"iters_mean": float(iters.mean()), "iters_max": int(iters.max()),
                           "iters_hist": np.bincount(np.minimum(iters, 15), minlength=16).tolist(),
                           "status_optimal": int((status == 0).sum()), "batch": int(B),
                           "note": "a launch cannot end before its longest QP does: with 512 resident workgroups a 1024-QP launch of this stream is "
                                   "bounded by the one QP that takes iters_max iterations (DESIGN.md section 4, stragglers)"},
        }
        if pmc is not None and B == pmc.get("batch") and stream and world == 1:
            # the two other rooflines SURVEY 8(d) asks for beside HBM, from the COMMITTED counter passes (same workload, same batch): the
            # counted bytes / flops per launch over THIS run's launch time
            src = "profiles/pmc_latest.json (committed rocprofv3 PMC passes of round %s; counts per launch from there, time from this run)" % pmc.get("round")
            if "lds" in pmc and "bytes_per_launch" in pmc["lds"]:
                a = pmc["lds"]["bytes_per_launch"] / kern_avg_s / 1e12
                result["roofline"]["lds"] = {"bound": "lds", "achieved": a, "peak": 150.0, "unit": "TB/s", "frac": a / 150.0,
                                             "bank_conflict_share_of_active": pmc["lds"].get("bank_conflict_share_of_active"), "source": src}
            if "fp64" in pmc:
                a = pmc["fp64"]["flops_per_launch_issued"] / kern_avg_s / 1e12
                result["roofline"]["fp64"] = {"bound": "vector f64", "achieved": a, "peak": 78.6, "unit": "TFLOP/s", "frac": a / 78.6,
                                              "f64_share_of_valu": pmc["fp64"].get("f64_share_of_valu"), "source": src,
                                              "note": "issued lane-flops (counted instructions x 64): an upper bound on useful flops"}
        flops = st.flops_estimate(float(iters.mean()))
        result["fp64"] = {"flops_per_qp": flops, "achieved_TFLOPs": flops * B / kern_avg_s / 1e12,
                          "note": "analytic useful flops (Structure.flops_estimate) at the batch's mean iteration count; the path is a "
                                  "chain of dependent operations, see DESIGN.md section 4"}
        if gather["err"]:
            result["config"]["allgather_error"] = gather["err"]
        if gather["ok"]:
            try:  # what RCCL itself says about the communicator the timed all-gathers ran on
                from inria_wbc_amd import rccl
                result["config"]["rccl_nranks"] = rccl.comm_count(gather["comm"])
            except Exception as e:  # noqa: BLE001
                result["config"]["rccl_nranks_error"] = repr(e)
        if window is not None:
            result["window"] = window
        if exchange is None and distributed:
            result["allgather_tau"] = {"skipped_reason": exchange_skipped or gather["err"] or "not requested"}
        if exchange is not None:
            result["allgather_tau"] = {k: v for k, v in exchange.items() if k != "hung"}
            if "rccl_nranks" in exchange:
                result["config"]["rccl_nranks"] = exchange["rccl_nranks"]

        if world == 1 and stream and not args.headline_only:
            # how alike are consecutive ticks?  (the launch order is built on it)
            o1, o2 = new_out(), new_out()
            h.solve_batch(0, B, tick_dicts[5], o1, stream=sp)
            h.solve_batch(0, B, tick_dicts[6], o2, stream=sp)
            h.solve_batch(0, B, tick_dicts[9], d_out, stream=sp)
            torch.cuda.synchronize()
            i1, i2, i4 = (o["iters"].cpu().numpy() for o in (o1, o2, d_out))
            result["tick_to_tick"] = {"iters_corr_1_tick": _corr(i1, i2), "iters_equal_1_tick": float((i1 == i2).mean()),
                                      "iters_corr_4_ticks": _corr(i1, i4), "iters_equal_4_ticks": float((i1 == i4).mean()),
                                      "note": "active-set iteration counts of the same instances one and four ticks apart (the launch order is at most four launches old)"}

        if world == 1 and not args.no_compare:
            # a region of >= 0.5 s of the same stream: the K-step figure above must agree with it
            n_long = max(args.steps, int(0.6 / max(kern_avg_s, 1e-6)))
            el, evl = timed(n_long, 2)
            result["long_run"] = {"value": B * n_long / el, "unit": "QP/s", "steps": n_long, "seconds": el, "kernel_ms": evl / n_long * 1e3}
            result["value_over_long_run"] = value / result["long_run"]["value"]

            def variant(flags, note, replay=False, fresh=None):
                h2 = capi.Handle(device=local_rank, dtype=cdt, flags=flags)
                h2.set_structure(0, st)
                n2 = max(args.steps, 100)
                o2 = new_out()

                def st2(t):
                    if fresh is not None:
                        h2.solve_batch(0, B, fresh[t % len(fresh)], o2, stream=sp)
                    elif replay:
                        h2.solve_batch(0, B, tick_dicts[0], o2, stream=sp)
                    else:
                        h2.solve_batch(0, B, tick_dicts[t % len(tick_dicts)], o2, stream=sp)

                for t in range(8):
                    st2(t)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for t in range(n2):
                    st2(8 + t)
                torch.cuda.synchronize()
                r = {"value": B * n2 / (time.perf_counter() - t1), "unit": "QP/s", "note": note}
                h2.close()
                return r

            Q, HW = capi.FLAG_QUEUE, capi.FLAG_HW_DISPATCH
            result["replayed_batch"] = variant(base_flags, "tick 0 of the stream every step: the order comes from the very QPs it schedules (perfect foresight; an upper bound)", replay=True)
            if stream:  # round 1's workload, unchanged: SURVEY config 2's generator without the squat reference, replayed
                plain = [{k: v for k, v in d_in.items()}]
                result["config2_replayed"] = variant(base_flags, "SURVEY 8(d) config 2 exactly as round 1 timed it (no CoM reference stream, the same batch every step; round 1: 3.46 M QP/s): "
                                                                 "mean 3.8 iterations, longest QP 21 -- the tick stream's longest QP takes 40 and bounds a 1024-QP launch", fresh=plain)
                result["config2_replayed"]["frac_hbm"] = abytes * result["config2_replayed"]["value"] / 1e9 / HBM_PEAK_GBS
                try:  # the driver's own record of round 1, when it is in the tree
                    with open(os.path.join(ROOT, "BENCH_r01.json")) as fh:
                        r1 = json.load(fh).get("parsed", {}).get("value")
                    if r1 and args.robot == "talos" and B == 1024 and not f32:
                        result["config2_replayed"]["round1_value"] = r1
                        result["config2_replayed"]["vs_round1"] = result["config2_replayed"]["value"] / r1
                except Exception:  # noqa: BLE001
                    pass
            result["index_order"] = variant(capi.FLAG_INDEX_ORDER, "same stream, QPs taken in index order (no schedule at all)")
            result["hw_dispatch"] = variant(HW, "same stream, longest-first, one workgroup per QP dealt out by the hardware's dispatcher")
            result["queue_packed"] = variant(Q, "same stream, queue, bin-packed order where the launch is small enough (default at one workgroup per CU only)")
            result["refresh_every_launch"] = variant(base_flags | capi.flag_refresh(1), "same stream, order renewed after every launch (default: every 4th, WBCQP_FLAG_REFRESH)")
            result["full_lds_layout"] = variant(capi.FLAG_FULL_LDS, "same stream on round 1's layout: every array of the QP in LDS (Talos: one QP per CU), queue + packed order")
            try:  # two INDEPENDENT fleets (two handles, two HIP streams, out of phase): their launches overlap on the chip, so one
                # fleet's stragglers run beside the other's bulk.  Not `value`: consecutive ticks of ONE fleet cannot overlap.
                s2 = torch.cuda.Stream()
                ha, hb = capi.Handle(device=local_rank, dtype=cdt, flags=base_flags), capi.Handle(device=local_rank, dtype=cdt, flags=base_flags)
                ha.set_structure(0, st)
                hb.set_structure(0, st)
                oa, ob = new_out(), new_out()
                n2 = max(args.steps, 100)

                def both(t):
                    ha.solve_batch(0, B, tick_dicts[t % len(tick_dicts)], oa, stream=sp)
                    hb.solve_batch(0, B, tick_dicts[(t + 64) % len(tick_dicts)], ob, stream=s2.cuda_stream)

                s2.wait_stream(torch.cuda.current_stream())
                for t in range(8):
                    both(t)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for t in range(n2):
                    both(8 + t)
                torch.cuda.synchronize()
                result["two_independent_fleets"] = {"value": 2 * B * n2 / (time.perf_counter() - t1), "unit": "QP/s",
                                                    "note": "two fleets of %d robots on two HIP streams, 64 ticks out of phase: launches of different "
                                                            "fleets overlap (the tail of one beside the bulk of the other)" % B}
                ha.close()
                hb.close()
            except Exception as e:  # noqa: BLE001
                result["two_independent_fleets"] = {"error": repr(e)}
            try:  # unrelated batches: every launch is scheduled from the counts of QPs that have nothing to do with it
                fresh = []
                for j in range(4):
                    fi = synth.generate(st, B, synth.SEED_BASE[args.robot] + 100_000 * (j + 1))
                    fresh.append({k: torch.from_numpy(np.ascontiguousarray(v.astype(ndt))).to(dev) for k, v in fi.items() if v.size})
                result["unrelated_batches"] = variant(base_flags, "four unrelated batches in rotation: the order of a launch comes from other QPs (a lower bound: worse than no order only if the library mis-orders)", fresh=fresh)
            except Exception as e:  # noqa: BLE001
                result["unrelated_batches"] = {"error": repr(e)}

        if world == 1 and stream and not args.no_compare:
            try:  # OPT-IN, beside the headline and never it: eiquadprog-fast has no warm start (include/wbcqp.h, WBCQP_FLAG_WARM_START)
                result["warm_start"] = warm_start(local_rank, torch, st, tick_dicts, new_out, B, cdt, base_flags, args.steps, inputs,
                                                  b1_ticks, not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                result["warm_start"] = {"error": "%s: %s" % (type(e).__name__, e)}

        if world == 1 and args.robot == "talos" and not f32 and not args.headline_only:
            # after the path: state integration kernel on the solver's own output, outside `value`
            nvv, nq = st.nv, st.nv + 1
            gq = torch.zeros(B, nq, dtype=torch.float64, device=dev)
            gq[:, 6] = 1.0
            gdq = torch.zeros(B, nvv, dtype=torch.float64, device=dev)
            gqn, gvn, gqs = torch.zeros_like(gq), torch.zeros_like(gdq), torch.zeros_like(gdq)
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
            try:  # secondary section: never allowed to take the headline line down with it
                result["before_path"] = before_path(h, st, dev, B, torch, not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                result["before_path"] = {"error": "%s: %s" % (type(e).__name__, e)}

        if world == 1 and not args.no_sweep and not args.no_compare:
            # batch 1 .. 8192 on the same stream (BASELINE metric: "at batch 1..8192"); instances beyond 1024 repeat the 1024
            # generated ones on their own ticks
            table = []
            hs = capi.Handle(device=local_rank, dtype=cdt, flags=base_flags)
            hs.set_structure(0, st)
            for bsz in SWEEP:
                reps_in = (bsz + B - 1) // B
                di = {k: (v.repeat(reps_in, 1)[:bsz].contiguous()) for k, v in d_in.items()}
                nt = 8 if stream else 1
                dicts = []
                for t in range(nt):
                    d = dict(di)
                    if stream:
                        d["b1"] = b1_ticks[(17 * t) % N_TICKS].repeat(reps_in, 1)[:bsz].contiguous() if bsz > B else b1_ticks[t][:bsz].contiguous()
                    dicts.append(d)
                do = new_out(bsz)
                for t in range(6):
                    hs.solve_batch(0, bsz, dicts[t % nt], do, stream=sp)
                torch.cuda.synchronize()
                nrep = 40 if bsz <= 2048 else 20
                t1 = time.perf_counter()
                for t in range(nrep):
                    hs.solve_batch(0, bsz, dicts[t % nt], do, stream=sp)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / nrep
                table.append({"batch": bsz, "ms": round(dt * 1e3, 5), "qps": round(bsz / dt, 1),
                              "frac_hbm": round(abytes * bsz / dt / 1e9 / HBM_PEAK_GBS, 5)})
            hs.close()
            result["sweep"] = table
            if args.sweep:
                os.makedirs(os.path.dirname(os.path.abspath(args.sweep)), exist_ok=True)
                with open(args.sweep, "w") as fh:
                    json.dump(table, fh, indent=1)

        if world == 1 and not args.no_compare and args.robot == "talos" and not f32 and B == 1024:
            try:  # BASELINE configs 3 and 5, recorded by the same command (bounded: a few seconds each)
                result["other_configs"] = other_configs(local_rank, torch, not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                result["other_configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                result["dense_seam"] = dense_seam(local_rank, st, inputs, not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                result["dense_seam"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                result["franka_single_tick"] = franka_single_tick(local_rank, torch, not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                result["franka_single_tick"] = {"error": "%s: %s" % (type(e).__name__, e)}

        if not args.no_cpu_baseline:
            # (at N > 1 too: rank 0 times the host while the other ranks wait at the final barrier -- a SCALE record then carries the baseline
            #  beside its roofline.)  The oracle is the checker here and the reported CPU baseline -- never the thing shipped.  One call per figure:
            # work items from one counter, per-thread workspace, the clock inside the C driver between the threads' common
            # start line and the last one's end (loop shape of qp_timer_test.cpp:55-63)
            from oracle import oracle
            oracle.build()
            cores, cores_note = usable_cores()
            nsamp = min(B, 256)
            cpu_in = {k: v.copy() for k, v in inputs.items()}
            if stream:
                cpu_in["b1"] = b1_ticks[last_tick].cpu().numpy().astype(np.float64)
            sub = {k: v[:nsamp] for k, v in cpu_in.items()}
            s1, ref = oracle.tick_batch_timed(st, sub, nthreads=1, reps=1)
            single = nsamp / s1
            # one pass on all usable cores sizes the timed sample (the container may grant fewer CPUs than it shows)
            sp1, _ = oracle.tick_batch_timed(st, cpu_in, nthreads=cores, reps=1)
            budget = max(1.0, args.cpu_seconds - s1 - sp1)
            reps = int(max(1, min(4096, budget / max(sp1, 1e-6))))
            sm, _ = oracle.tick_batch_timed(st, cpu_in, nthreads=cores, reps=reps)
            multi = B * reps / sm
            ok = ref["status"] == 0
            xs = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
            native = None
            try:  # SURVEY 8(d) asks for -O3 -march=native: the same C file built that way, timed the same way (the default build is
                # the reproducibility build: -march=x86-64-v3 -ffp-contract=off, bit-identical across hosts)
                sn1, _ = oracle.tick_batch_timed(st, sub, nthreads=1, reps=1, native=True)
                snp, _ = oracle.tick_batch_timed(st, cpu_in, nthreads=cores, reps=1, native=True)
                nreps = int(max(1, min(4096, 0.5 * budget / max(snp, 1e-6))))
                snm, refn = oracle.tick_batch_timed(st, cpu_in, nthreads=cores, reps=nreps, native=True)
                native = {"value": B * nreps / snm, "single_thread": nsamp / sn1, "cores": cores, "flags": oracle.NATIVE_CFLAGS,
                          "max_rel_dx_vs_default_build": float((np.abs(refn["x"][:nsamp] - ref["x"]).max(axis=1) / xs).max())}
            except Exception as e:  # noqa: BLE001
                native = {"error": "%s: %s" % (type(e).__name__, e)}
            result["cpu_baseline"] = {
                "value": multi, "unit": "QP/s", "cores": cores, "kind": "port",
                "sample": "%d passes over the %d QPs of the last timed tick on %d pthreads (%.1f s); single thread %.0f QP/s on %d QPs" %
                          (reps, B, cores, sm, single, nsamp),
                "single_thread": single, "scaling_efficiency": multi / (single * cores), "cpu_model": cpu_model(),
                "cores_note": cores_note, "flags": oracle.DEFAULT_CFLAGS, "native_build": native,
            }
            result["parity"] = {
                "sample": nsamp, "tick": int(last_tick),
                "max_rel_dx": float((np.abs(x_gpu[:nsamp] - ref["x"]).max(axis=1) / xs)[ok].max()),
                "max_abs_dtau": float(np.abs(tau_gpu[:nsamp] - ref["tau"])[ok].max()) if st.na else 0.0,
                "status_equal": bool(np.array_equal(status[:nsamp], ref["status"])),
                "iters_equal_frac": float((iters[:nsamp] == ref["iters"]).mean()),
                # the final active set as eiquadprog numbers it (bit r of wbcqp_outputs.active_mask = one-sided CI row r), its size and the
                # objective, against the oracle's A / iq / f on the same sample
                "active_set_equal_frac": float((mask_gpu[:nsamp] == ref["active_mask"]).all(axis=1)[ok].mean()) if ok.any() else 1.0,
                "n_active_equal_frac": float((nact_gpu[:nsamp] == ref["n_active"])[ok].mean()) if ok.any() else 1.0,
                "max_rel_dobjective": float((np.abs(obj_gpu[:nsamp] - ref["fval"]) / np.maximum(1.0, np.abs(ref["fval"])))[ok].max(initial=0.0)),
            }
            if st.nc:  # raw contact-point forces: six directions per contact are held by the 1e-8 regulariser alone (tests/util.py: bounded at 1e-4) -- reported
                result["parity"]["max_rel_draw_force"] = float((np.abs(x_gpu[:nsamp, st.nv:] - ref["x"][:, st.nv:]).max(axis=1) / xs)[ok].max(initial=0.0))
                result["parity"]["max_rel_ddv"] = float((np.abs(x_gpu[:nsamp, :st.nv] - ref["x"][:, :st.nv]).max(axis=1) / xs)[ok].max(initial=0.0))
        print(json.dumps(result), flush=True)

    if exchange is not None and exchange.get("hung"):
        # a collective that never returned holds this process' stream and its handle: no barrier, no destructors.  The line is out (rank 0);
        # the exit code says that the run was not clean.  Every rank's own watchdog fires on the same collective, so nobody is left waiting.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)
    h.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
