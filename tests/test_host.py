"""CPU tests of the host side: task-stack structures, the C-ABI library surface (loads, exports
every symbol include/wbcqp.h declares, layout without a GPU, fails loudly without a device), reference streams."""
import ctypes
import os
import re

import numpy as np
import pytest

from inria_wbc_amd import structure, synth, trajs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,n,neq,nin,r1,dense", [
    ("franka", 9, 0, 0, 15, 6), ("icub", 62, 18, 66, 83, 39), ("talos", 74, 18, 122, 97, 41),
    ("talos_single_support", 62, 12, 105, 91, 41)])
def test_structure_dimensions_match_survey_appendix_b(name, n, neq, nin, r1, dense):
    st = structure.STRUCTURES[name]()
    assert (st.n, st.neq, st.nin, st.r1, st.n_dense) == (n, neq, nin, r1, dense)
    assert st.nin2 == 2 * nin


def test_talos_stack_constants():
    st = structure.talos_structure()
    w = dict(zip(st.task_names, st.default_weights))
    assert w["lf"] == 1000.0 and w["com"] == 1000.0 and w["posture"] == 1.75 and w["head"] == 1.0
    assert w["forcereg_contact_lfoot"] == 1e-3  # tasks.hpp:23
    B, lb, ub = st.friction()
    assert B.shape == (2, 17, 12) and lb[0, 16] == 5.0 and ub[0, 16] == 1500.0 and (lb[0, :16] == -1e10).all()
    # normal (0,0,1), mu 0.3: first pyramid row = (-t1 - mu n) with t1 = n x e_x = e_y
    assert np.allclose(B[0, 0, :3], [0.0, -1.0, -0.3])
    T = st.force_gen()[0]
    assert np.allclose(T[:3, :3], np.eye(3)) and np.allclose(T[3:, :3], structure._skew([-0.11, -0.069, 0.107]))
    assert st.algorithmic_bytes() == 35184  # SURVEY 8(d) counts 15 weights (35 152 B); this stack carries 19
    assert structure.icub_structure().algorithmic_bytes() == 2 * 11936


def test_library_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "wbcqp.h")).read()
    declared = set(re.findall(r"\b(wbcqp_[a-z_]+)\s*\(", hdr))
    from inria_wbc_amd import capi
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = ctypes.CDLL(built_lib)
    for sym in declared:
        assert hasattr(lib, sym), sym
    # and nothing else: the product library's dynamic symbol table holds the declared entry points only
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("wbcqp_")}
    assert exported == declared, exported ^ declared
    assert lib.wbcqp_version() == 151


def test_layout_without_gpu(built_lib):
    from inria_wbc_amd import capi
    for name, mk in structure.STRUCTURES.items():
        st = mk()
        L = capi.layout_of(st)  # includes the synthetic three-contact stack
        assert (L["n"], L["neq"], L["nin"], L["nin2"], L["r1"]) == (st.n, st.neq, st.nin, st.nin2, st.r1)
        fl = st.field_lengths()
        for k in capi.FIELDS:
            assert L["len_" + k] == fl[k], (name, k)
        assert 0 < L["lds_bytes"] <= 160 * 1024 and L["waves_per_cu"] >= 1
        assert L["algorithmic_bytes"] == st.algorithmic_bytes()
        # the shipped humanoid stacks have an instantiation of the compact kernel of their own (csrc/wbcqp_types.hpp: kSpecDims): its
        # literals must equal what the host derives for the stack -- a layout change that forgets the table shows up here
        assert L["specialised"] == {"talos": 1, "icub": 2, "talos_single_support": 3}.get(name, 0), (name, L["specialised"])


def test_invalid_structures_are_rejected(built_lib):
    from inria_wbc_amd import capi
    st = structure.talos_structure()
    st.dense_row_task = st.dense_row_task.copy(); st.dense_row_task[3] = 99
    with pytest.raises(capi.WbcqpError) as e:
        capi.layout_of(st)
    assert e.value.code == 1
    big = structure.tiago_structure(nv=130)
    with pytest.raises(capi.WbcqpError) as e:
        capi.layout_of(big)
    assert e.value.code == 3


def test_no_cpu_fallback(built_lib):
    """Without a gfx950 device the product path must fail loudly (this container has none)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from inria_wbc_amd import capi
    with pytest.raises(capi.WbcqpError) as e:
        capi.Handle(device=0)
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)


def test_min_jerk_matches_closed_form():
    """trajectory_generator.hpp:23-78 and etc/talos/squat.yaml: 2000 + 2000 samples, -0.2 m in z."""
    pos, vel, acc = trajs.move_com_stream([0.0, 0.0, 0.9], [[0.0, 0.0, -0.2]], "001", 1e-3, 2.0, loop=True)
    assert pos.shape == (4000, 3) and vel.shape == (4000, 3) and acc.shape == (4000, 3)
    assert np.allclose(pos[0], [0, 0, 0.9]) and abs(pos[1999, 2] - 0.7) < 1e-6 and abs(pos[3999, 2] - 0.9) < 1e-6
    s = 0.25
    assert abs(pos[500, 2] - (0.9 - 0.2 * (10 * s ** 3 - 15 * s ** 4 + 6 * s ** 5))) < 1e-12
    # derivatives are consistent with finite differences of the position stream
    fd = (pos[2:2000, 2] - pos[:1998, 2]) / 2e-3
    assert np.abs(fd - vel[1:1999, 2]).max() < 1e-5
    assert (pos[:, :2] == [0.0, 0.0]).all()


def test_synthetic_batches_are_feasible_and_order_independent():
    st = structure.talos_structure()
    a = synth.generate(st, 6, synth.SEED_BASE["talos"])
    b = synth.generate(st, 3, synth.SEED_BASE["talos"], first=3)
    for k in synth.FIELDS:
        assert np.array_equal(a[k][3:], b[k])
    one = synth.generate_one(st, synth.SEED_BASE["talos"])
    nv, nu = st.nv, st.nu
    M = np.zeros((nv, nv)); M[np.tril_indices(nv)] = one["M"]; M = M + M.T - np.diag(np.diag(M))
    assert np.linalg.eigvalsh(M).min() > 0
    dv, f = one["_dv_star"], one["_f_star"]
    Ac = one["Ac"].reshape(st.nc, 6, nv); T = st.force_gen()
    Jc = np.concatenate([T[c].T @ Ac[c] for c in range(st.nc)], 0)
    assert np.abs(M[:nu] @ dv - Jc[:, :nu].T @ f + one["h"][:nu]).max() < 1e-9
    tau = M[nu:] @ dv + one["h"][nu:] - Jc[:, nu:].T @ f
    assert (tau > one["tlb"]).all() and (tau < one["tub"]).all()
    B, lb, ub = st.friction()
    for c in range(st.nc):
        v = B[c] @ f[12 * c:12 * c + 12]
        assert (v > lb[c]).all() and (v < ub[c]).all()
    # squat stream (BASELINE config 4): instance i takes tick i mod 4000 of etc/talos/squat.yaml
    sq = synth.generate(st, 2, synth.SEED_BASE["talos_squat"], first=1000, squat=True)
    pl = synth.generate(st, 2, synth.SEED_BASE["talos_squat"], first=1000, squat=False)
    rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    d = sq["b1"][:, rows] - pl["b1"][:, rows]
    assert np.allclose(d[0], synth.squat_com_rhs(st, 1000, 30.0)) and abs(d[0, 2]) > 0 and d[0, 0] == 0


def test_check_model_on_the_host(built_lib):
    """wbcqp_check_model: the validation wbcqp_set_model applies, without a device -- the shipped stacks pass and size their LDS,
    broken tables are refused with a message."""
    import copy
    from inria_wbc_amd import capi, structure
    from inria_wbc_amd import model as mdl
    for m, st, stack in ((mdl.talos_like(), structure.talos_structure(), mdl.talos_stack()), (mdl.icub_like(), structure.icub_structure(), mdl.icub_stack()),
                         (mdl.franka_like(), structure.franka_structure(), mdl.franka_stack())):
        lds = capi.check_model(st, m, mdl.build_taskmap(m, st, stack))
        assert 1024 < lds < 40 * 1024, lds  # four workgroups per CU need less than 40 KB each
    m, st = mdl.talos_like(), structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    cases = [
        (lambda mm, tt: setattr(tt.blocks[3], "frame", m.nframe + 5), "does not exist"),
        (lambda mm, tt: setattr(tt, "dt", 0.0), "bad taskmap"),
        (lambda mm, tt: tt.blocks.pop(0), "n_dense"),
        (lambda mm, tt: mm.jtype.__setitem__(5, 9), "joint type"),
        (lambda mm, tt: mm.parent.__setitem__(20, 3), "depth-first"),
        (lambda mm, tt: mm.frame_body.__setitem__(2, 99), "does not exist"),
        (lambda mm, tt: setattr(tt, "posture_ref", tt.nref), "posture reference"),
    ]
    for mutate, text in cases:
        mm, tt = copy.deepcopy(m), copy.deepcopy(tm)
        mutate(mm, tt)
        with pytest.raises(capi.WbcqpError) as e:
            capi.check_model(st, mm, tt)
        assert text in str(e.value), (text, str(e.value))
    with pytest.raises(capi.WbcqpError):
        capi.check_model(structure.icub_structure(), m, tm)  # another robot's structure


def test_bench_accepts_the_drivers_command_line_and_counts_usable_cores(monkeypatch):
    """bench.py's host-side pieces that need no GPU: the driver's flag set parses to the documented defaults, and the CPU
    baseline is sized by what the container grants (affinity mask / cgroup quota), not by os.cpu_count() alone."""
    import importlib
    import sys as _sys
    _sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.batch, a.robot, a.dtype) == (1, 20, 5, 1024, "talos", "f64")
    assert not a.replay and not a.headline_only and a.backend == "nccl"
    n, note = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and "os.cpu_count()" in note
    assert bench.ALGORITHMIC_BYTES["talos"] == 35152


def test_the_build_refuses_a_barrier_without_the_lds_wait(tmp_path):
    """inria_wbc_amd/build.py reads the device assembly of the compilation that makes the library: an s_barrier that is not preceded, in
    its basic block and before any LDS instruction, by `s_waitcnt ... lgkmcnt(0)` is refused (round 4's race: tools/check_barriers.py)."""
    from inria_wbc_amd.build import check_barriers
    good = """
_ZN5wbcqp1kEv:                          ; @_ZN5wbcqp1kEv
	ds_write_b128 v1, v[2:5]
	s_waitcnt vmcnt(3) lgkmcnt(0)
	v_mov_b32_e32 v0, s1
	s_barrier
	ds_read_b128 v[2:5], v1
.LBB0_1:
	s_waitcnt lgkmcnt(0)
	s_barrier
"""
    bad = """
_ZN5wbcqp1kEv:                          ; @_ZN5wbcqp1kEv
	s_waitcnt lgkmcnt(0)
	ds_write_b128 v1, v[2:5]
	s_barrier
	s_waitcnt lgkmcnt(0)
.LBB0_1:                                ; the wait is in the predecessor: some other path may reach the label without it
	v_mov_b32_e32 v0, s1
	s_barrier
	s_waitcnt lgkmcnt(1)
	s_barrier
"""
    p = tmp_path / "k.s"
    p.write_text(good)
    assert check_barriers(str(p)) == (2, [])
    p.write_text(bad)
    total, flagged = check_barriers(str(p))
    assert total == 3 and [ln for _, ln in flagged] == [5, 9, 11] and flagged[0][0] == "_ZN5wbcqp1kEv"


def _remarks(spelling, kernels):
    """clang's -Rpass-analysis=kernel-resource-usage output in either of the two spellings ROCm 7.2 emits (one-step compile / the device
    compile run from a saved temporary, which is what build() does)."""
    out = []
    for name, fields in kernels:
        for k, v in [("Function Name", name)] + list(fields.items()):
            body = "%s: %s" % (k, v) if k == "Function Name" else "    %s: %s" % (k, v)
            if spelling == "one-step":
                out += ["/x/wbcqp_device.hpp:1014:1: remark: %s [-Rpass-analysis=kernel-resource-usage]" % body, " 1014 | {", "      | ^"]
            else:
                out.append("remark: /x/wbcqp_device.hpp:1014:0: %s [-Rpass-analysis=kernel-resource-usage]" % body)
    return "\n".join(out)


@pytest.mark.parametrize("spelling", ["one-step", "save-temps"])
def test_the_build_reads_the_register_remarks_in_both_spellings_and_refuses_an_empty_table(spelling):
    """Round 6: under -save-temps the remarks come as `remark: file:line:col: Function Name: ...`; the parser only knew `file:line:col: remark:
    Function Name: ...`, returned an empty table and every register check passed vacuously -- a generic kernel with 96 AGPRs at one workgroup
    per CU went into a measurement pass.  Both spellings are parsed now and a table without the solve kernels is itself a refusal."""
    from inria_wbc_amd.build import _resource_usage, check_resources
    ok = {"TotalSGPRs": 106, "VGPRs": 238, "AGPRs": 0, "ScratchSize [bytes/lane]": 0, "Occupancy [waves/SIMD]": 2, "VGPRs Spill": 0, "LDS Size [bytes/block]": 16}
    names = ["_ZN5wbcqp18solve_queue_kernelIdLb1ELi%dEEEvNS_10GroupTableIT_EEPii" % i for i in range(8)]
    good = [(n, ok) for n in names] + [("_ZN5wbcqp19solve_queue3_kernelIdLi2EEEvNS_10GroupTableIT_EEPii", dict(ok, VGPRs=168, **{"ScratchSize [bytes/lane]": 72, "Occupancy [waves/SIMD]": 3}))]
    u = _resource_usage(_remarks(spelling, good))
    assert len(u) == 9 and u[names[0]]["VGPRs"] == 238 and u[names[0]]["Occupancy [waves/SIMD]"] == 2
    check_resources(u)
    for bad, what in ((dict(ok, VGPRs=256, AGPRs=96, **{"Occupancy [waves/SIMD]": 1}), "AGPRs 96"), (dict(ok, **{"ScratchSize [bytes/lane]": 8}), "scratch 8"),
                      (dict(ok, VGPRs=300, **{"Occupancy [waves/SIMD]": 1}), "occupancy 1")):
        with pytest.raises(RuntimeError, match=what):
            check_resources(_resource_usage(_remarks(spelling, good[:3] + [(names[3], bad)] + good[4:])))
    with pytest.raises(RuntimeError, match="occupancy 2"):  # the three-per-CU twin that does not reach three
        check_resources(_resource_usage(_remarks(spelling, good[:8] + [(good[8][0], dict(good[8][1], **{"Occupancy [waves/SIMD]": 2}))])))
    with pytest.raises(RuntimeError, match="not understood"):
        check_resources({})
    with pytest.raises(RuntimeError, match="not understood"):
        check_resources(_resource_usage(_remarks(spelling, good).replace("Function Name", "Kernel Name")))


def test_the_library_on_disk_holds_no_agprs_and_no_scratch_in_its_solve_kernels(built_lib):
    """What the build refuses, read back from the code object of the library the tests and the bench load (tools/kernel_regs.py): a library put in
    place by another route -- a variant from tools/variants.sh copied over -- does not pass through build()'s checks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(os.path.dirname(__file__), "..", "tools", "kernel_regs.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not os.path.exists(kr.LLVM + "/llvm-readelf"):
        pytest.skip("no llvm-readelf")
    rows = kr.kernels(built_lib)
    queue = [r for r in rows if "solve_queue_kernel" in r[4]]
    assert len(queue) >= 14
    for v, a, scratch, _, name in rows:
        if "solve_queue_kernel" in name or "solve_kernel" in name:
            assert (a, scratch) == (0, 0) and v <= 256, (name, v, a, scratch)  # two workgroups per CU: 512 / 256
        if "solve_queue3_kernel" in name:
            assert a == 0 and v <= 168 and scratch <= 320, (name, v, a, scratch)  # three: 512 / 168


def test_every_workgroup_barrier_in_the_sources_is_bsync():
    """Source-level half of the barrier rule (the build checks the assembly): __syncthreads() appears once, inside bsync(), which issues the
    LDS wait the compiler may drop (csrc/wbcqp_prims.hpp)."""
    import re
    csrc = os.path.join(ROOT, "inria_wbc_amd", "csrc")
    hits = []
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hpp", ".hip")):
            continue
        for i, ln in enumerate(open(os.path.join(csrc, fn)).read().splitlines(), 1):
            code = ln.split("//")[0]
            if re.search(r"__syncthreads\s*\(|__builtin_amdgcn_s_barrier\s*\(", code):
                hits.append((fn, i))
    assert len(hits) == 1 and hits[0][0] == "wbcqp_prims.hpp", hits


def test_layout_reports_three_only_where_three_fit():
    """wbcqp_layout.waves_per_cu: three for the compact layout when three workgroups fit a CU's LDS (solve_queue3_kernel, tests/test_gpu_three_per_cu.py), else two."""
    from inria_wbc_amd import capi, structure
    one_foot = capi.layout_of(structure.icub_structure(single_support=True))
    assert one_foot["n"] == 50 and one_foot["neq"] == 12 and one_foot["waves_per_cu"] == 3 and 3 * one_foot["lds_bytes"] <= 160 * 1024
    two_feet = capi.layout_of(structure.icub_structure())  # BASELINE config 3's stack: 52.8 KB since its rows of J are n long and the friction table lives in them
    assert two_feet["n"] == 62 and two_feet["waves_per_cu"] == 3 and two_feet["specialised"] == 2 and 3 * two_feet["lds_bytes"] <= 160 * 1024
    # Talos on one foot fits since its layout's last diet (54 480 B) but stays at two: with actuation bounds the three-per-CU kernel measured SLOWER (its 38
    # registers of actuation rows go to scratch: csrc/wbcqp_api.hip, kThree)
    one_foot_talos = capi.layout_of(structure.talos_structure(single_support=True))
    assert one_foot_talos["waves_per_cu"] == 2 and one_foot_talos["specialised"] == 3 and one_foot_talos["lds_bytes"] <= 54592
    assert capi.layout_of(structure.talos_structure())["waves_per_cu"] == 2
    # the sizes DESIGN section 4 quotes (a layout change shows here first; kSpecDims and build.check_specialisations follow it)
    sizes = {name: capi.layout_of(structure.STRUCTURES[name]())["lds_bytes"] for name in ("talos", "icub", "talos_single_support", "icub_single_support")}
    assert sizes == {"talos": 71616, "icub": 52832, "talos_single_support": 54480, "icub_single_support": 46784}, sizes
    assert capi.layout_of(structure.franka_structure())["waves_per_cu"] == 2  # (below the queue's size: hardware dispatch of solve_kernel)
