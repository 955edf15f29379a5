"""Builds libwbcqp.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in CI containers too. The .so lands in
inria_wbc_amd/lib/ (git-ignored, but shipped to the GPU box with the working tree).
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libwbcqp.so")
SOURCES = ["wbcqp_api.hip"]
# every header under csrc/ (globbed: a hand-kept list once missed wbcqp_compact.hpp and wbcqp_dense.hpp, so an edit to the
# default solve kernel did not trigger a rebuild) + the C ABI header
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "wbcqp.h")]
ARCH = "gfx950"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_stamps(verbose: bool = False) -> str:
    """Diagnostic library with in-kernel phase stamps (never the one tests or bench load)."""
    out = os.path.join(LIBDIR, "libwbcqp_stamps.so")
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out  # (built by __graft_entry__.build() here; it travels to the GPU box with the tree)
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-ffp-contract=on",
           "-DWBCQP_STAMPS", *[os.path.join(CSRC, s) for s in SOURCES], "-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def build_poison(verbose: bool = False, force: bool = False) -> str:
    """Diagnostic library whose compact kernel fills its whole LDS block with signalling garbage (a NaN pattern) before the QP is loaded
    (-DWBCQP_POISON_LDS): a read of LDS the kernel never wrote, or wrote from another wave without a barrier in between, shows up as a result
    that differs from the product library's.  tests/test_gpu_layout_variants.py runs the layout branches through it (never the library tests
    or bench load otherwise)."""
    out = os.path.join(LIBDIR, "libwbcqp_poison.so")
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-ffp-contract=on",
           "-DWBCQP_POISON_LDS=0x7ff4dead7ff4beefLL", *[os.path.join(CSRC, s) for s in SOURCES], "-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


HOST = os.path.join(CSRC, "host")


def build_host(verbose: bool = False) -> dict:
    """C++ host facade (mirror of the reference's controller / behavior plugin surface) + its two programs, with g++."""
    build()
    inc = ["-I", os.path.join(HOST, "include"), "-I", os.path.join(_HERE, "..", "include")]
    common = ["g++", "-std=c++17", "-O2", "-Wall", "-Wno-unused-function", "-Wno-return-type", *inc]
    link = ["-L", LIBDIR, "-lwbcqp", "-Wl,-rpath," + LIBDIR]
    lib = os.path.join(LIBDIR, "libinria_wbc_hip.so")
    out = {"lib": lib, "test_facade": os.path.join(LIBDIR, "test_facade"), "qp_timer_test": os.path.join(LIBDIR, "qp_timer_test")}
    # (nothing to do when the three outputs are newer than every source of the facade, the C ABI header and the library)
    deps = [LIB, os.path.join(_HERE, "..", "include", "wbcqp.h"), os.path.abspath(__file__)]
    for root, _, files in os.walk(HOST):
        deps += [os.path.join(root, f) for f in files if f.endswith((".hpp", ".cpp", ".h"))]
    if all(os.path.exists(o) for o in out.values()):
        oldest = min(os.path.getmtime(o) for o in out.values())
        if all(os.path.getmtime(d) <= oldest for d in deps):
            return out
    cmds = [
        common + ["-fPIC", "-shared", os.path.join(HOST, "src", "registry.cpp"), "-o", lib] + link,
        common + [os.path.join(HOST, "tests", "test_facade.cpp"), os.path.join(HOST, "src", "registry.cpp"), "-o", out["test_facade"]] + link,
        common + [os.path.join(HOST, "tests", "qp_timer_test.cpp"), os.path.join(HOST, "src", "registry.cpp"), "-o", out["qp_timer_test"]] + link,
    ]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out


DEVICE_ASM = os.path.join(LIBDIR, "wbcqp_device.s")
_MEM = re.compile(r"^\s*(ds_|s_load|s_buffer_load|s_store|s_atomic|s_dcache|flat_)")  # whatever counts on lgkmcnt (flat: it may)


def check_barriers(path: str):
    """(number of s_barrier, [(function, line)] of those not preceded -- in their own basic block, with no LDS / scalar-memory
    instruction in between -- by an s_waitcnt carrying lgkmcnt(0)).  tools/check_barriers.py is the command-line form."""
    bad, total, fn = [], 0, None
    with open(path) as f:
        lines = f.read().splitlines()
    for i, ln in enumerate(lines):
        s = ln.strip()
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*;\s*@", s)
        if m:
            fn = m.group(1)
        if not s.startswith("s_barrier"):
            continue
        total += 1
        ok = False
        for j in range(i - 1, max(i - 400, -1), -1):
            t = lines[j].strip()
            if not t or t.startswith(";"):
                continue
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                ok = True
                break
            if t.startswith(".LBB") or t.endswith(":") or _MEM.match(t) or t.startswith("s_barrier"):
                break
        if not ok:
            bad.append((fn, i + 1))
    return total, bad


def check_resources(usage: dict) -> None:
    """The refusals of build(), on {kernel: {field: int}} as _resource_usage returns it.  An EMPTY or solve-kernel-free table is itself refused: for a
    whole round the remarks' spelling under -save-temps was not the one the parser knew, the table came back empty and every check below passed
    vacuously (a generic kernel with 96 AGPRs and one workgroup per CU shipped into a measurement pass: profiles/r06/not_kept.txt item 9)."""
    solve = [n for n in usage if "solve_queue_kernel" in n]
    if len(solve) < 8 or any("VGPRs" not in usage[n] or "Occupancy [waves/SIMD]" not in usage[n] for n in solve):
        raise RuntimeError("kernel-resource remarks not understood: %d solve_queue_kernel entries among %d kernels -- the build's register checks would "
                           "pass vacuously; refuse to ship" % (len(solve), len(usage)))
    for name, u in usage.items():
        if ("solve_kernel" in name or "solve_queue_kernel" in name or "terms_kernel" in name or "integrate_kernel" in name) and (u.get("AGPRs", 0) != 0 or u.get("ScratchSize [bytes/lane]", 0) != 0 or u.get("VGPRs Spill", 0) != 0):
            raise RuntimeError("%s: AGPRs %s, scratch %s B/lane, VGPR spills %s -- refuse to ship (see the comment in build.py)" %
                               (name, u.get("AGPRs"), u.get("ScratchSize [bytes/lane]"), u.get("VGPRs Spill")))
        # two workgroups per CU is what the queue kernels are sized and measured for (four waves each, one per SIMD: two waves per SIMD)
        if "solve_queue_kernel" in name and u.get("Occupancy [waves/SIMD]", 0) < 2:
            raise RuntimeError("%s: %s VGPRs, occupancy %s waves per SIMD -- refuse to ship" % (name, u.get("VGPRs"), u.get("Occupancy [waves/SIMD]")))
        # the three-per-CU twin of the compact queue kernel is the one place scratch is admitted: it is compiled for 168 VGPRs and measured
        # faster WITH its spills than the two-per-CU kernel without (tools/occ3_probe.py).  It must really reach three waves per SIMD, keep
        # AGPRs out (the bug above) and its spills bounded.
        if "solve_queue3_kernel" in name and (u.get("AGPRs", 0) != 0 or u.get("Occupancy [waves/SIMD]", 0) < 3 or u.get("ScratchSize [bytes/lane]", 0) > 320):
            raise RuntimeError("%s: AGPRs %s, scratch %s B/lane, occupancy %s -- refuse to ship" %
                               (name, u.get("AGPRs"), u.get("ScratchSize [bytes/lane]"), u.get("Occupancy [waves/SIMD]")))


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    tmpdir = tempfile.mkdtemp(prefix="wbcqp_build_")
    try:
        cmd = [hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}",
               "-fno-gpu-rdc", "-ffp-contract=on", "-Wall", "-Wno-unused-function", "-save-temps=cwd",
               *extra_flags,
               *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        # The kernel-resource remarks are part of the build: a solve kernel that parks live registers in AGPRs or spills to
        # scratch is refused.  Measured reason: with more than 256 live VGPRs the allocator split live ranges into AGPRs and
        # ROCm 7.2's clang placed such copies at the head of a join block BEFORE the `s_or_b64 exec` that restores the lanes,
        # i.e. under a partial EXEC mask -- values came back corrupted in the other lanes (non-deterministic results in the
        # diagnostic build; tools/chk_lib.py, tests/test_gpu_parity.py::test_diagnostic_build_is_a_canary).
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=tmpdir)
        usage = _resource_usage(proc.stdout)
        if proc.returncode != 0:
            sys.stderr.write(proc.stdout)
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        check_resources(usage)
        for line in proc.stdout.splitlines():
            if "remark:" not in line and "kernel-resource-usage" not in line and line.strip() and not line.lstrip().startswith(("|", "^")) \
                    and not line.strip()[:5].strip().isdigit():
                print(line, file=sys.stderr)
        # The device assembly is part of the build too: every workgroup barrier must wait for the wave's own LDS traffic first
        # (tools/check_barriers.py says why; bsync() in wbcqp_prims.hpp issues the wait).
        asm = [f for f in os.listdir(tmpdir) if f.endswith("gfx950.s")]
        if len(asm) != 1:
            raise RuntimeError("no device assembly among the build's temporaries: %s" % sorted(os.listdir(tmpdir)))
        total, bad = check_barriers(os.path.join(tmpdir, asm[0]))
        if bad or total == 0:
            shutil.copy(os.path.join(tmpdir, asm[0]), DEVICE_ASM + ".refused")
            raise RuntimeError("%d of %d s_barrier without a preceding s_waitcnt lgkmcnt(0): %s ... -- refuse to ship (assembly kept as %s.refused)" %
                               (len(bad), total, bad[:4], DEVICE_ASM))
        check_specialisations(LIB + ".tmp")
        os.replace(LIB + ".tmp", LIB)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)  # (-save-temps leaves some 40 MB there)
        if os.path.exists(LIB + ".tmp"):
            os.remove(LIB + ".tmp")
    return LIB


def check_specialisations(lib_path: str) -> None:
    """kSpecDims (csrc/wbcqp_types.hpp) repeats by hand what derive_compact computes for the shipped stacks; a layout change that is not carried over
    would move them to the generic kernel without a word (about 8 % slower, same results).  The library just built is asked for the layout of each
    stack (host code only: no GPU needed) and refused when one does not get its own instantiation; WBCQP_DEBUG_DUMP_STRUCT=1 prints the row to paste."""
    import ctypes
    from . import capi, structure
    if "torch" not in sys.modules:  # (torch's HIP runtime before the library's: capi.load_library says why)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(lib_path)  # (a private handle on the file just built: the process may already hold the previous libwbcqp.so)
    lib.wbcqp_layout_of.argtypes = [ctypes.POINTER(capi.CStructure), ctypes.POINTER(capi.CLayout)]
    for name, want in (("talos", 1), ("icub", 2), ("talos_single_support", 3)):
        sb, L = capi.StructureBuffers(structure.STRUCTURES[name]()), capi.CLayout()
        if lib.wbcqp_layout_of(ctypes.byref(sb.c), ctypes.byref(L)) != 0:
            raise RuntimeError("wbcqp_layout_of refuses the %s stack" % name)
        got = L.specialised
        if got != want:
            raise RuntimeError("%s: wbcqp_layout.specialised = %d, expected %d -- kSpecDims (csrc/wbcqp_types.hpp) no longer equals the layout derive_compact "
                               "computes for this stack; WBCQP_DEBUG_DUMP_STRUCT=1 python -c 'from inria_wbc_amd import capi, structure; "
                               "capi.layout_of(structure.STRUCTURES[\"%s\"]())' prints the row" % (name, got, want, name))


def _resource_usage(text: str) -> dict:
    """{kernel name: {field: int}} from clang's -Rpass-analysis=kernel-resource-usage remarks."""
    out, cur = {}, None
    for line in text.splitlines():
        if "remark:" not in line:
            continue
        # two spellings: "<file>:<line>:<col>: remark: Function Name: ..." (one-step compile) and "remark: <file>:<line>:<col>: Function Name: ..." (the
        # device compile run from a saved temporary, -save-temps, which is how build() compiles) -- take what follows the LAST "<digits>: " or "remark: "
        body = line.split("[-Rpass", 1)[0]
        body = re.split(r"(?:remark:|:\d+:\d+:)\s*", body)[-1].strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            out[cur] = {}
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            try:
                out[cur][k.strip()] = int(v.strip())
            except ValueError:
                pass
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--stamps" in sys.argv:
        print(build_stamps(verbose=True))
    if "--poison" in sys.argv:
        print(build_poison(verbose=True))
    if "--host" in sys.argv:
        print(build_host(verbose=True))
