#!/usr/bin/env python3
"""Every s_barrier of the device code must be preceded, in its own basic block and with no LDS or scalar-memory instruction in
between, by an `s_waitcnt` that carries lgkmcnt(0).

Why this is checked (round 4): __syncthreads() is a workgroup release fence + s_barrier + acquire fence, and the waitcnt pass of
ROCm 7.2's clang drops the fence's `s_waitcnt lgkmcnt(0)` at some loop headers -- the QR loop of the equality phase is one: the
ds_write_b128s of the next reflector were followed by the barrier with no wait, and on MI355X about one QP in 50 000 then read a stale
reflector (tools/determinism_probe.py; the results differ from run to run).  bsync() (wbcqp_prims.hpp) therefore issues the wait itself; this
check is part of inria_wbc_amd/build.py (it reads the assembly of the very compilation that makes the library and refuses to ship a
barrier without the wait); this is its command-line form.

python tools/check_barriers.py file.s"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from inria_wbc_amd.build import check_barriers
    if len(sys.argv) < 2:
        sys.exit("usage: python tools/check_barriers.py file.s   (hipcc ... -save-temps keeps <name>-hip-amdgcn-amd-amdhsa-gfx950.s)")
    total, bad = check_barriers(sys.argv[1])
    print("%d barriers, %d without a preceding s_waitcnt lgkmcnt(0)" % (total, len(bad)))
    for fn, ln in bad[:40]:
        print("  %s  line %d" % (fn, ln))
    sys.exit(1 if bad else 0)
