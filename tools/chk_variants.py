#!/usr/bin/env python3
"""Solves the compact layout's branch stacks (tests/test_gpu_layout_variants.py: CASES, among them the friction table in the R region) and the
shipped humanoid stacks with the named build of the library -- libwbcqp.so or the diagnostic libwbcqp_poison.so (LDS pre-filled with a NaN
pattern, inria_wbc_amd/build.py:build_poison) -- and saves every output.  One process per library: two builds of the same symbols must not
share a process.      python tools/chk_variants.py <library file name> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from inria_wbc_amd import capi, structure, synth
    from tests.test_gpu_layout_variants import CASES, _stack
    from tests.util import device_outputs, host_outputs
    which, out_path = sys.argv[1], sys.argv[2]
    capi.LIB_PATH = os.path.join(ROOT, "inria_wbc_amd", "lib", which)
    capi.load_library(capi.LIB_PATH)
    dev = torch.device("cuda", 0)
    stacks = [(name, _stack(name, nv, na, nc, act), 77_000 + 31 * nv + nc, dict(task_noise=1.5, p_act=0.3, p_bnd=0.2)) for name, nv, na, nc, act in CASES]
    stacks += [(name, structure.STRUCTURES[name](), synth.SEED_BASE[name] + 99, dict(task_noise=2.0)) for name in ("talos", "icub", "talos_single_support")]
    saved = {}
    for name, st, seed, kw in stacks:
        B = 256
        inputs = synth.generate(st, B, seed, **kw)
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
        for tag, flags in (("queue", 0), ("hw", capi.FLAG_HW_DISPATCH), ("generic", capi.FLAG_GENERIC_KERNEL)):
            h = capi.Handle(0, capi.F64, flags=flags)
            h.set_structure(0, st)
            o = device_outputs(B, st, dev)
            for _ in range(2):
                h.solve_batch(0, B, d_in, o, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            h.close()
            for k, v in host_outputs(o, st).items():
                saved["%s/%s/%s" % (name, tag, k)] = v
        print(which, name, "iters max", int(saved["%s/queue/iters" % name].max()), "status != 0:", int((saved["%s/queue/status" % name] != 0).sum()))
    np.savez(out_path, **saved)


if __name__ == "__main__":
    main()
