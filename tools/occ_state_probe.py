"""Which HIP runtime answers the library's occupancy question?  libwbcqp.so links libamdhip64.so.7; a process that imports torch FIRST shares torch's bundled
runtime, one that loads the library first gets /opt/rocm's; either ALONE answers 2 (Talos) and 3 (iCub).  The bad case is both: the library first, torch after it
(--torch-after: the CDLL is made here by hand, past capi.load_library's own `import torch`) -- the first runtime then answers 1.  WBCQP_DEBUG_LAUNCH=1 prints the
occupancy each launch was sized with.      WBCQP_DEBUG_LAUNCH=1 python tools/occ_state_probe.py [--torch-first | --torch-after]"""
import sys
sys.path.insert(0, "/root/repo")
if "--torch-first" in sys.argv:
    import torch  # noqa: F401
import numpy as np
if "--torch-after" in sys.argv:
    import ctypes
    import os
    ctypes.CDLL(os.path.join("/root/repo", "inria_wbc_amd", "lib", "libwbcqp.so"), mode=ctypes.RTLD_GLOBAL)  # /opt/rocm's runtime comes with it
    import torch  # noqa: F401
    torch.zeros(4, device="cuda")
from inria_wbc_amd import capi, structure, synth
for name in ("talos", "icub"):
    st = structure.STRUCTURES[name]()
    print(name, "layout:", {k: capi.layout_of(st)[k] for k in ("lds_bytes", "waves_per_cu", "specialised")}, flush=True)
    inp = synth.generate(st, 1024, 5)
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    out = h.solve_batch_host(0, inp)
    print(name, "optimal", int((out["status"] == 0).sum()), flush=True)
    h.close()
    if "torch" in sys.modules:  # device-resident throughput at B = 8192 in this process (does the hardware place what the launch was sized for?)
        import time
        import torch
        dev = torch.device("cuda", 0)
        B = 8192
        big = synth.generate(st, 1024, 6)
        d_in = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (8, 1)))).to(dev) for k, v in big.items() if v.size}
        o = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                 status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
        h = capi.Handle(0, capi.F64)
        h.set_structure(0, st)
        sp = torch.cuda.current_stream().cuda_stream
        for _ in range(4):
            h.solve_batch(0, B, d_in, o, stream=sp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            h.solve_batch(0, B, d_in, o, stream=sp)
        torch.cuda.synchronize()
        print(name, "B = 8192: %.2f M QP/s" % (B * 10 / (time.perf_counter() - t0) / 1e6), flush=True)
        h.close()
import ctypes
for lib in ("libamdhip64.so.7", "libamdhip64.so"):
    pass
print(open("/proc/self/maps").read().count("libamdhip64"), [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:3])
