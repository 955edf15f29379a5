import sys, os, glob, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inria_wbc_amd import capi, structure, synth
from oracle import oracle
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda", 0)
st = structure.talos_structure()
for noise, B in ((0.5, 64), (2.0, 64), (5.0, 32)):
    inp = synth.generate(st, B, synth.SEED_BASE["talos"] + 42, task_noise=noise)
    ref = oracle.tick_batch(st, inp, nthreads=8)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    o = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
             status=torch.full((B,), -9, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    h.solve_batch(0, B, d_in, o, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h.close()
    stt, it, x = o["status"].cpu().numpy(), o["iters"].cpu().numpy(), o["x"].cpu().numpy()
    sc = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
    dx = np.abs(x - ref["x"]).max(axis=1) / sc
    bad = np.nonzero((stt != ref["status"]) | (dx > 1e-8))[0]
    print(os.path.basename(capi.LIB_PATH), "noise", noise, "status!=", int((stt != ref["status"]).sum()), "dx>1e-8", int((dx > 1e-8).sum()), "iters!=", int((it != ref["iters"]).sum()),
          "bad:", [(int(b), int(stt[b]), int(it[b]), int(ref["iters"][b])) for b in bad[:8]])
