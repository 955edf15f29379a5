import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from inria_wbc_amd import capi, structure, synth
from oracle import oracle
dev = torch.device("cuda", 0)
st = structure.talos_structure()
B = 64
inp = synth.generate(st, B, synth.SEED_BASE["talos"] + 5, task_noise=2.0)
ref = oracle.tick_batch(st, inp, nthreads=8)
d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
res = {}
for name, flags in (("compact", 0), ("full", capi.FLAG_FULL_LDS), ("generic", capi.FLAG_GENERIC_KERNEL if hasattr(capi,'FLAG_GENERIC_KERNEL') else 0)):
    h = capi.Handle(0, capi.F64, flags=flags)
    h.set_structure(0, st)
    o = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
             status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev),
             active_mask=torch.zeros(B, 8, dtype=torch.int32, device=dev))
    h.solve_batch(0, B, d_in, o, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    res[name] = {k: v.cpu().numpy() for k, v in o.items()}
    h.close()
sc = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
for name in res:
    dx = np.abs(res[name]["x"] - ref["x"]).max(axis=1) / sc
    di = res[name]["iters"] - ref["iters"]
    print(name, "iters != oracle:", int((di != 0).sum()), "max rel dx", dx.max(), "n(dx>1e-8)", int((dx > 1e-8).sum()), "status", np.unique(res[name]["status"]))
    print("   iters diff:", di[di != 0].tolist(), "at", np.nonzero(di)[0].tolist())
    print("   dx of those:", dx[di != 0].tolist())
same = (res["compact"]["active_mask"] == res["full"]["active_mask"]).all(axis=1)
print("mask differs at", np.nonzero(~same)[0].tolist())
dxcf = np.abs(res["compact"]["x"] - res["full"]["x"]).max(axis=1) / sc
print("compact vs full dx at those:", dxcf[~same].tolist())
print("oracle iters:", ref["iters"].tolist())
