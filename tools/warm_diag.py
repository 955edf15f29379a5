import numpy as np, torch, sys
sys.path.insert(0, '.')
from tests.test_gpu_warm import _stream, _solve
from inria_wbc_amd import capi, structure
from oracle import oracle
st = structure.talos_structure(); B = 1024
at = _stream(st, B, 0); dev = torch.device("cuda", 0)
cold = capi.Handle(0, capi.F64); cold.set_structure(0, st)
warm = capi.Handle(0, capi.F64, flags=capi.FLAG_WARM_START); warm.set_structure(0, st)
c0 = _solve(cold, st, at(40), torch.zeros(B, 8, dtype=torch.int32, device=dev))
mask = torch.from_numpy(c0["active_mask"].copy()).to(dev)
for t in range(41, 46):
    inp = at(t)
    w = _solve(warm, st, inp, mask)
    cc = _solve(cold, st, inp, torch.zeros(B, 8, dtype=torch.int32, device=dev))
    ref = oracle.tick_batch(st, inp, nthreads=8)
    scale = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
    e = np.abs(w["x"][:, :st.nv] - ref["x"][:, :st.nv]).max(axis=1) / scale
    bad = np.where(e > 1e-8)[0]
    print("tick", t, "max rel err", e.max(), "n bad", bad.size, "iters warm/cold sum", w["iters"].sum(), cc["iters"].sum(), "max", w["iters"].max(), cc["iters"].max(),
          "status eq", np.array_equal(w["status"], ref["status"]), "nact eq", (w["n_active"] == cc["n_active"]).mean())
    top = np.argsort(-cc["iters"])[:8]
    print("   heaviest cold:", cc["iters"][top].tolist(), "warm:", w["iters"][top].tolist(), "err", ["%.1e" % e[i] for i in top])
    for i in bad[:3]:
        H, g, CE, ce0, CI, ci0 = oracle.assemble(st, inp, i)
        xw = w["x"][i]; xc = cc["x"][i]
        sw = CI @ xw + ci0; sc = CI @ xc + ci0
        fw = 0.5 * xw @ H @ xw + g @ xw; fc = 0.5 * xc @ H @ xc + g @ xc
        print("   qp", i, "err", e[i], "iters", w["iters"][i], cc["iters"][i], "nact", w["n_active"][i], cc["n_active"][i], "min s warm/cold", sw.min(), sc.min(), "f warm-cold", fw - fc, "eq res", np.abs(CE @ xw + ce0).max())
