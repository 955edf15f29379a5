#!/usr/bin/env python3
"""One ragged launch against one launch per structure (round 5): iCub + Talos + Talos on one foot, 1638 QPs each (BASELINE config 5's compact share), through
wbcqp_solve_ragged (one launch of the generic kernel at the largest group's LDS size) and as one wbcqp_solve_batch per group (each its own instantiation
and residency: iCub three per CU) on one stream and on three.  Measured: single launch 0.696 ms; three launches on one stream 0.790 ms (three tails); on
three free-running streams 0.641 ms -- but that figure pipelines consecutive calls into each other; with the fork and the join a library call needs
(built into wbcqp_solve_ragged, measured, reverted) it is 0.759 ms: a persistent queue kernel holds every workgroup slot of the chip until its queue
is empty, so kernels on other streams start when it ends, not beside it.  Not kept; recorded in profiles/r05/not_kept.txt.
    python tools/ragged_split_probe.py"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from inria_wbc_amd import capi, structure, synth
dev = torch.device("cuda", 0)
sp = torch.cuda.current_stream().cuda_stream
NAMES = ["icub", "talos", "talos_single_support"]
h = capi.Handle(0, capi.F64)
groups = []
for slot, name in enumerate(NAMES):
    st = structure.STRUCTURES[name]()
    cnt = 1638
    inp = synth.generate(st, 256, 5_000_000 + 10_000 * slot, task_noise=1.0)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (7, 1))[:cnt])).to(dev) for k, v in inp.items() if v.size}
    d_out = dict(x=torch.zeros(cnt, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(cnt, st.na, dtype=torch.float64, device=dev),
                 status=torch.full((cnt,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(cnt, dtype=torch.int32, device=dev))
    h.set_structure(slot, st)
    groups.append((slot, cnt, d_in, d_out))
def t(fn, n=30):
    for _ in range(6): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
together = t(lambda: h.solve_ragged(groups, stream=sp))
x_t = [g[3]["x"].clone() for g in groups]
def split():
    for (slot, cnt, d_in, d_out) in groups:
        h.solve_batch(slot, cnt, d_in, d_out, stream=sp)
apart = t(split)
same = all(torch.equal(a, g[3]["x"]) for a, g in zip(x_t, groups))
s2 = torch.cuda.Stream(); s3 = torch.cuda.Stream()
streams = [sp, s2.cuda_stream, s3.cuda_stream]
def split_streams():
    for (slot, cnt, d_in, d_out), s in zip(groups, streams):
        h.solve_batch(slot, cnt, d_in, d_out, stream=s)
apart3 = t(split_streams)
print("wbcqp_solve_ragged (one launch) %.3f ms; one launch per group: on one stream %.3f ms, on three free-running streams %.3f ms; same bits %s" % (together, apart, apart3, same))
