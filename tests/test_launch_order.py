"""Host model of the launch-order packer (inria_wbc_amd/launch_order.py; the device kernel is compared with it in
tests/test_gpu_launch_order.py): it always returns a permutation, it beats plain longest-first under list scheduling on the
iteration histograms the bench and the model-produced rows show, and the dispatch models reproduce the microbenchmark
(tools/ubench/dispatch_order.hip, traces in profiles/r01/dispatch/)."""
import os

import numpy as np
import pytest

from inria_wbc_amd import launch_order as lo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_HIST = [0, 65, 180, 261, 220, 134, 85, 44, 19, 8, 4, 1, 2, 0, 0, 1]  # profiles/bench_r01_v15.json, active_set.iters_hist


def _bench_iters(seed=1):
    it = np.repeat(np.arange(len(BENCH_HIST)), BENCH_HIST)
    it[-1] = 21  # iters_max of that batch (the histogram's last bin is ">= 15")
    np.random.default_rng(seed).shuffle(it)
    return it


@pytest.mark.parametrize("total,resident", [(1024, 256), (512, 256), (2048, 256), (288, 256), (1536, 256), (4096, 2048)])
def test_always_a_permutation(total, resident):
    rng = np.random.default_rng(total)
    for it in (rng.integers(0, 30, total), np.ones(total, int), rng.integers(0, 200, total), np.zeros(total, int),
               np.where(rng.random(total) < 0.4, 1, rng.geometric(0.2, total))):
        cls = np.clip(it, 0, lo.MAX_CLASS)
        lpt = np.argsort(-cls, kind="stable")
        out = lo.pack_order(lpt, it, resident)
        assert sorted(out.tolist()) == list(range(total))


def test_the_hosts_condition():
    assert lo.packs(1024, 256) and lo.packs(2048, 256) and lo.packs(288, 256)
    assert not lo.packs(256, 256)        # one QP per workgroup: nothing to balance
    assert not lo.packs(1000, 256)       # not a multiple of the sub-problem count
    assert not lo.packs(4096, 256)       # more than eight per workgroup: longest-first is already within 2 %
    assert not lo.packs(1024, 256, n_groups=2)


def test_packing_beats_longest_first_on_the_bench_histogram():
    it = _bench_iters()
    cost = 94.0 + 16.8 * it   # tools/cost_dump.py fit of the bench batch, k cycles
    lpt = np.argsort(-it, kind="stable")
    packed = lo.pack_order(lpt, it, 256)
    perfect = cost.sum() / 256
    assert lo.makespan(cost[lpt]) / perfect > 1.12
    assert lo.makespan(cost[packed]) / perfect < 1.07
    # and the queue beats the hardware's dispatcher on the same order
    assert lo.makespan_hw(cost[lpt]) > lo.makespan(cost[lpt])
    assert lo.makespan_hw(cost) > lo.makespan(cost)


def test_dispatch_model_reproduces_the_microbenchmark():
    """profiles/r01/dispatch/trace_*.txt: index, xcc, se, cu, start and end (10 ns ticks) of 1024 CU-filling workgroups.
    The hardware: XCD = i % 8, shader engine fixed by (i / 8) % 4, starts monotone inside an XCD; the model's makespan is
    within 2 % of the measured one."""
    for name in ("lpt", "packed"):
        d = np.loadtxt(os.path.join(ROOT, "profiles", "r01", "dispatch", "trace_%s.txt" % name), dtype=np.int64)
        idx, xcc, se, cu, t0, t1 = d.T
        assert np.array_equal(xcc, idx % 8)
        for x in range(8):
            m = xcc == x
            assert np.all(np.diff(t0[m]) >= -4)  # one tick of the 100 MHz clock's 4-tick granularity
            s = se[m]
            assert all(len(set(s[r::4].tolist())) == 1 for r in range(4))
        spin = (t1 - t0).astype(float)
        assert abs(lo.makespan_hw(spin) - t1.max()) / t1.max() < 0.02
