"""The same records launched again and again give the same bits, whatever the dispatch and whichever kernel (generic or specialised).

Round 4: with the specialised kernels about one Talos QP in 50 000 differed from run to run -- one wave had applied a stale reflector in
the QR of the equality phase, because the compiler had dropped the LDS wait of a __syncthreads() (wbcqp_prims.hpp, bsync();
inria_wbc_amd/build.py refuses a build with such a barrier).  A race of that rate needs many QPs to show: 5 dispatches x 40 launches x
2048 QPs = 400 k here (the faulty build failed this about nine times in ten)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("robot,batch,launches", [("talos", 2048, 40), ("icub", 2048, 12), ("talos_single_support", 1024, 12)])
def test_every_launch_gives_the_same_bits(robot, batch, launches):
    import determinism_probe
    s = determinism_probe.run(batch=batch, launches=launches, robot=robot, verbose=True)
    assert s["deterministic"], s


@pytest.mark.parametrize("robot,batch,launches,full", [("talos", 1024, 10, True), ("talos_torque_cop", 512, 8, False), ("franka", 8192, 8, False),
                                                       ("tiago", 4096, 8, False)])
def test_the_other_kernels_too(robot, batch, launches, full):
    """The full LDS layout (round 1's kernel), a stack with a torque and a cop task (H as one matrix, full layout) and the one-wavefront
    kernel: every workgroup barrier of every kernel is bsync() now, so they are held to the same bar."""
    import determinism_probe
    from inria_wbc_amd import capi
    s = determinism_probe.run(batch=batch, launches=launches, robot=robot, verbose=True, extra_flags=capi.FLAG_FULL_LDS if full else 0)
    assert s["deterministic"], s
