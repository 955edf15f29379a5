"""K6 / SURVEY 8(e): the one optional exchange step -- `wbcqp_allgather_tau` on a caller-supplied ncclComm_t -- executed on
the GPU: a 1-rank communicator made through the same librccl the process has (inria_wbc_amd.rccl), solver output in,
gathered torques out, both boundary dtypes.  The 1 -> 8 GPU run is the driver's (bench.py --gpus N --allgather)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_allgather_tau_on_solver_output(dtype):
    import torch
    from inria_wbc_amd import capi, rccl, structure, synth
    st = structure.talos_structure()
    B = 96
    dev = torch.device("cuda", 0)
    tdt, ndt, cdt = (torch.float64, np.float64, capi.F64) if dtype == "f64" else (torch.float32, np.float32, capi.F32)
    inputs = synth.generate(st, B, synth.SEED_BASE["talos_squat"], squat=True)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v.astype(ndt))).to(dev) for k, v in inputs.items() if v.size}
    d_out = dict(x=torch.zeros(B, st.n, dtype=tdt, device=dev), tau=torch.zeros(B, st.na, dtype=tdt, device=dev),
                 status=torch.full((B,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    h = capi.Handle(0, cdt)
    h.set_structure(0, st)
    comm = rccl.comm_init_rank(rccl.unique_id(), 1, 0)
    try:
        sp = torch.cuda.current_stream().cuda_stream
        recv = torch.full((B, st.na), float("nan"), dtype=tdt, device=dev)
        h.solve_batch(0, B, d_in, d_out, stream=sp)
        h.allgather_tau(comm, d_out["tau"].data_ptr(), recv.data_ptr(), B * st.na, stream=sp)
        torch.cuda.synchronize()
        assert (d_out["status"] == 0).all()
        assert torch.equal(recv, d_out["tau"])
        assert torch.isfinite(recv).all()
    finally:
        rccl.comm_destroy(comm)
        h.close()


def test_allgather_tau_rejects_null_arguments():
    from inria_wbc_amd import capi, structure
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, structure.franka_structure())
    with pytest.raises(capi.WbcqpError):
        h.allgather_tau(0, 0, 0, 4)
    h.close()
