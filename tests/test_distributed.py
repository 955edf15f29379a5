"""world_size-2 gloo test of the N > 1 path: contiguous batch shards per rank (no collective on the solve path)
and the all-gather of joint torques -- on CPU the per-shard solve is stood in for by the oracle, because what is
under test here is the sharding + exchange logic bench.py uses, not the kernel."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inria_wbc_amd import structure, synth
    from oracle import oracle
    st = structure.icub_structure()
    shard = synth.generate(st, B, synth.SEED_BASE["icub"] + 11, first=rank * B)  # rank r owns [r*B, (r+1)*B)
    out = oracle.tick_batch(st, shard)
    tau = torch.from_numpy(out["tau"].copy())
    gathered = torch.zeros(world * B, st.na, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, tau)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction bench.py does
    assert t.item() == world
    np.save(os.path.join(out_dir, "tau_%d.npy" % rank), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_allgather_equals_single_rank(tmp_path, oracle_mod):
    from inria_wbc_amd import structure, synth
    world, B = 2, 6
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    st = structure.icub_structure()
    full = synth.generate(st, world * B, synth.SEED_BASE["icub"] + 11)
    ref = oracle_mod.tick_batch(st, full)["tau"]
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "tau_%d.npy" % r))
        assert np.array_equal(got, ref), r
