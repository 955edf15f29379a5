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


def _worker_squat(rank, world, port, B, out_dir):
    """BASELINE config 4 in small: Talos squat stream, contiguous shards (inria_wbc_amd.shard), all-gather of tau."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inria_wbc_amd import shard, structure, synth
    from oracle import oracle
    st = structure.talos_structure()
    begin, end = shard.contiguous_shards(world * B, world)[rank]
    mine = synth.generate(st, end - begin, synth.SEED_BASE["talos_squat"], first=begin, squat=True)
    out = oracle.tick_batch(st, mine)
    tau = torch.from_numpy(out["tau"].copy())
    gathered = torch.zeros(world * B, st.na, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, tau)
    np.save(os.path.join(out_dir, "squat_tau_%d.npy" % rank), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_config4_squat_stream_shards_and_gathers(tmp_path, oracle_mod):
    from inria_wbc_amd import structure, synth
    world, B = 2, 4
    mp.spawn(_worker_squat, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    st = structure.talos_structure()
    full = synth.generate(st, world * B, synth.SEED_BASE["talos_squat"], squat=True)
    ref = oracle_mod.tick_batch(st, full)["tau"]
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "squat_tau_%d.npy" % r)), ref), r


def test_ragged_shards_balance_cost_and_cover_every_qp():
    """a ragged batch is cut by cumulative n^3 cost, not by count."""
    from inria_wbc_amd import shard
    groups = [(9, 3000), (12, 1000), (62, 700), (74, 2000), (62, 492)]  # Franka, Tiago, iCub, Talos, Talos single support
    for world in (1, 2, 3, 4, 8):
        shards = shard.ragged_shards(groups, world)
        assert len(shards) == world
        seen = {gi: [] for gi in range(len(groups))}
        for pieces in shards:
            last = (-1, 0)
            for gi, b, e in pieces:
                assert 0 <= b < e <= groups[gi][1]
                assert (gi, b) >= last  # contiguous in the concatenated batch index
                last = (gi, e)
                seen[gi].append((b, e))
        for gi, (_, cnt) in enumerate(groups):
            iv = sorted(seen[gi])
            assert iv[0][0] == 0 and iv[-1][1] == cnt and all(iv[i][1] == iv[i + 1][0] for i in range(len(iv) - 1))
        cost = shard.shard_costs(groups, shards)
        mean = sum(cost) / world
        assert max(abs(c - mean) for c in cost) <= 74.0 ** 3  # within one QP of the largest structure
    # by count the same batch would be far off: the first half holds the small robots
    by_count = shard.shard_costs(groups, [[(0, 0, 3000), (1, 0, 596)], [(1, 596, 1000), (2, 0, 700), (3, 0, 2000), (4, 0, 492)]])
    assert by_count[1] > 50 * by_count[0]
    assert shard.contiguous_shards(8192, 8)[3] == (3072, 4096)


def test_the_bench_lines_8_rank_ragged_partition_replayed():
    """BASELINE config 5 over 8 ranks, exactly the partition bench.py's line reports as `other_configs.config5_ragged_b8192.shards_8_ranks`
    (tools/ragged_bench.py: the same seed draws the mix, shard.ragged_shards cuts it): every QP on exactly one rank, every rank's n^3 cost within one
    Talos QP of the mean, and the counts the line prints are the counts of the plan."""
    from inria_wbc_amd import shard, structure
    from tools import ragged_bench
    kinds = np.random.default_rng(5_000_000).integers(0, len(ragged_bench.NAMES), size=8192)
    groups = [(structure.STRUCTURES[name]().n, int((kinds == slot).sum())) for slot, name in enumerate(ragged_bench.NAMES)]
    groups = [g for g in groups if g[1]]
    assert sum(c for _, c in groups) == 8192 and len(groups) == len(ragged_bench.NAMES)
    plan = shard.ragged_shards(groups, 8)
    cost = shard.shard_costs(groups, plan)
    mean = sum(cost) / 8
    nmax = max(n for n, _ in groups)
    assert max(abs(c - mean) for c in cost) <= float(nmax) ** 3, (cost, mean)
    per_rank = [sum(e - b for _, b, e in pieces) for pieces in plan]
    assert sum(per_rank) == 8192 and min(per_rank) > 0
    covered = {gi: 0 for gi in range(len(groups))}
    for pieces in plan:
        for gi, b, e in pieces:
            assert b == covered[gi]  # (contiguous, in order, no QP twice)
            covered[gi] = e
    assert all(covered[gi] == cnt for gi, (_, cnt) in enumerate(groups))
    # by count the ranks would be far apart: the small robots cost a thousandth of a humanoid
    assert max(per_rank) > 2 * min(per_rank)
    share = [c / sum(cost) for c in cost]
    assert max(share) - min(share) < 0.01


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def test_bench_gpus_n_starts_its_own_ranks_as_a_child(tmp_path):
    """`python bench.py --gpus N` exactly as typed (no launcher, no WORLD_SIZE): the parent never touches the GPU and starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>`
    as a child -- the driver's own command.  On this GPU-less box the ranks get as far as the first GPU call and stop with the
    product's own message (no CPU path to fall back to); the parent hands the child's exit code on."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.child_command(8, 29512, argv)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29512"]
    assert cmd[10] == os.path.join(ROOT, "bench.py") and cmd[11:] == argv
    if torch.cuda.is_available():
        return  # the GPU box runs the two ranks for real: tests/test_gpu_bench.py
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--headline-only"],
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "--nproc-per-node 2" in out and "torch.distributed.run" in out, out[-2000:]
    # (the launcher ends the other rank as soon as the first one fails: one or two copies of the message)
    assert out.count("bench.py needs an MI355X: the product path has no CPU fallback") >= 1, out[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no result line from a run that failed


def test_bench_refuses_a_rank_count_mismatch_under_a_launcher():
    """Under a launcher whose rank count differs from --gpus the bench stops with the way to launch it."""
    import subprocess
    env = _clean_env()
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--nproc-per-node 2" in (r.stdout + r.stderr) and "started 3 ranks" in (r.stdout + r.stderr)
