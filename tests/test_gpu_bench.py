"""bench.py on the GPU box, the way the driver types it: `python bench.py --gpus N ...` with no launcher around it starts its own
ranks (a child `torch.distributed.run`), and the N = 1 line is what the plain run prints.  Two ranks share cuda:0 here (gloo,
--single-device): what is under test is the launch path and the N > 1 code path of the bench, RCCL needs one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_2_typed_without_a_launcher_prints_one_line_with_n_gpus_2(built_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                        "--steps", "6", "--warmup", "2", "--headline-only"], capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "batch-shard x2" and d["config"]["batch_per_gpu"] == 1024
    # whole-job aggregate: both ranks' QPs over the slower rank's time
    assert abs(d["value"] - 2 * 1024 * 6 / (d["ms_per_step"] * 6e-3)) <= 1e-6 * d["value"]
    assert d["active_set"]["status_optimal"] == 1024
    assert "torch.distributed.run" in r.stderr  # the parent says what it starts


def test_the_n_gpus_2_line_carries_the_cpu_baseline_and_the_parity_sample(built_lib):
    """Round 4's line had `cpu_baseline` and `parity` at N = 1 only: a SCALE record would have carried a roofline with no baseline beside it.
    Rank 0 now times the host (and checks its sample against the oracle) while the other ranks wait at the final barrier."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                        "--steps", "6", "--warmup", "2", "--no-compare", "--no-sweep", "--cpu-seconds", "3"],
                       capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["parity"]["status_equal"] and d["parity"]["max_rel_dx"] <= 1e-8 and d["parity"]["sample"] >= 64
    assert d["roofline"]["frac"] > 0


def test_allgather_asked_without_rccl_says_why_it_was_skipped(built_lib):
    """`--allgather` on a run whose ranks cannot form an RCCL communicator (gloo, both ranks on cuda:0): the line must SAY that the exchange leg was
    skipped and why (`allgather_tau.skipped_reason`) instead of omitting the key -- a SCALE record without exchange figures is then explained by its own line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--allgather",
                        "--steps", "4", "--warmup", "2", "--headline-only"], capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["allgather_tau"] is False
    assert "skipped_reason" in d["allgather_tau"] and "one GPU per rank" in d["allgather_tau"]["skipped_reason"]


def test_n1_line_has_the_contract_keys(built_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--no-sweep", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["parity"]["status_equal"] and d["parity"]["max_rel_dx"] <= 1e-8
    # identical active set -- the last timed launch's active_mask / n_active / objective against the oracle's A / iq / f on the sample
    assert d["parity"]["active_set_equal_frac"] >= d["parity"]["iters_equal_frac"] - 1e-12 and d["parity"]["active_set_equal_frac"] >= 0.95
    assert d["parity"]["n_active_equal_frac"] >= 0.95 and d["parity"]["max_rel_dobjective"] <= 1e-6
    # no leg of the line may have failed quietly: a leg that raises leaves {"error": ...} in its place (round 4: `before_path` did, for
    # three profile passes, over an empty record field)
    def errors(node, path=""):
        found = []
        if isinstance(node, dict):
            for k, v in node.items():
                if k == "error":
                    found.append((path, v))
                found += errors(v, path + "/" + k)
        elif isinstance(node, list):
            for i, v in enumerate(node):
                found += errors(v, "%s[%d]" % (path, i))
        return found
    assert errors(d) == []
    assert d["roofline"]["kernel"] == "wbcqp::solve_queue_kernel<double, true, 1>"
    bp = d["before_path"]
    assert bp["parity"] and max(bp["parity"].values()) <= 1e-9 and bp["whole_tick"]["status_optimal"] == 1024
