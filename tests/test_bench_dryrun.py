"""CPU dry run of `bench.py --gpus N` exactly as the driver may launch it -- a plain `python bench.py --gpus 2`, no torchrun
environment: the script must spawn its own ranks (it raised SystemExit), run the shared step driver
(`ssdn.hip.dp.exchange_step`, bucketed exchange) with barrier + max-over-ranks timing, and print ONE JSON line.  With
SSDN_BENCH_STUB=1 a stub stands where the HIP engine stands and the ranks talk gloo; without a GPU and without the stub the same
launch must END in one JSON {"error": ...} line, not in a traceback without a result."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_bench_gpus2_self_spawns_and_prints_one_line():
    p, lines = _run({"SSDN_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["stub"] is True
    assert r["value"] > 0 and r["unit"] == "patches/s" and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["config"]["global_batch"] == 64 and r["config"]["parallelism"] == "dp2" and r["config"]["dp_plan"] == "split"
    assert abs(r["value"] - 3 * 64 / (r["ms_per_step"] * 3 / 1e3)) < 1e-2 * r["value"]
    assert "resident" not in r["data"]
    # N > 1: the line says how long the optimiser waited for the gradient exchange behind the backward pass
    assert r["allreduce_exposed_us"] is not None and r["allreduce_exposed_us"] >= 0 and r["allreduce"]["steps"] == 20
    assert sum(r["allreduce"]["collectives_per_step_bytes"]) == 4 * 1269129


def test_bench_dp_plan_is_echoed():
    """SSDN_DP_PLAN picks the weight-gradient plan of the benchmarked engines (a scaling run can A/B "split" against "buckets")."""
    p, lines = _run({"SSDN_BENCH_STUB": "1", "SSDN_DP_PLAN": "buckets"}, "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert json.loads(lines[-1])["config"]["dp_plan"] == "buckets"
    p, lines = _run({"SSDN_BENCH_STUB": "1", "SSDN_DP_PLAN": "nonsense"}, "--steps", "2", "--warmup", "1")
    assert p.returncode != 0 and "error" in json.loads(lines[-1])


def test_bench_single_rank_stub_line():
    p, lines = _run({"SSDN_BENCH_STUB": "1"}, "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 1 and r["config"]["global_batch"] == 32 and r["stub"] is True


@pytest.mark.skipif(torch.cuda.is_available(), reason="describes the behaviour on a box WITHOUT a GPU")
def test_bench_failure_is_one_json_error_line():
    p, lines = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert p.returncode != 0
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert "error" in r and r["n_gpus"] == 2
