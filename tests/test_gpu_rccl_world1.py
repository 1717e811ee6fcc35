"""The RCCL ("nccl" backend) calls of the product path on a one-GPU box: a world-size-1 RCCL group with the
schedules forced onto their multi-step code path (tests/_rccl_w1_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_every_schedule_runs_its_exchange_on_rccl_world_size_1():
    from conftest import free_port

    env = dict(os.environ)
    for k in ("RFA_ZIGZAG_EXCHANGE", "RFA_DKV_WIRE", "RFA_LLAMA3_GATHER_MAX_BYTES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_w1_worker.py"), str(free_port())],
                       capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ALL OK" in r.stdout


@pytest.mark.parametrize("workload,exchange", [("zigzag", "gather"), ("zigzag", "ring"), ("zigzag", "gather_ps"), ("llama3", None)])
def test_bench_multi_rank_branches_on_rccl_world_size_1(workload, exchange):
    """bench.py's N > 1 branches (RCCL init with device_id, barrier + all_reduce(MAX) on device tensors, the
    fixed-count spin-up, the loopback `comm` block) on a one-rank RCCL group: one JSON line, with the block"""
    import json

    env = dict(os.environ)
    env["RFA_BENCH_FORCE_RCCL"] = "1"
    env["MASTER_PORT"] = str(__import__("conftest").free_port())
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-breakdown", "--workload", workload]
    # the N > 1 line's CPU baseline (the zigzag schedule over gloo CPU processes, oracle/cpu_ring_baseline.py) rides on one
    # of the cases, with a budget that keeps it to the 2048-token probe
    cmd += ["--cpu-baseline-budget-s", "1"] if exchange == "gather" else ["--no-cpu-baseline"]
    if exchange:
        cmd += ["--exchange", exchange]
    if exchange in ("gather_ps", "ring"):
        cmd += ["--exchange-check-steps", "2"]       # the audited steps of profiles/collect_scale.sh
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["forced_rccl_world1"] and d["comm"]["backend"] == "nccl" and d["comm"]["compute_only_ms"] > 0
    assert d["value"] > 0 and d["n_gpus"] == 1
    assert d["comm"]["exchange_check_steps_passed"] == (2 if exchange in ("gather_ps", "ring") else 0)
    if exchange == "gather":
        cb = d["cpu_baseline"]
        assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 2 and "gloo CPU processes" in cb["sample"]
        assert not d.get("errors", {}).get("cpu_baseline")
