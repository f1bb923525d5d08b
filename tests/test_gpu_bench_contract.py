"""bench.py's output contract on a real GPU: one JSON line from rank 0 with the fields the
driver reads, the roofline / cpu_baseline objects, and a 2-rank launch through torchrun (both
ranks on the one GPU of the test box; gloo carries the bookkeeping collectives)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int,
            "warmup": int, "ms_per_step": float, "higher_is_better": bool, "scaling": str,
            "dtype": str, "data": str, "config": dict}


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]       # exactly ONE JSON line
    return json.loads(lines[0])


def _check_common(d, n_gpus, steps, warmup):
    for k, t in REQUIRED.items():
        assert k in d and isinstance(d[k], t), k
    assert d["metric"] == "sampled_edges_per_s" and d["unit"] == "edges/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (n_gpus, steps, warmup)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_single_gpu_line_with_roofline_and_cpu_baseline():
    d = _run([sys.executable, "bench.py", "--steps", "60", "--warmup", "5", "--cpu-seconds", "1"])
    _check_common(d, 1, 60, 5)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert 0 < r["frac"] < 1 and r["launches_timed"] > 0 and r["avg_launch_us"] > 0
    # PMC traffic is attached only to the command line it was measured with (the driver's)
    assert r["traffic"] is None
    # the timed region covers whole chronological replays whatever --steps is ...
    assert d["repeats"] == 75 and d["config"]["timed_steps"] == 60 * 75
    assert d["ms_per_step"] == pytest.approx(1e3 * d["config"]["timed_seconds"] / (60 * 75))
    assert d["config"]["timed_seconds"] >= 0.1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "edges/s" and c["value"] > 0
    assert c["single_thread_value"] > 0
    assert isinstance(c["sample"], str) and c["sample"]
    assert d["value"] > c["value"]
    # ... and the CPU baseline ran the same batches: same sampled edges per step
    assert c["edges_per_step"] == pytest.approx(d["config"]["edges_per_step"], rel=0.02)
    # ... and moved the same feature rows (both legs serve the roots' layer's edge block from
    # the outer block's rows)
    assert c["rows_per_step"] == pytest.approx(d["config"]["rows_per_step"], rel=0.02)
    assert c["host_logical_cpus"] >= c["cores"]
    # north_star's hash-partitioned split rides in the same line (one rank: no exchange)
    h = d["hash_partition"]
    assert "error" not in h and h["world_size"] == 1 and h["value"] > 0 and h["steps"] == 1121
    assert h["value"] > 0.5 * d["value"]         # the hash path is not a second-class citizen


def test_driver_command_line_is_representative():
    """`--steps 20 --warmup 5` (what the driver runs): full-replay mean edges per step."""
    d = _run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    _check_common(d, 1, 20, 5)
    assert d["repeats"] == 225
    full = _run([sys.executable, "bench.py", "--no-cpu-baseline"])
    assert full["repeats"] == 4
    assert d["config"]["edges_per_step"] == pytest.approx(full["config"]["edges_per_step"],
                                                         rel=0.02)


def test_two_ranks_through_torchrun():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2",
              "--steps", "40", "--warmup", "5", "--min-replays", "1"],
             env={"GNNFLOW_BENCH_DEVICE": "0", "GNNFLOW_BENCH_BACKEND": "gloo"})
    _check_common(d, 2, 40, 5)
    # N > 1: the headline is north_star's split — the graph hash-partitioned over the ranks,
    # both ranks exchanging roots / replies per layer (staged through gloo on this one-GPU box)
    assert d["config"]["parallelism"] == "hash-dp2"
    assert "equal-split" in d["config"]["exchange"] and "no host sync" in d["config"]["exchange"]
    assert "cpu_baseline" not in d       # rank 0 at N = 1 only
    r = d["replica"]                     # the per-GPU-replica figure rides along
    assert "error" not in r and r["world_size"] == 2 and r["value"] > 0
