"""bench.py's own launcher (`python bench.py --gpus N` without torchrun) on a box without a GPU:
the children fail their GPU assertion, the parent — which must not import torch or touch HIP —
still prints exactly one JSON line and exits non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_reports_failing_ranks_with_one_line():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = ""          # also on a GPU box: no rank finds a device
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    # every rank is a supervisor that walks the ladder (hash, hash-simple, replica) with a fresh
    # worker per rung; here every worker dies at once, and rank 0's supervisor says so
    assert d["value"] == 0.0 and d["n_gpus"] == 2 and "no rung printed a record" in d["error"]
    assert [h["arrangement"] for h in d["ladder"]] == ["hash", "hash-simple"]
    assert all("exited with status" in h["error"] and not h["worker_killed"] for h in d["ladder"])
    assert "bench.py needs a GPU" in p.stderr          # the workers' own message got through


def test_launcher_parent_does_not_load_torch():
    """The parent's work happens in launch_ranks(): importing bench and calling parse() must
    not pull torch in (a process that initialised the GPU must not start the ranks)."""
    code = ("import sys; sys.argv=['bench.py','--gpus','2']; import bench; "
            "a = bench.parse(sys.argv[1:]); assert a.gpus == 2; "
            "assert 'torch' not in sys.modules and 'gnnflow_amd' not in sys.modules")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True,
                       timeout=60)
    assert p.returncode == 0, p.stderr
