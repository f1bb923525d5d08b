"""GPU: the end-to-end usage example (examples/train_edge_prediction.py) — batch pipeline →
sampler → LRU feature cache → SAGEConv on the block ops → backward → optimizer — composes and
learns on synthetic data."""
import importlib.util
import math
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_example_runs_and_loss_goes_down():
    spec = importlib.util.spec_from_file_location(
        "train_edge_prediction", os.path.join(ROOT, "examples", "train_edge_prediction.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    losses = mod.main(num_batches=40, verbose=False)
    assert len(losses) == 40 and all(math.isfinite(x) for x in losses)
    assert sum(losses[-5:]) / 5 < sum(losses[:5]) / 5


def test_tgn_epoch_stand_in_runs_and_reports_its_stages():
    """examples/tgn_epoch.py (config 5's epoch-time stand-in) on a small MAG-shaped graph: one
    JSON line with seconds per epoch and the per-stage split; the loss is finite."""
    import json
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "tgn_epoch.py"),
                        "--nodes", "20000", "--edges", "400000", "--dim-node", "64", "--batch",
                        "1000", "--epochs", "1", "--max-batches", "12"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    ep = d["epochs"][0]
    assert ep["batches"] == 12 and ep["seconds"] > 0 and ep["sampled_edges"] > 0
    assert set(ep["host_seconds_by_stage"]) == {"sample", "fetch", "mem_gather", "mem_update",
                                                "model", "write_back"}
    assert ep["loss"] == ep["loss"] and abs(ep["loss"]) < 100      # finite
