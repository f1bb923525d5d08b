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
