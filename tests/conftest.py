import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Order of the GPU run: the hot path's parity files first, the files that start further
# processes on the card last — a late failure (or a box limit) must not keep the parity
# tests of the timed path from running.  Files not named keep their alphabetical place
# between the two groups.
_FIRST = ["test_gpu_golden", "test_gpu_sampler_parity", "test_gpu_pipeline_parity",
          "test_gpu_cache", "test_gpu_fuzz", "test_gpu_harness", "test_errors",
          "test_gpu_config3", "test_gpu_pybind_libgnnflow", "test_memory",
          "test_gpu_block_ops"]
_LAST = ["test_gpu_partitioned", "test_gpu_dist_features", "test_gpu_configs_4_5",
         "test_gpu_bench_contract"]

# The GPU box kills a run in which more than 6 processes hold the card.  A multi-process GPU
# test may start at most this many ranks (+ the pytest parent = 5 holders).
MAX_GPU_RANKS = 4


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _FIRST:
        return (0, _FIRST.index(name))
    if name in _LAST:
        return (2, _LAST.index(name))
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped when no device is visible, so a plain `pytest tests/`
    on a CPU box is green; `-m gpu` on the GPU box runs them for real."""
    items.sort(key=_file_rank)          # stable: order inside a file is kept
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
