"""Child process of tests/test_gpu_pipeline_parity.py::test_fused_lru_update_fallback_paths: the one-launch
LRU list update with GNNFLOW_LRU_FUSE_SPINS=0 — every look-back gives up at once and recomputes
its value from the launch's inputs (fuse_recount_rows / fuse_recount_tile / fuse_walk_tile /
the direct walk) — checked step by step against the oracle.  Prints the number of recounts."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("GNNFLOW_LRU_FUSE_SPINS") == "0"

from tests import test_gpu_pipeline_parity as T  # noqa: E402
from gnnflow_amd import _capi  # noqa: E402
from gnnflow_amd.pipeline import ReplayPipeline  # noqa: E402

steps = 40
for first, ratio in ((900, 0.2), (100, 0.01)):
    w = T._World(first, steps, cache_ratio=ratio)
    pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=True)

    def on_step(i, mfgs):
        snap = T._snapshot(w.cache, mfgs)
        w.torch.cuda.synchronize()
        T._check_step(w, i, snap)
        T._cached_sets_equal(w)

    pipe.run(0, steps, on_step)
    assert w.cache._edge.lru_state()["queue_form"] == 0
n = C.c_uint64(0)
_capi.check(_capi.load().gf_debug_lru_recounts(C.byref(n)))
print("recounts", n.value)
