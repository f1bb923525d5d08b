"""GPU parity, part 1: the HIP product path against the reference's own known-answer
tests (the same fixtures that pin the oracle), for all four memory resource types the
reference parameterises over."""
import pytest

from tests.golden_runner import load_cases, run_case

pytestmark = pytest.mark.gpu
CASES = load_cases()
MB, GB = 1 << 20, 1 << 30


def _make_graph(mem):
    import gnnflow_amd

    def make(**kw):
        cfg = {"initial_pool_size": 1 * MB, "maximum_pool_size": 64 * MB,
               "mem_resource_type": mem, "minimum_block_size": 64,
               "blocks_to_preallocate": 128, "insertion_policy": "insert"}
        cfg.update(kw)
        return gnnflow_amd.DynamicGraph(**cfg)
    return make


def _make_sampler(g, **kw):
    import gnnflow_amd
    return gnnflow_amd.TemporalSampler(g, **kw)


@pytest.mark.parametrize("mem", ["cuda", "unified", "pinned", "shared"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_matches_reference_golden(case, mem):
    run_case(case, _make_graph(mem), _make_sampler)


def test_older_edges_raise_value_error():
    """tests/test_dynamic_graph.py:324-344 (skipped upstream as "not implemented"):
    the documented ValueError, instead of the reference's CHECK -> abort."""
    import numpy as np
    g = _make_graph("cuda")()
    g.add_edges(np.array([0, 1, 2]), np.array([1, 2, 3]), np.array([0, 1, 2]))
    with pytest.raises(ValueError):
        g.add_edges(np.array([2]), np.array([1]), np.array([0]))
    # the failed call must not have changed the graph
    assert g.num_edges() == 3
    assert g.out_degree(np.array([2])).tolist() == [1]
