"""Pins the CPU oracle to the reference's own known-answer tests:
every literal expectation of tests/test_dynamic_graph.py and
tests/test_temporal_sampler.py, transcribed into tests/golden/reference_tests.json."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.golden_runner import load_cases, run_case

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_golden(case):
    run_case(case, O.OracleGraph, O.OracleSampler)


def test_philox_known_answers():
    # Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)
    assert O.philox_first(0, 0, 0) == 0x6627E8D5
    full = 0xFFFFFFFFFFFFFFFF
    assert O.philox_first(full, full, full) == 0x408F276D
    # counter = digits of pi, key = more digits of pi
    tid = (0x85A308D3 << 32) | 0x243F6A88
    call = (0x03707344 << 32) | 0x13198A2E
    seed = (0x299F31D0 << 32) | 0xA4093822
    assert O.philox_first(seed, tid, call) == 0xD16CFE09


def test_block_policy_matches_reference_walkthrough():
    """Block layout of the reference's offload test (minimum_block_size=4, two
    batches of 3 edges/node): block 0 = [0,1,2,3] full, block 1 holds [4,5] with
    capacity pow2(max(2, 3/1)) = 4 (dynamic_graph.cu:226-260)."""
    g = O.OracleGraph(minimum_block_size=4)
    s = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2])
    d = np.array([1, 2, 3, 1, 2, 3, 1, 2, 3])
    g.add_edges(s, d, np.array([0, 1, 2] * 3))
    assert g.blocks(0) == [(3, 4, 0.0, 2.0)]
    g.add_edges(s, d, np.array([3, 4, 5] * 3))
    assert g.blocks(0) == [(4, 4, 0.0, 3.0), (2, 4, 4.0, 5.0)]
    assert g.avg_linked_list_length() == pytest.approx(6 / 4)
    assert g.get_graph_memory_usage() == 6 * 4 * 20
    assert g.offload_old_blocks(3.5) == 3
    assert g.blocks(0) == [(2, 4, 4.0, 5.0)]
    assert g.num_edges() == 6


def test_replace_policy_single_block():
    g = O.OracleGraph(minimum_block_size=4, insertion_policy="replace")
    s = np.array([0, 0, 0])
    g.add_edges(s, np.array([1, 2, 3]), np.array([0, 1, 2]))
    g.add_edges(s, np.array([1, 2, 3]), np.array([3, 4, 5]))
    assert g.blocks(0) == [(6, 6, 0.0, 5.0)]


def test_older_edges_rejected():
    g = O.OracleGraph()
    g.add_edges(np.array([0, 1, 2]), np.array([1, 2, 3]), np.array([5, 6, 7]))
    with pytest.raises(ValueError):
        g.add_edges(np.array([0]), np.array([1]), np.array([1]))


def test_uniform_is_with_replacement_and_in_window():
    g = O.OracleGraph(minimum_block_size=4)
    n = 50
    g.add_edges(np.zeros(n, np.int64), np.arange(1, n + 1), np.arange(n, dtype=np.float32))
    s = O.OracleSampler(g, [8], "uniform", seed=7)
    b = s.sample(np.array([0, 0]), np.array([20.0, 0.0]))[0][0]
    # root 0 at t=20: 20 candidates -> all 8 slots valid; root 1 at t=0: none
    assert b.num_src_nodes() == 2 + 8
    assert (b.srcdata["ts"][2:] < 20).all()
    assert b.edges()[1].tolist() == [0] * 8
    # a second call advances the stream
    b2 = s.sample(np.array([0, 0]), np.array([20.0, 0.0]))[0][0]
    assert b2.edata["ID"].tolist() != b.edata["ID"].tolist()


def test_threaded_oracle_is_identical_to_the_scalar_one():
    """bench.py's cpu_baseline may run the oracle's OpenMP form; it must be the same function."""
    import numpy as np
    from oracle import oracle as O
    from tests import synth
    src, dst, ts, eid = synth.powerlaw_graph(300, 9000, seed=4, tie_levels=200)
    g = O.OracleGraph(minimum_block_size=8)
    synth.ingest_chunks(g, src, dst, ts, eid, 1000, add_reverse=True)
    for strategy in ("recent", "uniform"):
        a = O.OracleSampler(g, [7, 3], strategy, num_snapshots=2, snapshot_time_window=80.0)
        b = O.OracleSampler(g, [7, 3], strategy, num_snapshots=2, snapshot_time_window=80.0,
                            threads=4)
        nodes, t = synth.random_roots(300, 500, 1000.0, seed=1, extra_ids=[305])
        for la, lb in zip(a.sample(nodes, t), b.sample(nodes, t)):
            for x, y in zip(la, lb):
                for k in ("ID", "ts"):
                    assert np.array_equal(x.srcdata[k], y.srcdata[k])
                for k in ("ID", "dt"):
                    assert np.array_equal(x.edata[k], y.edata[k])
                assert np.array_equal(x.edges()[1], y.edges()[1])
    feats = np.random.RandomState(0).rand(50, 9).astype(np.float32)
    ids = np.random.RandomState(1).randint(0, 50, 1000)
    assert np.array_equal(O.gather_rows(feats, ids), O.gather_rows(feats, ids, threads=4))
