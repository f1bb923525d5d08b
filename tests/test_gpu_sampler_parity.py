"""GPU parity, part 2: the HIP sampler against the block-walking CPU oracle on seeded
power-law temporal graphs — every SamplingResult array bit-for-bit (int64 ids,
float32 timestamps / deltas), for recent and uniform policies, multi-layer,
multi-snapshot, windows, prop_time, ties, chunked ingestion, offload, duplicate and
out-of-range roots, and several search-group widths."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
MB = 1 << 20


def _graphs(min_block, policy="insert", adaptive=True):
    import gnnflow_amd
    from oracle import oracle as O
    g = gnnflow_amd.DynamicGraph(1 * MB, 1024 * MB, "cuda", min_block, 128, policy,
                                 adaptive_block_size=adaptive)
    o = O.OracleGraph(minimum_block_size=min_block, insertion_policy=policy,
                      adaptive_block_size=adaptive)
    return g, o


def _cmp_blocks(hip_mfgs, ora_mfgs, where):
    assert len(hip_mfgs) == len(ora_mfgs)
    for li, (hl, ol) in enumerate(zip(hip_mfgs, ora_mfgs)):
        assert len(hl) == len(ol)
        for si, (hb, ob) in enumerate(zip(hl, ol)):
            w = "{} block[{}][{}]".format(where, li, si)
            assert hb.num_src_nodes() == ob.num_src_nodes(), w
            assert hb.num_dst_nodes() == ob.num_dst_nodes(), w
            for key in ("ID", "ts"):
                a = hb.srcdata[key].cpu().numpy()
                assert a.dtype == ob.srcdata[key].dtype, w
                assert np.array_equal(a, ob.srcdata[key]), w + " srcdata " + key
            for key in ("dt", "ID"):
                a = hb.edata[key].cpu().numpy()
                assert a.dtype == ob.edata[key].dtype, w
                assert np.array_equal(a.view(np.uint8), ob.edata[key].view(np.uint8)), \
                    w + " edata " + key
            assert np.array_equal(hb.edges()[0].cpu().numpy(), ob.edges()[0]), w + " col"
            assert np.array_equal(hb.edges()[1].cpu().numpy(), ob.edges()[1]), w + " row"


CONFIGS = [
    dict(fanouts=[10], sample_strategy="recent"),
    dict(fanouts=[10, 10], sample_strategy="recent"),
    dict(fanouts=[5, 3, 2], sample_strategy="recent"),
    dict(fanouts=[10, 10], sample_strategy="uniform", seed=99),
    dict(fanouts=[4, 4], sample_strategy="recent", num_snapshots=3, snapshot_time_window=50.0),
    dict(fanouts=[4], sample_strategy="uniform", num_snapshots=2, snapshot_time_window=100.0,
         prop_time=True),
    dict(fanouts=[7], sample_strategy="recent", snapshot_time_window=30.0),
    dict(fanouts=[40], sample_strategy="recent", prop_time=True),
    dict(fanouts=[3, 3], sample_strategy="recent", is_static=True),
]


@pytest.mark.parametrize("group", ["16", "4", "2"])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(i) for i in range(len(CONFIGS))])
def test_random_graph_bit_exact(cfg, group, monkeypatch):
    import gnnflow_amd
    from oracle import oracle as O
    monkeypatch.setenv("GNNFLOW_SEARCH_GROUP", group)
    N, E = 2000, 60000
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=7, tie_levels=5000)
    g, o = _graphs(min_block=8)
    synth.ingest_chunks(g, src, dst, ts, eid, 7000)
    synth.ingest_chunks(o, src, dst, ts, eid, 7000)
    hs = gnnflow_amd.TemporalSampler(g, **cfg)
    os_ = O.OracleSampler(o, **cfg)
    for it, R in enumerate([1, 63, 600, 1800]):
        nodes, t = synth.random_roots(N, R, 1000.0, seed=100 + it, extra_ids=[N + 5, N - 1])
        _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "R={}".format(R))


def test_graph_accessors_match_oracle_add_reverse_and_offload():
    N, E = 500, 20000
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=3, tie_levels=300)
    g, o = _graphs(min_block=4)
    synth.ingest_chunks(g, src, dst, ts, eid, 3000, add_reverse=True)
    synth.ingest_chunks(o, src, dst, ts, eid, 3000, add_reverse=True)

    def compare():
        assert g.num_edges() == o.num_edges()
        assert g.num_vertices() == o.num_vertices()
        assert g.num_source_vertices() == o.num_source_vertices()
        assert g.max_vertex_id() == o.max_vertex_id()
        assert np.array_equal(g.nodes(), o.nodes())
        assert np.array_equal(g.src_nodes(), o.src_nodes())
        assert np.array_equal(np.sort(g.edges()), np.sort(o.edges()))
        ids = np.arange(0, o.max_vertex_id() + 1)
        assert np.array_equal(g.out_degree(ids), o.out_degree(ids))
        assert g.avg_linked_list_length() == pytest.approx(o.avg_linked_list_length())
        assert g.get_graph_memory_usage() == o.get_graph_memory_usage()
        for v in list(range(0, 40)) + [int(ids[-1])]:
            for a, b in zip(g.get_temporal_neighbors(v), o.get_temporal_neighbors(v)):
                assert np.array_equal(a, b), v

    compare()
    assert g.offload_old_blocks(400.0) == o.offload_old_blocks(400.0)
    compare()
    # sampling after offload sees only the surviving blocks
    import gnnflow_amd
    from oracle import oracle as O
    hs = gnnflow_amd.TemporalSampler(g, [6, 6])
    os_ = O.OracleSampler(o, [6, 6])
    nodes, t = synth.random_roots(N, 700, 1000.0, seed=11)
    _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "after offload")
    # and ingest continues correctly afterwards
    src2, dst2, ts2, eid2 = synth.powerlaw_graph(N, 5000, seed=5)
    ts2 = (ts2 + 1000.0).astype(np.float32)
    eid2 = eid2 + E
    g.add_edges(src2, dst2, ts2, eid2)
    o.add_edges(src2, dst2, ts2, eid2)
    compare()
    nodes, t = synth.random_roots(N, 700, 2100.0, seed=12)
    _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "after re-ingest")


@pytest.mark.parametrize("policy,adaptive", [("replace", True), ("insert", False)])
def test_block_policies_match_oracle(policy, adaptive):
    N, E = 300, 8000
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=9)
    g, o = _graphs(min_block=4, policy=policy, adaptive=adaptive)
    synth.ingest_chunks(g, src, dst, ts, eid, 500)
    synth.ingest_chunks(o, src, dst, ts, eid, 500)
    assert g.avg_linked_list_length() == pytest.approx(o.avg_linked_list_length())
    assert g.get_graph_memory_usage() == o.get_graph_memory_usage()
    assert g.offload_old_blocks(500.0) == o.offload_old_blocks(500.0)
    assert g.num_edges() == o.num_edges()


def test_uniform_distribution_chi_square():
    """Uniform policy is distribution-matched, not bit-matched, to cuRAND: every
    in-window edge must be equally likely (chi-square over many independent roots)."""
    import gnnflow_amd
    n = 16
    g, _ = _graphs(min_block=4)
    g.add_edges(np.zeros(n, np.int64), np.arange(1, n + 1), np.arange(n, dtype=np.float32))
    s = gnnflow_amd.TemporalSampler(g, [8], "uniform", seed=2024)
    R = 20000
    b = s.sample(np.zeros(R, np.int64), np.full(R, 12.0, np.float32))[0][0]
    picked = b.srcdata["ID"][R:].cpu().numpy()
    assert len(picked) == R * 8
    counts = np.bincount(picked, minlength=n + 1)[1:13]   # 12 candidates: dst 1..12
    assert counts.sum() == R * 8
    expected = R * 8 / 12.0
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert chi2 < 40.0, chi2   # 11 dof: P(chi2 > 40) ~ 4e-5


def test_sample_layer_result_object_and_debug_sample():
    """SamplingResult accessor methods (api.cc:87-109) and the benchmark harness's
    `_sample` entry point (benchmarks/benchmark_sampler.py:83)."""
    import gnnflow_amd
    from oracle import oracle as O
    N, E = 400, 9000
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=21)
    g, o = _graphs(min_block=16)
    g.add_edges(src, dst, ts, eid)
    o.add_edges(src, dst, ts, eid)
    hs = gnnflow_amd.TemporalSampler(g, [5, 5])
    os_ = O.OracleSampler(o, [5, 5])
    nodes, t = synth.random_roots(N, 333, 1000.0, seed=1)
    res, _ = hs._sample(nodes, t)
    ref = os_.sample_raw(nodes, t)
    for l in range(2):
        r, q = res[l][0], ref[l][0]
        for name in ("row", "col", "all_nodes", "all_timestamps", "delta_timestamps", "eids"):
            assert np.array_equal(getattr(r, name)(), getattr(q, name)()), (l, name)
        assert r.num_src_nodes() == q.num_src_nodes()
        assert r.num_dst_nodes() == q.num_dst_nodes()


@pytest.mark.parametrize("R", [3000, 70000, (1 << 20) + 77])
def test_negative_timestamps_and_roots_around_the_newest_edge(R):
    """The node entry carries its newest edge's timestamp: a root later than that takes the whole
    segment without a search, and a window open at 0 starts at the first edge — but only on a
    graph without negative timestamps.  Here half of the timestamps ARE negative, roots sit
    before, at and after every node's newest edge, with and without a window — small layers,
    large ones (fences) and the lane-per-root pass — bit-exact against the oracle."""
    import gnnflow_amd
    from oracle import oracle as O
    N, E = 3000, 60000
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=31, tie_levels=700)
    ts = (ts - np.float32(500.0)).astype(np.float32)        # [-500, 500)
    # the reference's fresh block starts with end_timestamp = 0 and CHECKs it against a batch's
    # newest edge (temporal_block init; utils.cu:42-43), so every vertex's FIRST batch must end
    # at t >= 0: one batch, and one closing edge per vertex at t = 499.5
    src = np.concatenate([src, np.arange(N, dtype=np.int64)])
    dst = np.concatenate([dst, (np.arange(N, dtype=np.int64) + 1) % N])
    ts = np.concatenate([ts, np.full(N, 499.5, np.float32)])
    eid = np.arange(E + N, dtype=np.int64)
    E = E + N
    g, o = _graphs(min_block=8)
    g.add_edges(src, dst, ts, eid)
    o.add_edges(src, dst, ts, eid)
    rng = np.random.RandomState(R % 1000)
    nodes = rng.randint(0, N, R).astype(np.int64)
    # a third of the roots exactly at an edge time of their node (ties with the newest edge
    # included), a third later than everything, the rest anywhere, some before the first edge
    t = rng.uniform(-600.0, 600.0, R).astype(np.float32)
    pick = rng.randint(0, E, R)
    at_edge = rng.rand(R) < 0.33
    nodes[at_edge] = src[pick[at_edge]]
    t[at_edge] = ts[pick[at_edge]]
    t[rng.rand(R) < 0.33] = np.float32(1e6)
    for cfg in (dict(fanouts=[5, 2], sample_strategy="recent"),
                dict(fanouts=[4], sample_strategy="recent", snapshot_time_window=90.0),
                dict(fanouts=[3], sample_strategy="uniform", num_snapshots=2,
                     snapshot_time_window=250.0, seed=8)):
        hs = gnnflow_amd.TemporalSampler(g, **cfg)
        os_ = O.OracleSampler(o, **cfg)
        os_.threads = 8
        _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "neg ts R={} {}".format(R, cfg))


@pytest.mark.parametrize("strategy", ["recent", "uniform"])
def test_large_batch_parallel_scan_path(strategy):
    """Layers with more than 65 536 roots take the tiled 3-phase scan instead of the
    single-workgroup one; 70 000 roots x fanouts [3, 2] exercises it (layer 1 bound 280 000)
    bit-exactly against the oracle."""
    import gnnflow_amd
    from oracle import oracle as O
    N, E = 5000, 200000
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=13, tie_levels=20000)
    g, o = _graphs(min_block=16)
    synth.ingest_chunks(g, src, dst, ts, eid, 50000)
    synth.ingest_chunks(o, src, dst, ts, eid, 50000)
    cfg = dict(fanouts=[3, 2], sample_strategy=strategy, seed=5)
    hs = gnnflow_amd.TemporalSampler(g, **cfg)
    os_ = O.OracleSampler(o, **cfg)
    nodes, t = synth.random_roots(N, 70000, 1000.0, seed=77)
    _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "R=70000")


@pytest.mark.parametrize("hybrid", ["1", "0"])
@pytest.mark.parametrize("cfg", [
    dict(fanouts=[4, 3], sample_strategy="recent", snapshot_time_window=120.0),
    dict(fanouts=[3, 2], sample_strategy="uniform", num_snapshots=2, snapshot_time_window=200.0),
    dict(fanouts=[5], sample_strategy="recent", prop_time=True),
])
def test_large_layers_lane_per_root_search(cfg, hybrid, monkeypatch):
    """Layers of >= 2^20 roots use the lane-per-root search pass (segments of <= 16
    timestamps are resolved by one lane, the others go through a segmented worklist to the
    16-lane group search); windows, snapshots, out-of-range roots, offloaded prefixes —
    bit-exact against the oracle, and the group-per-root kernel alone
    (GNNFLOW_SAMPLER_HYBRID_SEARCH=0) gives the same."""
    import gnnflow_amd
    from oracle import oracle as O
    monkeypatch.setenv("GNNFLOW_SAMPLER_HYBRID_SEARCH", hybrid)
    N, E = 30000, 300000            # power law: most nodes have < 16 edges, a few have 10^4
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=21, tie_levels=5000)
    g, o = _graphs(min_block=8)
    synth.ingest_chunks(g, src, dst, ts, eid, 60000)
    synth.ingest_chunks(o, src, dst, ts, eid, 60000)
    assert g.offload_old_blocks(150.0) == o.offload_old_blocks(150.0)   # live_off != 0
    hs = gnnflow_amd.TemporalSampler(g, seed=3, **cfg)
    os_ = O.OracleSampler(o, seed=3, **cfg)
    os_.threads = 8
    R = (1 << 20) + 1234
    nodes, t = synth.random_roots(N, R, 1000.0, seed=5, extra_ids=[N + 9, N - 1, 0])
    _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "R=2^20+ " + hybrid)


def test_offload_to_file_writes_reference_record_layout(tmp_path, monkeypatch):
    """offload_old_blocks(to_file=True) writes temporal_block_<node>-<k>.bin with the
    reference's record layout (temporal_block_allocator.cu:182-221): size, capacity (u64),
    start_ts, end_ts (f32), dst[size] i64, ts[size] f32, eid[size] i64, prev, next (ptr)."""
    import struct
    monkeypatch.chdir(tmp_path)
    g, o = _graphs(min_block=4)
    s = np.array([0, 0, 0, 1, 1, 1])
    d = np.array([1, 2, 3, 1, 2, 3])
    for t0, e0 in ((0, 0), (3, 6)):
        g.add_edges(s, d, np.array([t0, t0 + 1, t0 + 2] * 2), np.arange(e0, e0 + 6))
        o.add_edges(s, d, np.array([t0, t0 + 1, t0 + 2] * 2), np.arange(e0, e0 + 6))
    assert g.offload_old_blocks(3.5, to_file=True) == o.offload_old_blocks(3.5) == 2
    for node, eids in ((0, [0, 1, 2, 6]), (1, [3, 4, 5, 9])):
        raw = open("temporal_block_{}-0.bin".format(node), "rb").read()
        size, cap = struct.unpack_from("<QQ", raw, 0)
        start_ts, end_ts = struct.unpack_from("<ff", raw, 16)
        assert (size, cap, start_ts, end_ts) == (4, 4, 0.0, 3.0)
        off = 24
        dst = np.frombuffer(raw, np.int64, size, off); off += 8 * size
        ts = np.frombuffer(raw, np.float32, size, off); off += 4 * size
        eid = np.frombuffer(raw, np.int64, size, off); off += 8 * size
        assert dst.tolist() == [1, 2, 3, 1] and ts.tolist() == [0, 1, 2, 3]
        assert eid.tolist() == eids
        assert len(raw) == off + 16      # prev, next
    assert g.num_edges() == o.num_edges() == 4


@pytest.mark.parametrize("seed", range(24))
def test_fuzzed_configurations_bit_exact(seed):
    """Randomly drawn graph shapes, block sizes, insertion policies, chunkings, fan-outs,
    snapshot / window / prop_time settings, policies and root batches — HIP vs oracle, every
    array bit for bit, including an offload in the middle of the stream."""
    import gnnflow_amd
    from oracle import oracle as O
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.choice([5, 50, 700, 4000]))
    E = int(rng.choice([40, 900, 20000]))
    ties = int(rng.choice([3, 60, 5000]))
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=seed, alpha=float(rng.choice([0.0, 1.0, 1.6])),
                                             tie_levels=ties)
    policy = str(rng.choice(["insert", "replace"]))
    g, o = _graphs(min_block=int(rng.choice([1, 4, 62])), policy=policy,
                   adaptive=bool(rng.randint(2)))
    chunk = int(rng.choice([7, 333, 6000]))
    reverse = bool(rng.randint(2))
    half = (E // 2 // chunk) * chunk
    synth.ingest_chunks(g, src[:half], dst[:half], ts[:half], eid[:half], chunk, add_reverse=reverse)
    synth.ingest_chunks(o, src[:half], dst[:half], ts[:half], eid[:half], chunk, add_reverse=reverse)
    if rng.randint(2) and half:
        cut = float(ts[half // 2])
        assert g.offload_old_blocks(cut) == o.offload_old_blocks(cut)
    synth.ingest_chunks(g, src[half:], dst[half:], ts[half:], eid[half:], chunk, add_reverse=reverse)
    synth.ingest_chunks(o, src[half:], dst[half:], ts[half:], eid[half:], chunk, add_reverse=reverse)
    layers = int(rng.randint(1, 4))
    snaps = int(rng.choice([1, 1, 2, 3]))
    cfg = dict(fanouts=[int(x) for x in rng.randint(1, 12, layers)],
               sample_strategy=str(rng.choice(["recent", "recent", "uniform"])),
               num_snapshots=snaps,
               snapshot_time_window=float(rng.choice([25.0, 200.0])) if snaps > 1
               else float(rng.choice([0.0, 0.0, 80.0])),
               prop_time=bool(rng.randint(2)), seed=int(rng.randint(1, 1 << 30)))
    hs = gnnflow_amd.TemporalSampler(g, **cfg)
    os_ = O.OracleSampler(o, **cfg)
    for it in range(3):
        R = int(rng.choice([1, 17, 300, 1500]))
        nodes, t = synth.random_roots(N, R, 1000.0, seed=seed * 10 + it, extra_ids=[N + 2])
        _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "seed={} cfg={}".format(seed, cfg))
    assert g.num_edges() == o.num_edges() and g.num_vertices() == o.num_vertices()


def test_million_edge_batch_in_one_call():
    """One add_edges call of > 2^20 edges (threaded gathers, id sets on the helper thread):
    unsorted timestamps inside the batch, ties, reverse edges, a sparse high source id — graph
    state and samples must equal the oracle's."""
    import gnnflow_amd
    from oracle import oracle as O
    N, E = 40000, 700000            # x2 with reverse edges = 1.4 M per call
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=77, tie_levels=2000)
    rng = np.random.RandomState(1)
    p = rng.permutation(E)          # arbitrary order inside the batch
    src, dst, ts, eid = src[p], dst[p], ts[p], eid[p]
    src[:3] = N + 70000             # far beyond the dense id range: several empty buckets
    g, o = _graphs(min_block=16)
    g.add_edges(src, dst, ts, eid, add_reverse=True)
    o.add_edges(src, dst, ts, eid, add_reverse=True)
    assert g.num_edges() == o.num_edges() and g.num_vertices() == o.num_vertices()
    assert g.num_source_vertices() == o.num_source_vertices()
    assert g.max_vertex_id() == o.max_vertex_id()
    ids = rng.randint(0, N, 2000)
    assert np.array_equal(g.out_degree(ids), o.out_degree(ids))
    assert g.get_graph_memory_usage() == o.get_graph_memory_usage()
    for v in [int(ids[0]), int(src[0]), int(np.bincount(src[3:]).argmax())]:
        for a, b in zip(g.get_temporal_neighbors(v), o.get_temporal_neighbors(v)):
            assert np.array_equal(a, b), v
    hs = gnnflow_amd.TemporalSampler(g, [10, 5])
    os_ = O.OracleSampler(o, [10, 5])
    nodes, t = synth.random_roots(N, 3000, 1000.0, seed=5, extra_ids=[N + 70000])
    _cmp_blocks(hs.sample(nodes, t), os_.sample(nodes, t), "after a 1.4 M-edge batch")


def test_many_asynchronous_samples_in_flight_and_a_failing_one():
    """Up to 4 samples may be begun before the first is waited for (more: the oldest is
    completed first); results come back in order and equal the synchronous call.  A sample
    whose enqueue fails raises from ITS wait() only and leaves the sampler usable."""
    import torch
    import gnnflow_amd
    N, E = 3000, 90000
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=17, tie_levels=3000)
    g = gnnflow_amd.DynamicGraph(1 << 20, 1 << 28, "cuda", 16, 128, "insert")
    synth.ingest_chunks(g, src, dst, ts, eid, 30000)
    s = gnnflow_amd.TemporalSampler(g, [6, 4], "recent")
    side = torch.cuda.Stream()
    reqs = [synth.random_roots(N, 400 + 37 * k, 1000.0, seed=k) for k in range(9)]
    dev = [(torch.from_numpy(n).cuda(), torch.from_numpy(t).cuda()) for n, t in reqs]
    for worker in (False, True):
        pend = [s.sample_async(n, t, stream=side, worker_enqueue=worker) for n, t in dev]
        got = [p.wait() for p in reversed(pend)][::-1]      # waited newest first on purpose
        for (n, t), mf in zip(reqs, got):
            want = s.sample(n, t)
            for gl, wl in zip(mf, want):
                assert torch.equal(gl[0].edata["ID"], wl[0].edata["ID"])
                assert torch.equal(gl[0].srcdata["ID"], wl[0].srcdata["ID"])
                assert torch.equal(gl[0].edges()[1], wl[0].edges()[1])
    # a failing enqueue in the middle of a pipeline
    R_bad = 777
    nb, tb = synth.random_roots(N, R_bad, 1000.0, seed=99)
    s._bytes_cache[R_bad] = 64                       # far too small: the native begin refuses
    ok1 = s.sample_async(*dev[0], stream=side, worker_enqueue=True)
    bad = s.sample_async(torch.from_numpy(nb).cuda(), torch.from_numpy(tb).cuda(), stream=side,
                         worker_enqueue=True)
    ok2 = s.sample_async(*dev[1], stream=side, worker_enqueue=True)
    a = ok1.wait()
    with pytest.raises(ValueError):
        bad.wait()
    with pytest.raises(ValueError):
        bad.wait()                                   # and again: the error sticks to ITS handle
    b = ok2.wait()
    del s._bytes_cache[R_bad]
    for mf, (n, t) in ((a, reqs[0]), (b, reqs[1])):
        want = s.sample(n, t)
        assert torch.equal(mf[0][0].edata["ID"], want[0][0].edata["ID"])
    assert len(s._inflight) == 0
    assert s.sample(nb, tb)[0][0].num_dst_nodes() > 0   # the same roots work once sized right


def test_philox_known_answers_on_the_device():
    """The three Random123 Philox4x32-10 known-answer vectors (kat_vectors) evaluated by a
    KERNEL (gf_debug_philox), not on the host: the uniform sampler and the CPU oracle share
    include/gnnflow_rng.h, so HIP-vs-oracle parity alone would not notice a device-side
    miscompile of the rounds that stayed uniform.  Plus 20 000 random (seed, slot, call)
    triples against the oracle's host evaluation."""
    import ctypes as C
    import torch
    from gnnflow_amd import _capi
    from oracle import oracle as O
    lib = _capi.load()
    full = 0xFFFFFFFFFFFFFFFF
    kat = [((0, 0, 0), 0x6627E8D5), ((full, full, full), 0x408F276D),
           (((0x299F31D0 << 32) | 0xA4093822, (0x85A308D3 << 32) | 0x243F6A88,
             (0x03707344 << 32) | 0x13198A2E), 0xD16CFE09)]      # counter / key = digits of pi
    rng = np.random.RandomState(7)
    extra = rng.randint(0, 1 << 63, size=(20000, 3)).astype(np.uint64) * 2 + \
        rng.randint(0, 2, size=(20000, 3)).astype(np.uint64)
    triples = np.concatenate([np.array([k for k, _ in kat], dtype=np.uint64), extra])
    d_in = torch.from_numpy(triples.view(np.int64)).cuda()
    d_out = torch.zeros(len(triples), dtype=torch.int32, device="cuda")
    _capi.check(lib.gf_debug_philox(C.c_void_p(d_in.data_ptr()), len(triples),
                                    C.c_void_p(d_out.data_ptr()), None))
    got = d_out.cpu().numpy().view(np.uint32)
    assert [int(x) for x in got[:3]] == [v for _, v in kat]
    want = np.array([O.philox_first(int(s), int(t), int(c)) for s, t, c in extra[:2000]],
                    dtype=np.uint32)
    assert np.array_equal(got[3:2003], want)
