"""GPU: the NATIVE multi-rank chains at world size 8 inside one process.

The one-GPU test box allows 6 processes on its card, so 8 ranks cannot run as processes; here the
8 ranks are 8 shards + 8 samplers + 8 caches of ONE process, each driven by its own thread,
meeting in the library's loopback transport (include/gnnflow_hip.h gf_loopback_comm_create: an
exchange = thread barrier + device copies out of the peers' send buffers).  What runs is exactly
what runs over RCCL: `gf_sampler_sample_partitioned_comm` (the slotted chain: plan -> request
slots out -> own share + serve -> reply slots back -> merge, per layer) and `gf_pull_round` (the
natively planned feature pull) — only the transport differs.

Checked on the REDDIT-shaped replay (config 2's graph, fanout [10,10], most-recent, batch 600):
  * every MFG of every rank bit-exact against OracleSampler over the WHOLE graph;
  * slack 1.2 — below the 1.61x largest bucket of this stream at P = 8 — overflows, the flag
    reaches all 8 ranks in the same exchange and every rank redoes the same samples through the
    variable-size exchange (over the same communicator), results still bit-exact;
  * owner-sharded features: every fetched row against the full tables, hit ratios and cached-id
    sets after every fetch against a LOCAL cache over the full tables started from the same
    state, sampler and cache sharing one communicator.
Reference: gnnflow/distributed/dist_sampler.py:244-314 (merge), kvstore.py:285-339 (pull).
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 8
MiB = 1 << 20


def _same(gb, wb):
    ok = np.array_equal(gb.srcdata["ID"].cpu().numpy(), wb.srcdata["ID"])
    ok &= np.array_equal(gb.srcdata["ts"].cpu().numpy(), wb.srcdata["ts"])
    ok &= np.array_equal(gb.edata["ID"].cpu().numpy(), wb.edata["ID"])
    ok &= np.array_equal(gb.edata["dt"].cpu().numpy().view(np.uint8),
                         np.asarray(wb.edata["dt"]).view(np.uint8))
    ok &= np.array_equal(gb.edges()[0].cpu().numpy(), wb.edges()[0])
    ok &= np.array_equal(gb.edges()[1].cpu().numpy(), wb.edges()[1])
    return bool(ok)


@pytest.fixture(scope="module")
def world():
    """The REDDIT-shaped stream, its 8 shards in HBM and the oracle over the whole graph."""
    import torch
    from gnnflow_amd import DynamicGraph, synthetic
    from gnnflow_amd.dist import PartitionedGraph
    from oracle import oracle as O
    g = synthetic.reddit_like(seed=42)
    dev = torch.device("cuda", 0)
    shards = [DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert") for _ in range(P)]
    parts = [PartitionedGraph(s, r, P) for r, s in enumerate(shards)]
    full = O.OracleGraph(minimum_block_size=62, insertion_policy="insert")
    for lo in range(0, g["num_edges"], 100000):
        sl = slice(lo, lo + 100000)
        for pg in parts:
            pg.add_edges(g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl])
        full.add_edges(g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl])
    batches = list(synthetic.replay_batches(g, 600, seed=42))
    return dict(g=g, dev=dev, shards=shards, full=full, batches=batches, O=O)


def _run_ranks(fn, n=P):
    """fn(rank) on n threads; returns their results, re-raising the first failure."""
    out, err = [None] * n, [None] * n

    def body(r):
        import torch
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                out[r] = fn(r)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:      # noqa: BLE001 — surfaced below
            import traceback
            err[r] = "rank {}: {}: {}\n{}".format(r, type(e).__name__, e,
                                                  traceback.format_exc()[-1500:])
    threads = [threading.Thread(target=body, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank's thread did not finish"
    bad = [e for e in err if e]
    assert not bad, "\n".join(bad)
    return out


def _to_host(mfgs):
    """detach a sample's blocks from the sampler's slab: numpy copies"""
    class B:
        pass
    out = []
    for mfg in mfgs:
        row = []
        for b in mfg:
            h = B()
            h.srcdata = {k: b.srcdata[k].cpu() for k in ("ID", "ts")}
            h.edata = {k: b.edata[k].cpu() for k in ("ID", "dt")}
            col, row_ = b.edges()
            col, row_ = col.cpu(), row_.cpu()
            h.edges = lambda c=col, r=row_: (c, r)
            row.append(h)
        out.append(row)
    return out


@pytest.mark.parametrize("slack,first,count", [(2.0, 400, 64), (1.2, 900, 96)])
def test_native_slotted_chain_at_world_8_matches_the_oracle(world, slack, first, count):
    import torch
    from gnnflow_amd import TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm
    w = world
    comms = NativeComm.loopback(P, w["dev"])
    mine = [w["batches"][first + r:first + count:P] for r in range(P)]

    def rank_body(r):
        part = DevicePartitionedSampler(TemporalSampler(w["shards"][r], [10, 10], "recent"),
                                        comm=comms[r], slack=slack, slot_roots=1800,
                                        narrow_ids=(slack == 2.0))   # 12-byte reply slots / 24
        assert part._P == P and part._rank == r and part.lanes == 1
        got = []
        # half of the batches one by one, the rest three in flight on a side stream (the redo of
        # an overflowed sample then happens inside wait(), younger samples of the lane first)
        half = len(mine[r]) // 2
        for roots, ts, _ in mine[r][:half]:
            got.append(_to_host(part.sample(roots, ts)))
        side = torch.cuda.Stream()
        rest = mine[r][half:]
        for lo in range(0, len(rest), 3):
            pend = [part.sample_async(torch.from_numpy(n).to(w["dev"]),
                                      torch.from_numpy(t).to(w["dev"]), stream=side,
                                      worker_enqueue=True) for n, t, _ in rest[lo:lo + 3]]
            got.extend(_to_host(p.wait()) for p in pend)
        return got, part.overflows

    res = _run_ranks(rank_body)
    ref = w["O"].OracleSampler(w["full"], [10, 10], "recent", threads=4)
    edges = 0
    for r in range(P):
        got, _ = res[r]
        assert len(got) == len(mine[r])
        for (roots, ts, _), mfgs in zip(mine[r], got):
            for gl, wl in zip(mfgs, ref.sample(roots, ts)):
                for gb, wb in zip(gl, wl):
                    assert _same(gb, wb), (r, slack)
                    edges += len(wb.edata["ID"])
    assert edges > 100000
    over = [res[r][1] for r in range(P)]
    if slack >= 2.0:
        assert over == [0] * P           # slack 2.0 never overflows on this stream
    else:
        # the flag travels in the slot headers: EVERY rank redid the SAME samples
        assert over[0] > 0 and over == [over[0]] * P, over
    for c in comms:
        c.close()


def test_sharded_feature_pull_at_world_8(world):
    import ctypes as C
    import torch
    from gnnflow_amd import TemporalSampler, _capi
    from gnnflow_amd.cache import LRUCache
    from gnnflow_amd.dist import (DevicePartitionedSampler, FeatureShards, NativeComm,
                                  ShardedFeatures)
    w = world
    g, dev = w["g"], w["dev"]
    N, E, d = g["num_nodes"], g["num_edges"], 32
    rng = np.random.RandomState(7)
    nfeat, efeat = rng.rand(N, d).astype(np.float32), rng.rand(E, d).astype(np.float32)
    nfeat_d, efeat_d = torch.from_numpy(nfeat).to(dev), torch.from_numpy(efeat).to(dev)
    comms = NativeComm.loopback(P, dev)
    first, count = 800, 48
    mine = [w["batches"][first + r:first + count:P] for r in range(P)]
    lib = _capi.load()

    def rank_body(r):
        part = DevicePartitionedSampler(TemporalSampler(w["shards"][r], [10, 10], "recent"),
                                        comm=comms[r], slack=2.0, slot_roots=1800)
        shards = ShardedFeatures(
            node=FeatureShards.from_full(nfeat, np.arange(N), r, P, dev),
            edge=FeatureShards.from_full(efeat, g["src"], r, P, dev), comm=comms[r])
        assert shards.P == P and shards.rank == r
        cache = LRUCache(0.2, 0.2, N, E, dev, None, None, d, d, kvstore_client=shards,
                         distributed=True)
        cache.init_cache()
        # a LOCAL cache over the full tables, started from the distributed cache's state
        # (cache.py:161-173: node cache empty, edge cache = the first rows of the rank's shard)
        ref = LRUCache(0.2, 0.2, N, E, dev, nfeat_d, efeat_d, d, d)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _capi.check(lib.gf_cache_init_rows(ref._node.h, None, 0, None, st))
        n = min(ref.edge_capacity, int(shards.edge.local_ids.shape[0]))
        ids, rows = shards.edge.local_ids[:n].contiguous(), shards.edge.rows[:n].contiguous()
        _capi.check(lib.gf_cache_init_rows(ref._edge.h, ids.data_ptr(), n, rows.data_ptr(), st))
        torch.cuda.current_stream().synchronize()
        pulled = 0
        for it, (roots, ts, eids) in enumerate(mine[r]):
            mfgs = part.sample(roots, ts)
            cache.fetch_feature(mfgs, eids)
            for b in mfgs[0]:
                assert torch.equal(b.srcdata["h"], nfeat_d[b.srcdata["ID"]]), ("h", r, it)
            for mfg in mfgs:
                for b in mfg:
                    if b.num_edges():
                        assert torch.equal(b.edata["f"], efeat_d[b.edata["ID"]]), ("f", r, it)
            assert torch.equal(cache.target_edge_features,
                               efeat_d[torch.from_numpy(eids).to(dev)]), ("target", r, it)
            # the same fetch on the local cache: identical hit counts and cached-id sets
            ref.fetch_feature(part.sample(roots, ts), eids)
            assert float(cache.cache_node_ratio) == float(ref.cache_node_ratio), ("node", r, it)
            assert float(cache.cache_edge_ratio) == float(ref.cache_edge_ratio), ("edge", r, it)
            for kind in ("node", "edge"):
                assert np.array_equal(cache.slot_ids(kind), ref.slot_ids(kind)), (kind, r, it)
            pulled = shards.rows_pulled
        cache.check_pulls()
        assert shards.host_syncs <= 2 * len(mine[r])     # one count read-back per fetch round
        return pulled

    pulled = _run_ranks(rank_body)
    assert all(p > 0 for p in pulled)      # 7/8 of the missed rows live on other ranks
    for c in comms:
        c.close()


@pytest.mark.parametrize("ranks,chain", [(2, 2), (2, 4), (3, 3), (3, 4), (5, 2), (5, 4)])
def test_shared_chains_with_ragged_and_empty_batches(ranks, chain):
    """Several samples per chain (gf_sampler_sample_partitioned_comm_group) where the partners
    differ: empty batches, single roots, different sizes on different ranks in the same exchange,
    two snapshots (no shared chains: single ones), a number of samples that does not fill the
    last chain (it goes out with those held when one of them is waited for) — every MFG against
    the oracle over the whole graph."""
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm, PartitionedGraph
    from oracle import oracle as O
    from tests import synth
    dev = torch.device("cuda", 0)
    src, dst, ts, eid = synth.powerlaw_graph(400, 12000, seed=5, tie_levels=500)
    full = O.OracleGraph(minimum_block_size=8)
    shards = [DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert") for _ in range(ranks)]
    parts = [PartitionedGraph(s, r, ranks) for r, s in enumerate(shards)]
    for lo in range(0, len(src), 2500):
        sl = slice(lo, lo + 2500)
        full.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=True)
        for pg in parts:
            pg.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=True)
    sizes = [0, 1, 97, 600, 2000, 3, 0, 1500, 37]          # 9 samples: the last is issued alone
    cases = [dict(fanouts=[6, 4], num_snapshots=1, snapshot_time_window=0.0),
             dict(fanouts=[5], num_snapshots=2, snapshot_time_window=60.0),
             # slots far too small: samples of a pair overflow (each on its own flag), every
             # rank redoes the same ones through the variable-size exchange
             dict(fanouts=[6, 4], num_snapshots=1, snapshot_time_window=0.0, slack=0.05)]
    for case, kw in enumerate(cases):
        kw = dict(kw)
        slack = kw.pop("slack", 2.0)
        kw_narrow = (case + ranks + chain) % 2 == 0     # 12-byte reply slots / 24-byte ones
        comms = NativeComm.loopback(ranks, dev)
        batches = [[synth.random_roots(400, sizes[(it + r) % len(sizes)], 1000.0,
                                       seed=1000 * r + it, extra_ids=[403])
                    for it in range(len(sizes))] for r in range(ranks)]

        def rank_body(r):
            part = DevicePartitionedSampler(
                TemporalSampler(shards[r], sample_strategy="recent", **kw), comm=comms[r],
                slack=slack, slot_roots=max(sizes), chain_samples=chain,
                narrow_ids=kw_narrow,
                # compact reply slots: roomy enough for this dense little graph where the test
                # asserts that nothing overflows, far too small in the last case (-> redo)
                edge_fill=1.0 if slack >= 2.0 else 0.05)
            side = torch.cuda.Stream()
            pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev),
                                      stream=side, worker_enqueue=True) for n, t in batches[r][:4]]
            got = [_to_host(p.wait()) for p in pend]
            pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev),
                                      stream=side) for n, t in batches[r][4:]]
            got += [_to_host(p.wait()) for p in pend]
            return got, (part.pairs, part.chained), part.overflows

        res = _run_ranks(rank_body, ranks)
        ref = O.OracleSampler(full, kw["fanouts"], "recent", num_snapshots=kw["num_snapshots"],
                              snapshot_time_window=kw["snapshot_time_window"])
        for r in range(ranks):
            got, pairs, over = res[r]
            if slack >= 2.0:
                assert over == 0
            else:
                assert over > 0 and over == res[0][2]      # the same samples on every rank
            # 4 samples, all waited for, then 5: chains of 2: 2 + 2, then 2 + 2 + the last alone;
            # of 3: 3 + one alone, then 3 + the two held ones; of 4: 4, then 4 + the last alone —
            # a sample alone travels as a chain of ONE (compact replies, same wire format)
            want = ({2: 5, 3: 4, 4: 3}[chain], 9) if kw["num_snapshots"] == 1 else (0, 0)
            assert pairs == want
            for (n, t), mfgs in zip(batches[r], got):
                for gl, wl in zip(mfgs, ref.sample(n, t)):
                    for gb, wb in zip(gl, wl):
                        assert _same(gb, wb), (ranks, r, kw)
        for c in comms:
            c.close()


@pytest.mark.parametrize("ranks", [3, 8])
def test_slot_capacity_adapts_on_every_rank_at_the_same_sample(ranks):
    """adapt_slack: two overflowed samples within 64 raise the slot capacity by a quarter for
    the chains issued from then on.  Every rank sees the same samples overflow, so all take
    the step at the same sample (else their equal-split exchanges would disagree on the slot
    size and the MFGs below could not come out right); chains already out with the old
    capacity may overflow too without counting again."""
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm, PartitionedGraph
    from oracle import oracle as O
    from tests import synth
    dev = torch.device("cuda", 0)
    src, dst, ts, eid = synth.powerlaw_graph(400, 12000, seed=9, tie_levels=500)
    full = O.OracleGraph(minimum_block_size=8)
    shards = [DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert") for _ in range(ranks)]
    parts = [PartitionedGraph(s, r, ranks) for r, s in enumerate(shards)]
    full.add_edges(src, dst, ts, eid, add_reverse=True)
    for pg in parts:
        pg.add_edges(src, dst, ts, eid, add_reverse=True)
    n_samples = 40
    batches = [[synth.random_roots(400, 600, 1000.0, seed=100 * r + it) for it in range(n_samples)]
               for r in range(ranks)]
    comms = NativeComm.loopback(ranks, dev)

    def rank_body(r):
        part = DevicePartitionedSampler(TemporalSampler(shards[r], [6, 4], "recent"), comm=comms[r],
                                        slack=0.5, slot_roots=600, adapt_slack=True)
        side = torch.cuda.Stream()
        got = []
        for lo in range(0, n_samples, 6):
            pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev),
                                      stream=side) for n, t in batches[r][lo:lo + 6]]
            got += [_to_host(p.wait()) for p in pend]
        return got, (part.overflows, part._slack, part._slack_epoch)

    res = _run_ranks(rank_body, ranks)
    ref = O.OracleSampler(full, [6, 4], "recent")
    state = res[0][1]
    assert state[0] >= 2 and state[2] >= 1 and 0.5 < state[1] <= 1.0
    for r in range(ranks):
        got, st = res[r]
        assert st == state                      # same overflows, same capacity, same step count
        for (n, t), mfgs in zip(batches[r], got):
            for gl, wl in zip(mfgs, ref.sample(n, t)):
                for gb, wb in zip(gl, wl):
                    assert _same(gb, wb), (ranks, r)
    for c in comms:
        c.close()
