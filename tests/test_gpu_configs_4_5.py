"""BASELINE configs 4 and 5 at reduced scale, every subsystem they involve together, on the one
GPU of the test box (several ranks share it; the exchanges are staged through gloo):

  config 4  GDELT-shaped, TGAT (2 layers, fanout [10,10], uniform; gnnflow/config.py:45-59,
            157-167: minimum block 123, node + edge features 413 / 186), graph HASH-PARTITIONED
            over the ranks, features SHARDED by owner, LRU cache per GPU;
  config 5  MAG-shaped, TGN (1 layer, fanout [10], most-recent, batch 4000, memory module;
            config.py:28-43,169-179: minimum block 11, 768-d node features, no edge features).

What is checked: MFGs of the partitioned sampler against the CPU oracle over the WHOLE graph
(bit-exact for most-recent; for uniform every property that does not depend on the draws:
membership, window, `fanout iff candidates`), every fetched row against the full tables, the
TGN memory tables after every step against oracle/memory_oracle.py.  Sizes: GDELT's real node
count with 1/100 of its edges; 1/100 of MAG's nodes and edges.  No multi-GPU box has been
available, so RCCL itself is not exercised.
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MiB = 1 << 20


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(num_nodes, num_edges, seed, alpha):
    from gnnflow_amd import synthetic
    return synthetic.powerlaw(num_nodes, num_edges, seed=seed, alpha=alpha, t_max=1e5)


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(fn, args=(world, _free_port(), ret) + args, nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}, dict(ret)


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


# ---- config 4 ------------------------------------------------------------------------------
GD = dict(N=16682, E=1_900_000, d_n=413, d_e=186, min_block=123, batch=600)


def _config4_worker(rank, world, port, ret, policy):
    dist = _init(rank, world, port)
    try:
        import torch
        from gnnflow_amd import DynamicGraph, TemporalSampler
        from gnnflow_amd.cache import LRUCache
        from gnnflow_amd.dist import (DevicePartitionedSampler, FeatureShards, PartitionedGraph,
                                      ShardedFeatures)
        from gnnflow_amd.pipeline import ReplayPipeline
        from oracle import oracle as O
        g = _graph(GD["N"], GD["E"], seed=4, alpha=0.8)
        N, E = GD["N"], GD["E"]
        dev = torch.device("cuda", 0)
        shard = DynamicGraph(64 * MiB, 2048 * MiB, "unified", GD["min_block"], 8196, "insert")
        pg = PartitionedGraph(shard, rank, world)
        full = O.OracleGraph(minimum_block_size=GD["min_block"])
        for lo in range(0, E, 500_000):
            sl = slice(lo, lo + 500_000)
            pg.add_edges(g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl])
            full.add_edges(g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl])
        sampler = DevicePartitionedSampler(TemporalSampler(shard, [10, 10], policy, seed=99))
        ref = O.OracleSampler(full, [10, 10], "recent", threads=4)
        rng = np.random.RandomState(0)              # same tables on every rank
        nfeat = rng.rand(N, GD["d_n"]).astype(np.float32)
        efeat = rng.rand(E, GD["d_e"]).astype(np.float32)
        shards = ShardedFeatures(
            node=FeatureShards.from_full(nfeat, np.arange(N), rank, world, dev),
            edge=FeatureShards.from_full(efeat, g["src"], rank, world, dev))
        cache = LRUCache(0.2, 0.2, N, E, dev, None, None, GD["d_n"], GD["d_e"],
                         kvstore_client=shards, distributed=True)
        cache.init_cache()
        B = GD["batch"]
        r2 = np.random.RandomState(10 + rank)
        batches = []
        for it in range(6):                          # each rank replays its own batches
            lo = int(E * 0.9) + (it * world + rank) * B
            pick = np.arange(lo, lo + B)
            roots = np.concatenate([g["src"][pick], g["dst"][pick], r2.randint(0, N, B)])
            batches.append((roots.astype(np.int64), np.tile(g["ts"][pick], 3).astype(np.float32),
                            g["eid"][pick]))
        dbat = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(e).to(dev))
                for r, t, e in batches]
        src_d, dst_d, ts_d = (torch.from_numpy(g[k]).to(dev) for k in ("src", "dst", "ts"))
        checked = {"n": 0}

        def on_step(i, mfgs):
            roots, rts, eids = batches[i]
            if policy == "recent":
                want = ref.sample(roots, rts)
                for gl, wl in zip(mfgs, want):
                    for gb, wb in zip(gl, wl):
                        assert np.array_equal(gb.srcdata["ID"].cpu().numpy(), wb.srcdata["ID"])
                        assert np.array_equal(gb.srcdata["ts"].cpu().numpy(), wb.srcdata["ts"])
                        assert np.array_equal(gb.edata["ID"].cpu().numpy(), wb.edata["ID"])
                        assert np.array_equal(gb.edata["dt"].cpu().numpy(), wb.edata["dt"])
                        assert np.array_equal(gb.edges()[1].cpu().numpy(), wb.edges()[1])
            else:
                # uniform: what holds whatever the draws are
                for mfg in mfgs:
                    for b in mfg:
                        R = b.num_dst_nodes()
                        eid, row = b.edata["ID"], b.edges()[1]
                        rid, rt = b.srcdata["ID"][:R], b.srcdata["ts"][:R]
                        assert torch.equal(src_d[eid], rid[row])
                        assert torch.equal(dst_d[eid], b.srcdata["ID"][R:])
                        assert bool((ts_d[eid] < rt[row]).all())
                        counts = torch.bincount(row, minlength=R)
                        assert bool(((counts == 10) | (counts == 0)).all())
            for b in mfgs[0]:
                assert np.array_equal(b.srcdata["h"].cpu().numpy(),
                                      nfeat[b.srcdata["ID"].cpu().numpy()])
            for mfg in mfgs:
                for b in mfg:
                    if b.num_edges():
                        assert np.array_equal(b.edata["f"].cpu().numpy(),
                                              efeat[b.edata["ID"].cpu().numpy()])
            assert np.array_equal(cache.target_edge_features.cpu().numpy(), efeat[eids])
            checked["n"] += 1

        ReplayPipeline(sampler, cache, dbat, dev, pipelined=True).run(0, len(dbat), on_step)
        torch.cuda.synchronize()
        assert checked["n"] == len(dbat)
        assert shards.edge.rows_pulled > 0 and shards.node.rows_pulled > 0
        ret[rank] = True
    except Exception as e:
        import traceback
        ret[rank] = "{}: {}\n{}".format(type(e).__name__, e, traceback.format_exc()[-1200:])
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("policy", ["recent", "uniform"])
@pytest.mark.parametrize("world", [2, 4])
def test_config4_gdelt_shaped_partitioned_sampling_and_sharded_features(world, policy):
    _spawn(_config4_worker, world, policy)


# ---- config 5 ------------------------------------------------------------------------------
MG = dict(N=1_220_000, E=13_000_000, d_n=768, min_block=11, batch=4000, d_mem=100)


def _config5_worker(rank, world, port, ret):
    dist = _init(rank, world, port)
    try:
        import torch
        from gnnflow_amd import DynamicGraph, TemporalSampler
        from gnnflow_amd.cache import LRUCache
        from gnnflow_amd.dist import (DevicePartitionedSampler, FeatureShards, PartitionedGraph,
                                      ShardedFeatures)
        from gnnflow_amd.memory import Memory
        from oracle import oracle as O
        from oracle.memory_oracle import OracleMemory
        g = _graph(MG["N"], MG["E"], seed=5, alpha=1.0)
        N, E, B = MG["N"], MG["E"], MG["batch"]
        dev = torch.device("cuda", 0)
        shard = DynamicGraph(256 * MiB, 8192 * MiB, "unified", MG["min_block"], 65536, "insert")
        pg = PartitionedGraph(shard, rank, world)
        for lo in range(0, E, 5_000_000):
            sl = slice(lo, lo + 5_000_000)
            pg.add_edges(g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl])
        sampler = DevicePartitionedSampler(TemporalSampler(shard, [10], "recent"))
        # node features only (MAG has no edge features); generated per id so that no rank needs
        # the whole 3.7 GB table to check a row
        def rows_of(ids):
            ids = np.asarray(ids, np.int64)
            base = (ids[:, None] * 31 + np.arange(MG["d_n"])[None, :]) % 1009
            return (base / 1009.0).astype(np.float32)
        from gnnflow_amd.dist import owner_of_np
        own = np.nonzero(owner_of_np(np.arange(N), world) == rank)[0]
        node_shards = FeatureShards(N, torch.from_numpy(own).to(dev),
                                    torch.from_numpy(rows_of(own)).to(dev))
        cache = LRUCache(0.0, 0.2, N, E, dev, None, None, MG["d_n"], 0,
                         kvstore_client=ShardedFeatures(node=node_shards), distributed=True)
        cache.init_cache()
        mem = Memory(N, 0, MG["d_mem"], dev)
        omem = OracleMemory(N, 0, MG["d_mem"])
        # the oracle samples on the edges whose source is a root of this rank's batches only
        # (a 1-layer sample needs nothing else), so that it stays small
        r2 = np.random.RandomState(50 + rank)
        batches = []
        for it in range(4):
            lo = int(E * 0.95) + (it * world + rank) * B
            pick = np.arange(lo, lo + B)
            roots = np.concatenate([g["src"][pick], g["dst"][pick], r2.randint(0, N, B)])
            batches.append((roots.astype(np.int64), np.tile(g["ts"][pick], 3).astype(np.float32)))
        needed = np.unique(np.concatenate([r for r, _ in batches]))
        mark = np.zeros(N, bool)
        mark[needed] = True
        sel = np.nonzero(mark[g["src"]])[0]
        og = O.OracleGraph(minimum_block_size=MG["min_block"])
        for lo in range(0, len(sel), 2_000_000):
            s = sel[lo:lo + 2_000_000]
            og.add_edges(g["src"][s], g["dst"][s], g["ts"][s], g["eid"][s])
        ref = O.OracleSampler(og, [10], "recent", threads=4)
        for it, (roots, rts) in enumerate(batches):
            mfgs = sampler.sample(roots, rts)
            b, wb = mfgs[0][0], ref.sample(roots, rts)[0][0]
            assert np.array_equal(b.srcdata["ID"].cpu().numpy(), wb.srcdata["ID"])
            assert np.array_equal(b.edata["ID"].cpu().numpy(), wb.edata["ID"])
            assert np.array_equal(b.edata["dt"].cpu().numpy(), wb.edata["dt"])
            cache.fetch_feature(mfgs, None)
            ids = b.srcdata["ID"].cpu().numpy()
            assert np.array_equal(b.srcdata["h"].cpu().numpy(), rows_of(ids))
            # TGN memory: gather for the block, then the batch's update (the model's new memory
            # stands in as a deterministic function of the step)
            mem.prepare_input(b)
            want = omem.prepare_input(ids)
            for key, w in zip(("mem", "mem_ts", "mail_ts", "mem_input"), want):
                assert np.array_equal(b.srcdata[key].cpu().numpy(), w), key
            R = b.num_dst_nodes()
            new_mem = (torch.arange(R * MG["d_mem"], device=dev, dtype=torch.float32)
                       .reshape(R, MG["d_mem"]) * (it + 1) % 97) / 97.0
            nid, ts_t = b.srcdata["ID"][:R], b.srcdata["ts"][:R]
            mem.update_mem_mail(nid, new_mem, ts_t, None, 1)
            omem.update_mem_mail(nid.cpu().numpy(), new_mem.cpu().numpy(), ts_t.cpu().numpy(),
                                 None, 1)
        torch.cuda.synchronize()
        touched = np.unique(np.concatenate([r for r, _ in batches]))
        assert np.array_equal(mem.node_memory[torch.from_numpy(touched).to(dev)].cpu().numpy(),
                              omem.node_memory[touched])
        assert np.array_equal(mem.mailbox[torch.from_numpy(touched).to(dev)].cpu().numpy(),
                              omem.mailbox[touched])
        assert np.array_equal(mem.node_memory_ts.cpu().numpy(), omem.node_memory_ts)
        assert node_shards.rows_pulled > 0
        ret[rank] = True
    except Exception as e:
        import traceback
        ret[rank] = "{}: {}\n{}".format(type(e).__name__, e, traceback.format_exc()[-1200:])
        raise
    finally:
        dist.destroy_process_group()


def test_config5_mag_shaped_tgn_memory_partitioned():
    _spawn(_config5_worker, 2)
