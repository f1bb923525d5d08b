"""Owner-sharded feature tables (SURVEY.md 8(e) "Features"; reference: KVStore pull of the
missed rows from the owning machine, gnnflow/cache/cache.py:293-312,351-388,
gnnflow/distributed/kvstore.py:70-126).

CPU: gnnflow_amd.dist.FeatureShards.pull over gloo, world sizes 2 and 3 — the rows come
back in request order and equal the full table's, empty requests take part in the
collectives.  GPU (tests/test_gpu_dist_features.py): Cache(distributed=True).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnflow_amd.dist import FeatureShards, owner_of_np
        rng = np.random.RandomState(0)          # same tables on every rank
        N, E, d = 500, 4000, 12
        nfeat = rng.rand(N, d).astype(np.float32)
        efeat = rng.rand(E, d).astype(np.float32)
        esrc = rng.randint(0, N, E).astype(np.int64)       # source node of every edge
        node = FeatureShards.from_full(nfeat, np.arange(N), rank, world, "cpu")
        edge = FeatureShards.from_full(efeat, esrc, rank, world, "cpu")
        ok = True
        # every rank holds exactly the rows it owns
        ok &= bool((owner_of_np(node.local_ids.numpy(), world) == rank).all())
        ok &= bool((owner_of_np(esrc[edge.local_ids.numpy()], world) == rank).all())
        r2 = np.random.RandomState(100 + rank)  # different requests per rank
        for it, n in enumerate([0, 1, 37, 900, 0, 250]):
            n = [0, 1, 37, 900, 0, 250][(it + rank) % 6]   # uneven, sometimes empty
            ids = r2.randint(0, N, n).astype(np.int64)
            got = node.pull(torch.from_numpy(ids), torch.from_numpy(ids))
            ok &= np.array_equal(got.numpy(), nfeat[ids])
            eids = r2.randint(0, E, n).astype(np.int64)
            got = edge.pull(torch.from_numpy(eids), torch.from_numpy(esrc[eids]))
            ok &= np.array_equal(got.numpy(), efeat[eids])
        ok &= node.rows_pulled > 0 and node.bytes_sent > 0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_feature_shards_pull_over_gloo(world):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _transport_worker(rank, world, port, ret):
    """ShardedFeatures' exchanges of a natively planned pull round, on CPU tensors over gloo:
    counts [nctx, P] -> what every rank asks THIS rank for; one variable-size exchange per
    context, rank-major on the receiving side."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnflow_amd.dist import FeatureShards, ShardedFeatures
        feats = np.arange(40, dtype=np.float32).reshape(10, 4)
        sh = ShardedFeatures(node=FeatureShards.from_full(feats, np.arange(10), rank, world, "cpu"))
        nctx = 3
        # rank r sends (r + 2 q + k) % 5 rows of context k to rank q
        sc = [[(rank + 2 * q + k) % 5 for q in range(world)] for k in range(nctx)]
        counts = torch.tensor(sc, dtype=torch.int32)
        got = sh.exchange_counts(counts)
        rc = [[(q + 2 * rank + k) % 5 for q in range(world)] for k in range(nctx)]
        ok = got.tolist() == rc
        # rows: context k's row for (sender s, receiver q, index i) carries 1000 s + 100 q + 10 k + i
        send, recv = [], []
        for k in range(nctx):
            rows = [[1000 * rank + 100 * q + 10 * k + i] * (k + 1)
                    for q in range(world) for i in range(sc[k][q])]
            send.append(torch.tensor(rows, dtype=torch.int64).reshape(-1, k + 1))
            recv.append(torch.empty((sum(rc[k]), k + 1), dtype=torch.int64))
        sh.exchange_segments(send, sc, recv, rc)
        for k in range(nctx):
            want = [[1000 * s + 100 * rank + 10 * k + i] * (k + 1)
                    for s in range(world) for i in range(rc[k][s])]
            ok &= recv[k].tolist() == want
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_pull_round_exchanges_over_gloo(world):
    ret = mp.Manager().dict()
    mp.spawn(_transport_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_feature_shards_single_process():
    from gnnflow_amd.dist import FeatureShards
    rng = np.random.RandomState(1)
    feats = rng.rand(50, 4).astype(np.float32)
    sh = FeatureShards.from_full(feats, np.arange(50), 0, 1, "cpu")
    ids = torch.tensor([3, 3, 49, 0])
    assert np.array_equal(sh.pull(ids, ids).numpy(), feats[[3, 3, 49, 0]])
    assert sh.pull(ids[:0], ids[:0]).shape == (0, 4)


def _worker_one(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnflow_amd.dist import FeatureShards
        rng = np.random.RandomState(2)
        feats = rng.rand(300, 5).astype(np.float32)
        sh = FeatureShards.from_full(feats, np.arange(300), 0, 1, "cpu")
        sh.always_exchange = True     # the collective protocol with one rank: rows travel to itself
        ok = True
        for n in (0, 1, 64, 1000):
            ids = rng.randint(0, 300, n).astype(np.int64)
            ok &= np.array_equal(sh.pull(torch.from_numpy(ids), torch.from_numpy(ids)).numpy(),
                                 feats[ids])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_feature_shards_forced_exchange_with_one_rank():
    ret = mp.Manager().dict()
    mp.spawn(_worker_one, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert dict(ret) == {0: True}
