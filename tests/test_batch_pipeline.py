"""Batch pipeline parity: gnnflow_amd.utils / gnnflow_amd.data against the
outputs of the reference's own gnnflow/utils.py + gnnflow/data.py recorded in
tests/golden/batch_reference.npz (tests/golden/make_batch_fixtures.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, SequentialSampler

from gnnflow_amd import data as D
from gnnflow_amd import utils as U

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_reference.npz")


@pytest.fixture(scope="module")
def ref(tmp_path_factory):
    import pandas as pd
    z = np.load(FIX)
    tmp = tmp_path_factory.mktemp("data")
    os.makedirs(os.path.join(tmp, "TOY"))
    pd.DataFrame({"src": z["src"], "dst": z["dst"], "time": z["time"],
                  "ext_roll": z["ext_roll"]}).to_csv(os.path.join(tmp, "TOY", "edges.csv"))
    frames = U.load_dataset("TOY", data_dir=str(tmp))
    return z, frames


def check(z, prefix, batches):
    batches = list(batches)
    assert len(batches) == int(z[prefix + "/n"][0]), prefix
    for i, (roots, ts, eid) in enumerate(batches):
        for name, got in (("roots", roots), ("ts", ts), ("eid", np.asarray(eid))):
            want = z["{}/{}/{}".format(prefix, i, name)]
            assert got.dtype == want.dtype and np.array_equal(got, want), (prefix, i, name)


def test_load_dataset_split_and_columns(ref):
    z, (train, val, test, full) = ref
    assert [len(train), len(val), len(test), len(full)] == list(z["split"])
    assert list(full.columns) == [str(c) for c in z["columns"]]
    assert val.index[0] == len(train)          # slices keep the file's row index
    with pytest.raises(ValueError):
        U.load_dataset("MISSING", data_dir="/nonexistent")


@pytest.mark.parametrize("part", ["train", "val", "test"])
def test_get_batch_matches_reference(ref, part):
    z, (train, val, test, full) = ref
    df = {"train": train, "val": val, "test": test}[part]
    neg = U.DstRandEdgeSampler(full["dst"].to_numpy(dtype=np.int32), seed=7)
    check(z, "get_batch/" + part, U.get_batch(df, 600, 0, neg))


def test_get_batch_no_neg_matches_reference(ref):
    z, (train, val, test, full) = ref
    check(z, "get_batch_no_neg/val", U.get_batch_no_neg(val, 256))


@pytest.mark.parametrize("chunks", [1, 8])
def test_dataloader_path_matches_reference(ref, chunks):
    z, (train, val, test, full) = ref
    torch.manual_seed(3)
    neg = U.DstRandEdgeSampler(train["dst"].to_numpy(dtype=np.int32), seed=11)
    ds = D.EdgePredictionDataset(train, neg)
    sampler = D.RandomStartBatchSampler(SequentialSampler(ds), batch_size=600,
                                        drop_last=False, num_chunks=chunks)
    for epoch in range(2):
        loader = DataLoader(ds, sampler=sampler, collate_fn=D.default_collate_ndarray,
                            num_workers=0)
        check(z, "loader/chunks{}/epoch{}".format(chunks, epoch), [tuple(b) for b in loader])


def test_dataset_without_negatives_and_length(ref):
    z, (train, val, test, full) = ref
    ds = D.EdgePredictionDataset(val, None)
    check(z, "dataset_no_neg", [ds[list(range(5, 40))]])
    assert [ds.length, len(ds)] == list(z["dataset_length"])


def test_default_collate_ndarray(ref):
    z, _ = ref
    c = D.default_collate_ndarray
    assert np.array_equal(c([np.arange(3), np.arange(3) + 10]), z["collate/arrays"])
    assert np.array_equal(c([0, 1, 2, 3]), z["collate/ints"])
    got = c([0.5, 1.5])
    assert got.dtype == z["collate/floats"].dtype and np.array_equal(got, z["collate/floats"])
    assert np.array_equal(c([np.float32(1.0), np.float32(2.0)]), z["collate/np_scalars"])
    m = c([{"A": 0, "B": 1}, {"A": 100, "B": 100}])
    assert np.array_equal(m["A"], z["collate/map_A"]) and np.array_equal(m["B"], z["collate/map_B"])
    t = c([(0, 1), (2, 3)])
    assert np.array_equal(t[0], z["collate/tuple_0"]) and np.array_equal(t[1], z["collate/tuple_1"])
    assert c(["a", "b"]) == ["a", "b"]
    with pytest.raises(RuntimeError):
        c([[1, 2], [1]])
    with pytest.raises(TypeError):
        c([np.array(["x"])])


def test_negative_samplers(ref):
    z, _ = ref
    s = U.RandEdgeSampler(z["src"], z["dst"], seed=5)
    a, b = s.sample(20)
    assert np.array_equal(a, z["rand_edge/src"]) and np.array_equal(b, z["rand_edge/dst"])
    d = U.DstRandEdgeSampler(z["dst"][:100], seed=5)
    assert np.array_equal(d.sample(10), z["dst_rand/first"])
    d.add_dst_list(z["dst"][100:200])
    d.reset_random_state()
    assert np.array_equal(d.sample(10), z["dst_rand/after_add"])


def test_random_start_keeps_every_row_once(ref):
    """Size-independent property: whatever the random start, an epoch covers every row
    exactly once, in order, and all batches after the first are full."""
    z, (train, _, _, _) = ref
    ds = D.EdgePredictionDataset(train, None)
    for seed in range(5):
        torch.manual_seed(seed)
        sampler = D.RandomStartBatchSampler(SequentialSampler(ds), 100, False, num_chunks=4)
        batches = list(sampler)
        assert sum(batches, []) == list(range(len(ds)))
        assert all(len(b) == 100 for b in batches[1:-1])
        assert len(batches[0]) in (25, 50, 75, 100)


def test_load_feat(tmp_path):
    root = tmp_path / "TOY"
    root.mkdir()
    with pytest.raises(ValueError):
        U.load_feat("TOY", data_dir=str(tmp_path))
    ef = np.random.RandomState(0).rand(7, 3).astype(np.float32)
    np.save(root / "edge_features.npy", ef)
    node, edge = U.load_feat("TOY", data_dir=str(tmp_path))
    assert node is None and torch.equal(edge, torch.from_numpy(ef))
    node, edge = U.load_feat("TOY", data_dir=str(tmp_path), load_edge=False)
    assert node is None and edge is None
    _, mm = U.load_feat("TOY", data_dir=str(tmp_path), memmap=True)
    assert isinstance(mm, np.memmap) and np.array_equal(mm, ef)


def _shared_feat_worker(rank, world, port, root, ret):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnflow_amd import utils as U
        node, edge = U.load_feat("SHM", data_dir=root, shared_memory=True, local_rank=rank,
                                 local_world_size=world)
        want = np.load(os.path.join(root, "SHM", "edge_features.npy"))
        ok = node is None and edge.dtype == torch.float32 and \
            np.array_equal(edge.numpy(), want.astype(np.float32))
        dist.barrier()
        if rank == 1:                 # the ranks map the SAME pages
            edge[0, 0] = 1234.5
        dist.barrier()
        ok &= float(edge[0, 0]) == 1234.5
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_load_feat_shared_memory_across_local_ranks(tmp_path):
    """gnnflow/utils.py:289-338: rank 0 reads and publishes, the others map the same pages."""
    import torch.multiprocessing as mp
    d = tmp_path / "SHM"
    d.mkdir()
    np.save(d / "edge_features.npy", np.random.RandomState(3).rand(50, 6))   # float64 on disk
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    try:
        mp.spawn(_shared_feat_worker, args=(2, port, str(tmp_path), ret), nprocs=2, join=True)
    finally:
        for name in ("node_feats", "edge_feats"):
            p = "/dev/shm/gnnflow_amd_" + name
            if os.path.exists(p):
                os.remove(p)
    assert dict(ret) == {0: True, 1: True}


# ---- world_size 2 (gloo): the samplers that talk to the process group ---------------------
def _sampler_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)       # rank 0's draw must win on every rank
        n = 1000
        ds = list(range(n))
        out = {}
        s = D.DistributedBatchSampler(SequentialSampler(ds), batch_size=64, drop_last=False,
                                      rank=rank, world_size=world, num_chunks=4)
        out["dist"] = [list(s) for _ in range(3)]
        out["dist_first"] = [s.random_size]
        r = D.RandomStartBatchSampler(SequentialSampler(ds), batch_size=64, drop_last=False,
                                      num_chunks=4, world_size=world)
        out["rand"] = [list(r) for _ in range(3)]
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_distributed_samplers_gloo_world2():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_sampler_worker, args=(2, port, ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    for epoch in range(3):
        # DistributedBatchSampler: rank r gets idx % 2 == r, all of them, in order; both
        # ranks cut their first batch at the same (rank 0's) random size
        fa, fb = sum(a["dist"][epoch], []), sum(b["dist"][epoch], [])
        assert fa == list(range(0, 1000, 2)) and fb == list(range(1, 1000, 2))
        assert len(a["dist"][epoch][0]) == len(b["dist"][epoch][0])
        assert len(a["dist"][epoch][0]) in (16, 32, 48, 64)
        assert all(len(x) == 64 for x in a["dist"][epoch][1:-1])
        # RandomStartBatchSampler(world_size=2): identical batches on both ranks
        assert a["rand"][epoch] == b["rand"][epoch]
        assert sum(a["rand"][epoch], []) == list(range(1000))


def test_early_stop_monitor():
    m = U.EarlyStopMonitor(max_round=2)
    assert [m.early_stop_check(v) for v in (0.5, 0.6, 0.6, 0.59)] == [False, False, False, True]
    assert m.best_epoch == 1 and m.epoch_count == 4
    low = U.EarlyStopMonitor(max_round=1, higher_better=False)
    assert [low.early_stop_check(v) for v in (1.0, 0.5, 0.7)] == [False, False, True]


def test_small_helpers(tmp_path):
    import pandas as pd
    os.makedirs(tmp_path / "TOY")
    pd.DataFrame({"src": [0, 1, 2], "dst": [3, 4, 5], "time": [0.1, 0.2, 0.3],
                  "ext_roll": [0, 1, 2]}).to_csv(tmp_path / "TOY" / "edges.csv")
    chunks = list(U.load_dataset_in_chunks("TOY", data_dir=str(tmp_path), chunksize=2))
    assert [len(c) for c in chunks] == [2, 1]
    assert set(chunks[0].columns) == {"src", "dst", "time", "Unnamed: 0", "ext_roll"}
    with pytest.raises(ValueError):
        U.load_dataset_in_chunks("NOPE", data_dir=str(tmp_path))
    assert U.get_pinned_buffers([10, 10], 1, 600, 172, 172) == ([], [])
    assert os.path.isdir(os.path.join(U.get_project_root_dir(), "gnnflow_amd"))
    assert U.get_node_feats() is None
