"""
Generates tests/golden/batch_reference.npz by running the REFERENCE's batch pipeline
(gnnflow/utils.py get_batch / get_batch_no_neg / DstRandEdgeSampler / load_dataset and
gnnflow/data.py EdgePredictionDataset / RandomStartBatchSampler / default_collate_ndarray,
imported read-only from /root/reference) on a seeded synthetic edge list.  Runs only in the
build container; the .npz (inputs + the reference's outputs) is what travels.

Stubs: `dgl`, `libgnnflow` (type names only) and `torch._six` (removed from modern torch;
only `string_classes` is taken from it).

Run:  python -B tests/golden/make_batch_fixtures.py
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    import torch  # noqa: F401
    dgl = _stub("dgl")
    dgl.heterograph = _stub("dgl.heterograph", DGLBlock=type("DGLBlock", (), {}))
    dgl.utils = _stub("dgl.utils")
    dgl.utils.shared_mem = _stub("dgl.utils.shared_mem",
                                 create_shared_mem_array=None, get_shared_mem_array=None)
    _stub("libgnnflow", InsertionPolicy=None, MemoryResourceType=None, _DynamicGraph=None,
          SamplingPolicy=None, SamplingResult=None, _TemporalSampler=None, KVStore=None)
    _stub("torch._six", string_classes=(str, bytes))
    pkg = _stub("gnnflow")
    pkg.__path__ = [os.path.join(REF, "gnnflow")]
    mods = {}
    for name in ("dynamic_graph", "utils", "data"):
        spec = importlib.util.spec_from_file_location(
            "gnnflow." + name, os.path.join(REF, "gnnflow", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["gnnflow." + name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["utils"], mods["data"]


def main():
    import pandas as pd
    import torch
    from torch.utils.data import DataLoader, SequentialSampler
    U, D = import_reference()
    rng = np.random.RandomState(99)
    n = 2500
    src = rng.randint(0, 300, n)
    dst = rng.randint(300, 500, n)
    time = np.sort(rng.rand(n) * 1000).astype(np.float64)
    roll = np.zeros(n, np.int64)
    roll[1700:2100] = 1
    roll[2100:] = 2
    out = {"src": src, "dst": dst, "time": time, "ext_roll": roll}

    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "TOY"))
        pd.DataFrame({"src": src, "dst": dst, "time": time, "ext_roll": roll}).to_csv(
            os.path.join(tmp, "TOY", "edges.csv"))
        train, val, test, full = U.load_dataset("TOY", data_dir=tmp)
    out["split"] = np.array([len(train), len(val), len(test), len(full)], np.int64)
    out["columns"] = np.array(list(full.columns))

    def put(prefix, batches):
        out[prefix + "/n"] = np.array([len(batches)], np.int64)
        for i, (roots, ts, eid) in enumerate(batches):
            out["{}/{}/roots".format(prefix, i)] = roots
            out["{}/{}/ts".format(prefix, i)] = ts
            out["{}/{}/eid".format(prefix, i)] = np.asarray(eid)

    # get_batch without a random start (num_chunks = 0) on the three slices
    for name, df in (("train", train), ("val", val), ("test", test)):
        neg = U.DstRandEdgeSampler(full["dst"].to_numpy(dtype=np.int32), seed=7)
        put("get_batch/" + name, list(U.get_batch(df, 600, 0, neg)))
    put("get_batch_no_neg/val", list(U.get_batch_no_neg(val, 256)))

    # DataLoader path: EdgePredictionDataset + RandomStartBatchSampler, CPU draw
    for chunks in (1, 8):
        torch.manual_seed(3)
        neg = U.DstRandEdgeSampler(train["dst"].to_numpy(dtype=np.int32), seed=11)
        ds = D.EdgePredictionDataset(train, neg)
        sampler = D.RandomStartBatchSampler(SequentialSampler(ds), batch_size=600,
                                            drop_last=False, num_chunks=chunks)
        for epoch in range(2):
            loader = DataLoader(ds, sampler=sampler, collate_fn=D.default_collate_ndarray,
                                num_workers=0)
            put("loader/chunks{}/epoch{}".format(chunks, epoch), [tuple(b) for b in loader])
    # dataset without a negative sampler
    ds = D.EdgePredictionDataset(val, None)
    put("dataset_no_neg", [ds[list(range(5, 40))]])
    out["dataset_length"] = np.array([ds.length, len(ds)], np.int64)

    # default_collate_ndarray on the input kinds it documents
    c = D.default_collate_ndarray
    out["collate/arrays"] = c([np.arange(3), np.arange(3) + 10])
    out["collate/ints"] = c([0, 1, 2, 3])
    out["collate/floats"] = c([0.5, 1.5])
    out["collate/np_scalars"] = c([np.float32(1.0), np.float32(2.0)])
    m = c([{"A": 0, "B": 1}, {"A": 100, "B": 100}])
    out["collate/map_A"], out["collate/map_B"] = m["A"], m["B"]
    t = c([(0, 1), (2, 3)])
    out["collate/tuple_0"], out["collate/tuple_1"] = t[0], t[1]

    # negative samplers
    s = U.RandEdgeSampler(src, dst, seed=5)
    a, b = s.sample(20)
    out["rand_edge/src"], out["rand_edge/dst"] = a, b
    d = U.DstRandEdgeSampler(dst[:100], seed=5)
    first = d.sample(10)
    d.add_dst_list(dst[100:200])
    d.reset_random_state()
    out["dst_rand/first"], out["dst_rand/after_add"] = first, d.sample(10)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "batch_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
