"""
Generates tests/golden/cache_reference.npz by running the REFERENCE's own Python feature
cache (gnnflow/cache/cache.py + lru_cache.py, imported read-only from /root/reference)
on seeded inputs.  Runs only in the build container; the .npz (inputs + the reference's
outputs) is what travels.

The reference package needs `dgl` and its native `libgnnflow`, which are not installed:
empty stub modules satisfy the imports (only type names are taken from them; the cache
code itself is pure torch and runs unmodified).  device='cpu:0' passes the reference's
"Cache must be on GPU" guard (cache.py:49-50) so its torch ops run on the CPU.

Run:  python -B tests/golden/make_cache_fixtures.py
"""
import os
import sys
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


def import_reference_cache():
    import torch  # noqa: F401
    dgl = _stub("dgl")
    dgl.heterograph = _stub("dgl.heterograph", DGLBlock=type("DGLBlock", (), {}))
    dgl.utils = _stub("dgl.utils")
    dgl.utils.shared_mem = _stub("dgl.utils.shared_mem",
                                 create_shared_mem_array=None, get_shared_mem_array=None)
    dgl.function = _stub("dgl.function")
    _stub("libgnnflow", InsertionPolicy=None, MemoryResourceType=None, _DynamicGraph=None,
          SamplingPolicy=None, SamplingResult=None, _TemporalSampler=None, KVStore=None)
    # the package __init__ pulls in data/utils modules that need more; load the two
    # cache files directly under the package name instead
    import importlib.util
    pkg = _stub("gnnflow")
    pkg.__path__ = [os.path.join(REF, "gnnflow")]
    dist = _stub("gnnflow.distributed")
    dist.__path__ = []
    _stub("gnnflow.distributed.kvstore", KVStoreClient=type("KVStoreClient", (), {}))
    cpkg = _stub("gnnflow.cache")
    cpkg.__path__ = [os.path.join(REF, "gnnflow", "cache")]
    mods = {}
    for name in ("cache", "lru_cache", "lfu_cache", "fifo_cache"):
        spec = importlib.util.spec_from_file_location(
            "gnnflow.cache." + name, os.path.join(REF, "gnnflow", "cache", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["gnnflow.cache." + name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return {"lru": mods["lru_cache"].LRUCache, "lfu": mods["lfu_cache"].LFUCache,
            "fifo": mods["fifo_cache"].FIFOCache}


class FakeBlock:
    def __init__(self, src_ids, edge_ids):
        import torch
        self.srcdata = {"ID": torch.from_numpy(np.asarray(src_ids, np.int64))}
        self.edata = {"ID": torch.from_numpy(np.asarray(edge_ids, np.int64))}


def main():
    import torch
    classes = import_reference_cache()
    rng = np.random.RandomState(1234)
    out = {}
    scenarios = [
        # name, num_nodes, num_edges, d_n, d_e, ratio, nbatches, nsrc, nedge, skew
        ("small", 200, 1000, 8, 12, 0.2, 6, 60, 150, 1.2),
        ("tiny_overflow", 20, 50, 4, 4, 0.2, 4, 15, 30, 0.0),
        ("edge_only", 100, 1200, 0, 172, 0.2, 5, 0, 300, 1.5),
        ("bool_feats", 64, 256, 6, 6, 0.5, 3, 40, 80, 0.5),
        # capacity 1: torch.topk never has a tie to break -> whole ratio sequence pinned
        ("cap1_tie_free", 10, 10, 4, 4, 0.1, 10, 4, 5, 0.0),
        # other policies (name prefix selects the reference class)
        ("fifo_small", 200, 1000, 8, 12, 0.2, 8, 60, 150, 1.2),
        ("fifo_overflow_wrap", 30, 60, 4, 4, 0.3, 8, 25, 40, 0.0),
        ("lfu_small", 200, 1000, 8, 12, 0.2, 6, 60, 150, 1.2),
        ("lfu_cap1_tie_free", 10, 10, 4, 4, 0.1, 10, 4, 5, 0.0),
    ]
    names = []

    def put_rows(key, arr, batch):
        """Reference output rows: sha256 of the float32 bytes for every block, the full
        array for the first batch only (keeps the fixture small)."""
        import hashlib
        arr = np.ascontiguousarray(arr, np.float32)
        out[key + "/sha256"] = np.frombuffer(hashlib.sha256(arr.tobytes()).digest(), np.uint8)
        out[key + "/shape"] = np.array(arr.shape, np.int64)
        if batch == 0:
            out[key] = arr

    for (name, N, E, dn, de, ratio, nb, nsrc, nedge, skew) in scenarios:
        # multiples of 1/256: exact in float32 and compressible
        nf = (rng.randint(0, 256, (N, dn)) / 256.0).astype(np.float32) if dn else None
        ef = (rng.randint(0, 256, (E, de)) / 256.0).astype(np.float32) if de else None
        if name == "bool_feats":
            nf = (nf > 0.5)
            ef = (ef > 0.5)
        policy = name.split("_")[0] if name.split("_")[0] in ("fifo", "lfu") else "lru"
        out[name + "/policy"] = np.array([policy])
        # FIFOCache.reset() only rewinds the edge pointer: exercised mid-replay.  (The LRU /
        # LFU reset() re-fills the slots without clearing the old id->slot map, so ids
        # cached before it read stale rows afterwards; not recorded as golden behaviour.)
        reset_after = 3 if name == "fifo_overflow_wrap" else -1
        out[name + "/reset_after"] = np.array([reset_after], np.int64)
        cache = classes[policy](ratio, ratio, N, E, torch.device("cpu:0"),
                         None if nf is None else torch.from_numpy(nf),
                         None if ef is None else torch.from_numpy(ef), dn, de)
        cache.init_cache()
        names.append(name)
        out[name + "/meta"] = np.array([N, E, dn, de, nb], np.int64)
        out[name + "/ratio"] = np.array([ratio], np.float64)
        if nf is not None:
            out[name + "/node_feats"] = nf
        if ef is not None:
            out[name + "/edge_feats"] = ef

        def draw(high, n):
            if skew <= 0:
                return rng.randint(0, high, n)
            p = np.arange(1, high + 1, dtype=np.float64) ** (-skew)
            p /= p.sum()
            return rng.choice(high, size=n, p=p)

        for b in range(nb):
            # two layers x one snapshot: mfgs[0] is the outer (largest) layer
            blocks = [[FakeBlock(draw(N, max(nsrc, 1)), draw(E, nedge))],
                      [FakeBlock(draw(N, max(nsrc // 4, 1)), draw(E, max(nedge // 4, 1)))]]
            eid = draw(E, 10)
            for li in range(2):
                out["{}/b{}/src_ids{}".format(name, b, li)] = blocks[li][0].srcdata["ID"].numpy()
                out["{}/b{}/edge_ids{}".format(name, b, li)] = blocks[li][0].edata["ID"].numpy()
            out["{}/b{}/eid".format(name, b)] = eid.astype(np.int64)
            cache.fetch_feature(blocks, eid)
            if dn:
                put_rows("{}/b{}/h".format(name, b), blocks[0][0].srcdata["h"].numpy(), b)
                out["{}/b{}/node_ratio".format(name, b)] = np.array(
                    [float(cache.cache_node_ratio)], np.float64)
                # state after the update: which ids are cached (order-free)
                out["{}/b{}/node_cached".format(name, b)] = np.sort(
                    torch.nonzero(cache.cache_node_flag).flatten().numpy())
            if de:
                for li in range(2):
                    put_rows("{}/b{}/f{}".format(name, b, li),
                             blocks[li][0].edata["f"].numpy(), b)
                out["{}/b{}/edge_ratio".format(name, b)] = np.array(
                    [float(cache.cache_edge_ratio)], np.float64)
                out["{}/b{}/edge_cached".format(name, b)] = np.sort(
                    torch.nonzero(cache.cache_edge_flag).flatten().numpy())
                put_rows("{}/b{}/target".format(name, b),
                         cache.target_edge_features.numpy(), b)
            if b == reset_after:
                cache.reset()
    out["scenarios"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
