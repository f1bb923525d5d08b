"""Extended fuzz of the LRU replacement order against the oracle, in both forms (list / queue
with dead entries), beyond what the test suite's 16 seeds cover: long sequences on edge-only
caches with capacities from a few slots to 100 k, blocks from one row to several times the
capacity, id distributions that hit old entries, recent entries, nothing or everything.

  python scripts/fuzz_lru_forms.py [--seeds 40] [--steps 60]
Prints one line per seed and form; exits non-zero on the first mismatch.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Blk:
    def __init__(self, src, edge):
        self.srcdata = {"ID": src}
        self.edata = {"ID": edge}


def run(seed, steps, form):
    from gnnflow_amd.cache import LRUCache
    from oracle.cache_oracle import OracleLRUCache
    if form == "queue":
        os.environ["GNNFLOW_LRU_QUEUE_MIN_CAPACITY"] = "1"
    else:
        os.environ.pop("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", None)
    if form == "list2":      # list scan + list install (two launches) instead of the fused update
        os.environ["GNNFLOW_LRU_FUSED"] = "0"
    else:
        os.environ.pop("GNNFLOW_LRU_FUSED", None)
    rng = np.random.RandomState(9000 + seed)
    cap = int(rng.choice([3, 50, 700, 5000, 40000, 100000]))
    E = cap * int(rng.choice([2, 5, 20]))
    d = int(rng.choice([1, 4, 7, 16]))
    ef = rng.rand(E, d).astype(np.float32)
    ratio = cap / E
    hip = LRUCache(ratio, 0.0, 1, E, "cuda:0", None, torch.from_numpy(ef), 0, d)
    ora = OracleLRUCache(ratio, 0.0, 1, E, None, ef, 0, d, overflow_rule="first_seen")
    assert hip.edge_capacity == ora.edge_capacity
    hip.init_cache()
    ora.init_cache()
    none = np.zeros(0, np.int64)
    for step in range(steps):
        n = int(rng.choice([1, 7, cap // 8 + 1, cap // 3 + 1, cap, 3 * cap]))
        n = min(n, 250000)
        kind = rng.randint(5)
        cached = ora.edge.cached_ids()
        if kind == 0 or len(cached) == 0:      # anything
            ids = rng.randint(0, E, n)
        elif kind == 1:                        # mostly cached ids + a few misses
            ids = np.concatenate([rng.choice(cached, n), rng.randint(0, E, max(1, n // 50))])
        elif kind == 2:                        # a narrow window (many duplicates)
            lo = rng.randint(0, E)
            ids = rng.randint(lo, min(E, lo + max(2, n // 4)), n)
        elif kind == 3:                        # all misses if possible
            ids = rng.randint(0, E, n)
            ids = ids[~np.isin(ids, cached)]
            if len(ids) == 0:
                ids = rng.randint(0, E, n)
        else:                                  # the lowest cached ids (old entries) + one miss
            ids = np.concatenate([np.sort(cached)[:n], rng.randint(0, E, 1)])
        ids = np.ascontiguousarray(ids, np.int64)
        hb = [[Blk(torch.from_numpy(none).cuda(), torch.from_numpy(ids).cuda())]]
        ob = [[Blk(none, ids)]]
        hip.fetch_feature(hb, None, target_edge_features=False)
        ora.fetch_feature(ob, None)
        ok = np.array_equal(hb[0][0].edata["f"].cpu().numpy(), ef[ids])
        ok &= abs(float(hip.cache_edge_ratio) - ora.cache_edge_ratio) < 1e-6
        got = hip._edge.slot_ids()
        ok &= np.array_equal(np.sort(got[got >= 0]), ora.edge.cached_ids())
        if not ok:
            print("MISMATCH seed", seed, "form", form, "step", step, "cap", cap, "n", n, "kind", kind)
            sys.exit(1)
    st = hip._edge.lru_state()
    print("seed %3d %-5s cap %6d ok  (compactions %d, list-form updates %d, lone walks %d)" % (
        seed, form, cap, st["compactions"], st["list_form_updates"], st["lone_walks"]), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    for seed in range(args.seeds):
        for form in ("list", "list2", "queue"):
            run(seed, args.steps, form)
    from gnnflow_amd import _capi
    import ctypes
    v = ctypes.c_uint64()
    _capi.load().gf_debug_lru_recounts(ctypes.byref(v))
    print("all ok (fused-update granules recomputed by waiters: %d)" % v.value)


if __name__ == "__main__":
    main()
