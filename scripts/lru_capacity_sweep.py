"""How the LRU update scales with the CAPACITY of the cache (the list passes are O(capacity)):
one edge-feature cache, blocks of `--rows` ids drawn from a sliding window of recent ids (so a
steady fraction of each block hits), capacity swept from the config-2 size to GDELT scale.
Prints one JSON line per capacity: µs per fetch with and without the update.

  python scripts/lru_capacity_sweep.py [--rows 30000] [--dim 16]
  GNNFLOW_LRU_QUEUE_MIN_CAPACITY=4000000000 python scripts/lru_capacity_sweep.py   # list form only
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gnnflow_amd.cache import LRUCache  # noqa: E402
from gnnflow_amd.mfg import MFGBlock    # noqa: E402


def block_of(ids):
    n = int(ids.shape[0])
    z = torch.zeros(n, dtype=torch.int64, device=ids.device)
    b = MFGBlock(n + 1, 1, col=z, row=z, num_edges=n, device=ids.device)
    b.edata["ID"] = ids
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=30000)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--capacities", type=str, default="134489,1000000,4000000,16000000,40000000")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(3)
    for cap in [int(c) for c in args.capacities.split(",")]:
        num_ids = cap * 5
        feats = torch.empty(num_ids, args.dim, dtype=torch.float32, device=dev).uniform_()
        cache = LRUCache(0.2, 0.0, 1, num_ids, dev, None, feats, 0, args.dim)
        cache.init_cache()
        # ids: a window of 4 * rows "recent" ids sliding by rows / 2 per block
        blocks = []
        for i in range(args.iters + 5):
            lo = cap + i * (args.rows // 2)
            ids = rng.randint(lo, lo + 4 * args.rows, size=args.rows).astype(np.int64)
            blocks.append(torch.from_numpy(ids).to(dev))
        out = {}
        for upd in (True, False):
            for i in range(5):
                cache.fetch_feature([[block_of(blocks[i])]], update_cache=upd,
                                    target_edge_features=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5, 5 + args.iters):
                cache.fetch_feature([[block_of(blocks[i])]], update_cache=upd,
                                    target_edge_features=False)
            e1.record()
            torch.cuda.synchronize()
            out["update" if upd else "lookup_only"] = e0.elapsed_time(e1) * 1e3 / args.iters
        print(json.dumps({"capacity": cap, "rows": args.rows, "dim": args.dim,
                          "us_per_fetch_with_update": round(out["update"], 1),
                          "us_per_fetch_lookup_only": round(out["lookup_only"], 1),
                          "lru_form": "queue" if cache._edge.lru_state()["queue_form"] else "list",
                          "lru_state": cache._edge.lru_state()}), flush=True)
        del cache, feats
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
