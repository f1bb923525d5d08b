"""What slot capacity the REDDIT-shaped replay needs, per layer and world size (CPU only: the
oracle samples the whole stream; owners as gnnflow_amd.dist.owner_of_np): for every batch the
largest per-owner bucket of each layer's roots against the even share of the layer's WORST-CASE
root count the slots are sized from (stride = slack x bound / P).  A rank's batches in the
partitioned run are every P-th batch of the same stream, so the per-batch figures carry over.

    python scripts/slack_needed.py [--every 4]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gnnflow_amd import synthetic
from gnnflow_amd.dist import owner_of_np
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--every", type=int, default=4, help="use every n-th batch")
args = ap.parse_args()
g = synthetic.reddit_like(seed=42)
full = O.OracleGraph(minimum_block_size=62)
full.add_edges(g["src"], g["dst"], g["ts"], g["eid"], add_reverse=False)
F = [10, 10]
smp = O.OracleSampler(full, F, "recent", threads=8)
R0 = 1800
bound = [R0, R0 * (1 + F[0])]
need = {P: [0.0, 0.0] for P in (2, 4, 8)}       # largest bucket / (bound / P)
eneed = {P: [0.0, 0.0] for P in (2, 4, 8)}      # most edges of one owner / (bound / P x fanout)
fill = [0.0, 0.0]                                # largest root count / bound
for i, (r, t, _) in enumerate(synthetic.replay_batches(g, 600)):
    if i % args.every:
        continue
    m = smp.sample(r, t)
    roots = [np.asarray(r), np.asarray(m[-1][0].srcdata["ID"])]     # layer 0, layer 1
    blocks = [m[-1][0], m[0][0]]                                    # their sampled edges
    for l in range(2):
        fill[l] = max(fill[l], len(roots[l]) / bound[l])
        erow = np.asarray(blocks[l].edges()[1])                     # root index of every edge
        for P in need:
            own = owner_of_np(roots[l], P)
            b = np.bincount(own, minlength=P).max()
            need[P][l] = max(need[P][l], b / (bound[l] / P))
            e = np.bincount(own[erow], minlength=P).max() if len(erow) else 0
            eneed[P][l] = max(eneed[P][l], e / (bound[l] / P * F[l]))
print(json.dumps({"what": "largest per-owner bucket over the replay, in units of the even share of "
                          "the layer's worst-case root count (= the smallest slack without overflow)",
                  "batches_used": "every {}th of 1121".format(args.every),
                  "largest_layer_over_worst_case": {"layer0": round(fill[0], 3), "layer1": round(fill[1], 3)},
                  "slack_needed": {"P={}".format(P): {"layer0": round(v[0], 3), "layer1": round(v[1], 3)}
                                   for P, v in need.items()},
                  "edges_of_the_fullest_owner": {
                      "what": "most sampled edges any one owner returns for a layer, in units of "
                              "(even share of the worst-case roots) x fanout; divided by the slack "
                              "it is the share of a reply slot's fixed records that hold an edge "
                              "(the compact slots' edge_fill must cover it)",
                      **{"P={}".format(P): {"layer0": round(v[0], 3), "layer1": round(v[1], 3)}
                         for P, v in eneed.items()}}}))
