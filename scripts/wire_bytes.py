"""Bytes one rank puts on the xGMI links per sample() of the headline batch (600 edges = 1 800
roots, fanout [10, 10]) through the slotted exchange of the partitioned sampler, by world size:
request slots + reply slots to each of the P - 1 peers, with the fixed-fanout reply records of
round 4 and with the compact reply slots of round 5 (offsets + packed edges, edge_fill 0.1).
The slot sizes are what the native layout (gf_sampler_part_layout_slotted /
Sampler::group_layout) gives for the default slack of every world size; nothing is sampled.

    python scripts/wire_bytes.py            (needs a GPU only to create the sampler handles)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnflow_amd                                              # noqa: E402
from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm   # noqa: E402

dev = torch.device("cuda", 0)
g = gnnflow_amd.DynamicGraph(1 << 20, 1 << 28, "cuda", 62, 64, "insert")
g.add_edges(np.array([0, 1]), np.array([1, 2]), np.array([1.0, 2.0], np.float32))
out = {"batch": "600 edges = 1800 roots, fanouts [10, 10], 12-byte reply records, shared chains",
       "by_world_size": {}}
for P in (2, 4, 8):
    comms = NativeComm.loopback(P, dev)
    row = {}
    for name, fill, chain in (("fixed_reply_slots_round4", 0.0, 4), ("compact_reply_slots", 0.1, 4),
                              # round 6: a chain of ONE sample (rung "hash-simple" of bench.py's
                              # ladder) ships compact replies too; before, the fixed records
                              ("single_chain_compact_round6", 0.1, 1)):
        part = DevicePartitionedSampler(gnnflow_amd.TemporalSampler(g, [10, 10], "recent"),
                                        comm=comms[0], slot_roots=1800, narrow_ids=True,
                                        edge_fill=fill, chain_samples=chain)
        w = part.wire_bytes_per_sample()
        row[name] = {"slack": w["slack"], "edge_fill": w["reply_edge_fill"],
                     "request_KB_to_each_peer": round(w["request_bytes_to_each_peer"] / 1e3, 1),
                     "reply_KB_to_each_peer": round(w["reply_bytes_to_each_peer"] / 1e3, 1),
                     "MB_on_links_per_rank_and_sample": round(w["bytes_on_links_per_rank"] / 1e6, 3)}
    row["reduction"] = round(row["fixed_reply_slots_round4"]["MB_on_links_per_rank_and_sample"] /
                             row["compact_reply_slots"]["MB_on_links_per_rank_and_sample"], 2)
    out["by_world_size"]["P={}".format(P)] = row
    for c in comms:
        c.close()
print(json.dumps(out, indent=1))
