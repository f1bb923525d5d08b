"""Kernel-level account of the config-5 stand-in epoch (examples/tgn_epoch.py) from a rocprofv3
--kernel-trace --stats run of it: GPU time by family — this repo's kernels (sampler, gather, LRU
update, TGN memory ops, block message passing = SURVEY 8(a) / 8(f)-1 / 8(f)-2) against the torch /
rocBLAS kernels of the model around them (out of scope) — and algorithmic bytes / average
duration of the block ops at the epoch's shapes.
    python scripts/tgn_epoch_kernel_account.py <kernel_stats.csv> <epoch.json> > summary.json
Reference: gnnflow/models/modules/layers.py:144-159 (edge_softmax + update_all),
gnnflow/models/modules/memory.py:156-269."""
import csv
import json
import re
import sys

stats, epoch = sys.argv[1], json.load(open(sys.argv[2]))
FAMILIES = [
    ("sampler", r"sample_\w+_kernel|philox"),
    ("feature gather", r"gather_rows\w*_kernel"),
    ("LRU update", r"lru_\w+_kernel|cache_\w+_kernel|list_fill"),
    ("TGN memory ops", r"memory_\w+_kernel"),
    ("block ops: edge_softmax", r"edge_softmax_\w+"),
    ("block ops: segment reduce", r"segment_(reduce|max)_\w+|segment_offsets"),
    ("edge store / ingest", r"move_segments|scatter_\w+|ingest|rocprim|node_table|publish"),
]
rows = list(csv.DictReader(open(stats)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
fam = {name: {"us": 0.0, "calls": 0, "kernels": {}} for name, _ in FAMILIES}
other = {"us": 0.0, "calls": 0, "kernels": {}}
for r in rows:
    name = r["Name"]
    ours = "gf::" in name
    short = re.sub(r"\(.*", "", name.replace("gf::(anonymous namespace)::", "").replace("void ", ""))
    us, calls = float(r["TotalDurationNs"]) / 1e3, int(r["Calls"])
    target = other
    if ours:
        for fname, pat in FAMILIES:
            if re.search(pat, short):
                target = fam[fname]
                break
    target["us"] += us
    target["calls"] += calls
    target["kernels"][short[:60]] = {"calls": calls, "avg_us": us / max(calls, 1),
                                     "total_ms": us / 1e3}
ep = epoch["epochs"][-1]
batches = sum(e["batches"] for e in epoch["epochs"])
E = ep["sampled_edges"] / ep["batches"]         # edges per block
R, H, D = 3 * 4000, 2, 100
# algorithmic bytes per launch (f32): edge_softmax fwd reads the scores and writes the
# weights [E, H]; bwd reads weights + grad and writes grad; segment reduce fwd reads [E, D]
# messages and writes [R, D]; bwd reads [R, D] and writes [E, D]; + 8 B offsets per destination
ALG = {"edge_softmax_fwd": 8 * E * H + 8 * R, "edge_softmax_bwd": 12 * E * H + 8 * R,
       "segment_reduce_fwd": 4 * (E + R) * D + 8 * R, "segment_reduce_bwd": 4 * (E + R) * D + 8 * R}
block = {}
for fname in ("block ops: edge_softmax", "block ops: segment reduce"):
    for k, v in fam[fname]["kernels"].items():
        for key, nbytes in ALG.items():
            if k.startswith(key):
                block[k] = {"calls": v["calls"], "avg_us": v["avg_us"],
                            "algorithmic_MB": nbytes / 1e6,
                            "GBps": nbytes / (v["avg_us"] * 1e-6) / 1e9,
                            "frac_of_8TBps": nbytes / (v["avg_us"] * 1e-6) / 8e12}
ours_us = sum(f["us"] for f in fam.values())
out = {
    "what": "GPU time of examples/tgn_epoch.py by kernel family (rocprofv3 --kernel-trace --stats)",
    "epoch": {k: ep[k] for k in ("seconds", "batches", "ms_per_batch", "host_seconds_by_stage")},
    "batches_profiled": batches,
    "gpu_busy_ms_total": total / 1e6,
    "gpu_busy_ms_per_batch": total / 1e6 / max(batches, 1),
    "this_repos_kernels": {"share_of_gpu_time": ours_us * 1e3 / total,
                           "ms_per_batch": ours_us / 1e3 / max(batches, 1)},
    "families": {name: {"share_of_gpu_time": f["us"] * 1e3 / total,
                        "us_per_batch": f["us"] / max(batches, 1), "calls": f["calls"]}
                 for name, f in fam.items() if f["calls"]},
    "torch_and_libraries": {
        "share_of_gpu_time": other["us"] * 1e3 / total,
        "us_per_batch": other["us"] / max(batches, 1), "calls": other["calls"],
        "top": dict(sorted(((k, round(v["total_ms"], 2)) for k, v in other["kernels"].items()),
                           key=lambda kv: -kv[1])[:12])},
    "block_ops_at_epoch_shapes": {
        "shape": "{} destinations, {:.0f} edges per block, {} heads, {}-d".format(R, E, H, D),
        "kernels": block},
}
print(json.dumps(out, indent=1))
