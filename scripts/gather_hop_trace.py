"""Where the fetch chain's two launches spend their time, on ONE clock: wave 0 of every workgroup
of gather_rows_kernel stamps the 100 MHz wall clock at the stages of its first tile, every
workgroup of lru_list_fused_kernel at its role's stages (gf_debug_lru_trace, both caches traced).
Replays the REDDIT-shaped stream (LRU 0.2, batch 600, the headline's fetch) through the plain loop
and prints, for N late steps, the stages' times since the gather launch's FIRST stamp.
A traced gather waits for its ids and map values before it issues the rows (the stamp needs the
wait), so its row phase starts one hop later than in an untraced launch.
    python scripts/gather_hop_trace.py [steps=40] > profiles/r06_gather_hop_trace.txt
Reference: gnnflow/cache/cache.py:255-413 (the fetch these two launches replace)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnflow_amd  # noqa: E402
from gnnflow_amd import _capi, synthetic  # noqa: E402
from gnnflow_amd.cache import LRUCache  # noqa: E402
from gnnflow_amd.utils import bind_to_device_cpus  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bind_to_device_cpus(0)
lib = _capi.load()
dev = torch.device("cuda", 0)
g = synthetic.reddit_like(seed=42)
MiB = 1 << 20
graph = gnnflow_amd.DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert")
for lo in range(0, g["num_edges"], 100000):
    hi = lo + 100000
    graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
sampler = gnnflow_amd.TemporalSampler(graph, [10, 10], "recent", seed=1234)
gen = torch.Generator(device=dev).manual_seed(1)
ef = torch.rand((g["num_edges"], 172), generator=gen, device=dev)
nf = torch.rand((g["num_nodes"], 172), generator=gen, device=dev)
cache = LRUCache(0.2, 0.2, g["num_nodes"], g["num_edges"], dev, nf, ef, 172, 172)
cache.init_cache()
batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(e).to(dev))
           for r, t, e in synthetic.replay_batches(g, 600, seed=42)]
first = len(batches) - steps - 1
for i in range(first):                       # bring the cache to the late part of the replay
    r, t, e = batches[i]
    cache.fetch_feature(sampler.sample(r, t), e)
torch.cuda.synchronize()
kinds = {"edge": cache._edge.h, "node": cache._node.h}
for h in kinds.values():
    _capi.check(lib.gf_debug_lru_trace_enable(h, 1))
LRU_WORDS = 4 + 8 * (2 * 2048 + 1024)          # kGatherTraceBase
GATHER = ["entry (kernel arguments in)", "ids and map values in", "rows in, stores issued",
          "stores acknowledged", "marks / claims / counters issued"]
acc = {k: [[] for _ in GATHER] for k in kinds}
by_wg = {k: {} for k in kinds}      # workgroup index -> its "rows in" times
lru_first, lru_last, gather_last = [], [], []
buf = (C.c_uint64 * (LRU_WORDS + 8 * 1024))()
n = C.c_size_t(0)
for i in range(first, first + steps):
    r, t, e = batches[i]
    mfgs = sampler.sample(r, t)
    cache.fetch_feature(mfgs, e)
    torch.cuda.synchronize()
    got = {}
    for k, h in kinds.items():
        _capi.check(lib.gf_debug_lru_trace(h, buf, len(buf), C.byref(n)))
        got[k] = np.frombuffer(buf, dtype=np.uint64, count=n.value).astype(np.int64).copy()
    gs = {k: w[LRU_WORDS:].reshape(-1, 8) for k, w in got.items()}
    t0 = min(s[s[:, 0] > 0, 0].min() for s in gs.values())
    glast = 0
    for k, s in gs.items():
        for wg in np.nonzero(s[:, 0] > 0)[0]:
            for j in range(len(GATHER)):
                if s[wg, j] > 0:
                    acc[k][j].append((s[wg, j] - t0) * 0.01)
                    glast = max(glast, s[wg, j] - t0)
                    if j == 2:
                        by_wg[k].setdefault(int(wg), []).append((s[wg, j] - s[wg, 1]) * 0.01)
    gather_last.append(glast * 0.01)
    lf, ll = [], []
    for k, w in got.items():
        cb, rb, wb = int(w[0]), int(w[1]), int(w[2])
        st = w[4:4 + 8 * (cb + rb + wb)].reshape(-1, 8)
        used = st[st[:, 0] > 0]
        if len(used):
            lf.append(used[:, 0].min() - t0)
            ll.append(used.max() - t0)
    if lf:
        lru_first.append(min(lf) * 0.01)
        lru_last.append(max(ll) * 0.01)
    for h in kinds.values():
        lib.gf_debug_lru_trace_enable(h, 1)      # zero the stamps for the next fetch
print("gather_rows_kernel + lru_list_fused_kernel of the headline replay's fetch (node block ~25 k rows,")
print("edge block ~11 k rows of 172 floats, 600 target rows), {} consecutive late steps; microseconds since the gather".format(steps))
print("launch's first stamp (wave 0 of each workgroup, its first tile):")
print("{:5s} {:42s} {:>7s} {:>7s} {:>7s} {:>7s} {:>6s}".format("ctx", "stage", "p10", "median", "p90", "max", "n"))
for k in kinds:
    for j, name in enumerate(GATHER):
        v = np.array(acc[k][j])
        if len(v):
            print("{:5s} {:42s} {:7.2f} {:7.2f} {:7.2f} {:7.2f} {:6d}".format(
                k, name, np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max(), len(v)))
for name, v in (("last stamp of the gather launch", gather_last),
                ("first stamp of the update launch behind it", lru_first),
                ("last stamp of the update launch", lru_last)):
    v = np.array(v)
    print("{:48s} median {:6.2f} us (p10 {:.2f}, p90 {:.2f})".format(name, np.median(v), np.percentile(v, 10),
                                                                   np.percentile(v, 90)))
print("rows phase (ids in -> rows in) by workgroup index, mean over the steps, in eighths of the grid:")
for k in kinds:
    idx = sorted(by_wg[k])
    if not idx:
        continue
    m = np.array([np.mean(by_wg[k][i]) for i in idx])
    parts = np.array_split(np.arange(len(idx)), 8)
    print("{:5s} ".format(k) + "  ".join("wg {:3d}-{:3d}: {:5.2f}".format(idx[p[0]], idx[p[-1]], m[p].mean())
                                       for p in parts if len(p)))
    worst = np.argsort(m)[-6:][::-1]
    print("      slowest: " + ", ".join("wg {} {:.2f}".format(idx[i], m[i]) for i in worst))
# the last step's edge context, workgroup by workgroup: what its 64 rows are
b = mfgs[0][0]
ids = b.edata["ID"].cpu().numpy()
s = gs["edge"]
rows_t = {int(wg): (s[wg, 2] - s[wg, 1]) * 0.01 for wg in np.nonzero(s[:, 1] > 0)[0]}
print("last step, edge block of {} rows ({} distinct ids, newest id {}): rows phase of a workgroup against its"
      .format(len(ids), len(np.unique(ids)), int(ids.max())))
print("64 rows — distinct ids among them, ids seen EARLIER in the block, median age (newest id - id):")
seen_before = np.zeros(len(ids), dtype=bool)
first_pos = {}
for i, v in enumerate(ids):
    if int(v) in first_pos:
        seen_before[i] = True
    else:
        first_pos[int(v)] = i
order = sorted(rows_t, key=rows_t.get)
for wg in order[:5] + order[-8:]:
    lo, hi = wg * 64, min(len(ids), wg * 64 + 64)
    if lo >= len(ids):
        continue
    part = ids[lo:hi]
    print("  wg {:3d} rows {:5d}-{:5d}: {:5.2f} us, {:2d} distinct, {:2d} seen earlier, median age {:8d}".format(
        wg, lo, hi, rows_t[wg], len(np.unique(part)), int(seen_before[lo:hi].sum()),
        int(np.median(ids.max() - part))))
# where the last step's workgroups ran (HW_ID / XCC_ID of wave 0) and how long their rows took
print("last step, both contexts: rows phase by the CU a workgroup ran on (xcc, se, sh, cu):")
place = {}
for k, sg in gs.items():
    for wg in np.nonzero(sg[:, 1] > 0)[0]:
        hw = int(sg[wg, 5]) & 0xFFFFFFFF
        xcc = (int(sg[wg, 5]) >> 32) & 0xF
        cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
        place.setdefault((xcc, se, sh, cu), []).append((k, int(wg), (sg[wg, 2] - sg[wg, 1]) * 0.01,
                                                       (sg[wg, 1] - t0) * 0.01))
loads = sorted(place.items(), key=lambda kv: -max(x[2] for x in kv[1]))
print("  CUs used: {}; workgroups per CU: {}".format(
    len(place), dict(zip(*np.unique([len(v) for v in place.values()], return_counts=True)))))
for key, v in loads[:10]:
    print("  xcc {} se {} sh {} cu {:2d}: ".format(*key) +
          ", ".join("{} wg {} rows {:.2f} us (ids in at {:.2f})".format(*x) for x in v))
per_n = {}
for v in place.values():
    per_n.setdefault(len(v), []).extend(x[2] for x in v)
for nn in sorted(per_n):
    print("  CUs with {} traced workgroups: rows phase mean {:.2f} us, max {:.2f}".format(
        nn, np.mean(per_n[nn]), np.max(per_n[nn])))
