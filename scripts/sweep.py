"""Size sweeps behind the roofline discussion in DESIGN.md / profiles/README.md
(BASELINE.json configs[2]: "HBM GB/s roofline sweep over batch size").

  python scripts/sweep.py gather     # fused gather kernel, rows x 172 floats, cache 0.2
  python scripts/sweep.py sampler    # uniform / recent sampling on a power-law graph
  python scripts/sweep.py blockops   # edge_softmax / update_all on sampler-shaped blocks
Prints one JSON object per line.  HIP-event timing via gf_profile_*.
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gnnflow_amd
from gnnflow_amd import _capi, synthetic
from gnnflow_amd.cache import LRUCache

dev = torch.device("cuda", 0)
lib = _capi.load()


def prof(slot):
    ms, n = C.c_double(0), C.c_uint64(0)
    lib.gf_profile_get(_capi.PROFILE_SLOTS[slot], C.byref(ms), C.byref(n))
    return ms.value, n.value


class Blk:
    def __init__(self, ids):
        self.srcdata = {"ID": torch.zeros(1, dtype=torch.int64, device=dev)}
        self.edata = {"ID": ids}


def gather_sweep():
    d = 172
    for table_rows in (672447, 4000000):
        feats = torch.rand((table_rows, d), device=dev)
        for ratio in (0.0, 0.2):
            cache = LRUCache(ratio, 0.0, 10, table_rows, dev, None, feats, 0, d)
            cache.init_cache()
            rows_list = [int(x) for x in os.environ.get(
                "SWEEP_ROWS", "2000,10000,20000,50000,198000,1000000,4000000").split(",")]
            for n in rows_list:
                g = torch.Generator(device=dev).manual_seed(n)
                ids = torch.randint(0, table_rows, (n,), generator=g, device=dev)
                for _ in range(3):
                    cache.fetch_feature([[Blk(ids)]], update_cache=False)
                torch.cuda.synchronize()
                lib.gf_profile_reset()
                lib.gf_profile_enable(1 << _capi.PROFILE_SLOTS["gather"])
                reps = 20
                t0 = time.perf_counter()
                for _ in range(reps):
                    cache.fetch_feature([[Blk(ids)]], update_cache=False)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / reps
                lib.gf_profile_enable(0)
                ms, k = prof("gather")
                alg = n * (8 + 8 * d)
                print(json.dumps({
                    "sweep": "gather", "table_rows": table_rows, "table_MB": table_rows * d * 4 / 1e6,
                    "cache_ratio": ratio, "rows": n, "algorithmic_MB": alg / 1e6,
                    "event_us": 1e3 * ms / k, "GBps_event": alg / (ms / k * 1e-3) / 1e9,
                    "wall_us": 1e6 * wall, "hit_ratio": float(cache.cache_edge_ratio)}))
            del cache
        del feats
        torch.cuda.empty_cache()


def sampler_sweep():
    N = int(os.environ.get("SWEEP_NODES", 2000000))
    E = int(os.environ.get("SWEEP_EDGES", 40000000))
    t0 = time.time()
    g = synthetic.powerlaw(N, E, seed=42)
    gen_s = time.time() - t0
    graph = gnnflow_amd.DynamicGraph(1 << 30, 64 << 30, "cuda", 16, 1024, "insert")
    t0 = time.time()
    for lo in range(0, E, 10000000):
        hi = lo + 10000000
        graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
    build_s = time.time() - t0
    print(json.dumps({"sweep": "sampler-build", "nodes": N, "edges": E, "gen_s": gen_s,
                      "ingest_s": build_s, "ingest_Medges_per_s": E / build_s / 1e6}))
    tail = int(E * 0.99)
    for strategy in ("uniform", "recent"):
        for group in ("16", "4"):
            os.environ["GNNFLOW_SEARCH_GROUP"] = group
            s = gnnflow_amd.TemporalSampler(graph, [10, 10], strategy)
            for B in (600, 6000, 60000, 300000):
                rng = np.random.RandomState(B)
                pick = rng.randint(tail, E, B)
                roots = np.concatenate([g["src"][pick], g["dst"][pick],
                                        rng.randint(0, N, B)]).astype(np.int64)
                ts = np.tile(g["ts"][pick], 3).astype(np.float32)
                r, t = torch.from_numpy(roots).to(dev), torch.from_numpy(ts).to(dev)
                for _ in range(2):
                    m = s.sample(r, t)
                torch.cuda.synchronize()
                lib.gf_profile_reset()
                lib.gf_profile_enable(0b1011)
                reps = 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    m = s.sample(r, t)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / reps
                lib.gf_profile_enable(0)
                edges = sum(b.num_edges() for mfg in m for b in mfg)
                roots_total = sum(b.num_dst_nodes() for mfg in m for b in mfg)
                ms_s, _ = prof("search")
                ms_e, _ = prof("emit")
                ms_c, _ = prof("scan")
                # R*(12+16) + R*8*log2(deg~E/N) + S*20 read + S*40 written + R*4
                alg = roots_total * (28 + 8 * np.log2(max(E / N, 2)) + 4) + edges * 60
                kern_ms = (ms_s + ms_e + ms_c) / reps
                print(json.dumps({
                    "sweep": "sampler", "strategy": strategy, "group": int(group), "batch": B,
                    "roots": int(roots_total), "edges": int(edges),
                    "wall_us": 1e6 * wall, "edges_per_s_wall": edges / wall,
                    "kernel_us": 1e3 * kern_ms, "search_us": 1e3 * ms_s / reps,
                    "emit_us": 1e3 * ms_e / reps, "scan_us": 1e3 * ms_c / reps,
                    "algorithmic_MB": alg / 1e6, "GBps_kernels": alg / (kern_ms * 1e-3) / 1e9}))


def blockops_sweep():
    """edge_softmax + update_all(copy_src, sum), forward and backward, on blocks shaped like
    the sampler's (R roots x fanout 10, every edge a new source node), against the same ops
    written with torch index_add / scatter_reduce.  torch.cuda events, 20 repetitions."""
    from gnnflow_amd import MFGBlock, ops

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps

    H, D = 2, 50          # TGN/TGAT attention: 2 heads, dim_out 100
    # real sampler output (so the ops see the sampler's layout, col[k] = num_dst + k): every
    # node of this graph has 16 edges, hence 10 sampled edges per root
    N = 100000
    g = gnnflow_amd.DynamicGraph(1 << 26, 1 << 30, "cuda", 16, 1024, "insert")
    src = np.repeat(np.arange(N, dtype=np.int64), 16)
    rs = np.random.RandomState(0)
    g.add_edges(src, rs.randint(0, N, len(src)).astype(np.int64),
                np.tile(np.arange(16, dtype=np.float32), N))
    sampler = gnnflow_amd.TemporalSampler(g, [10], "recent")
    for R in (1800, 19800, 198000, 1980000):
        E = R * 10
        blk = sampler.sample(rs.randint(0, N, R).astype(np.int64),
                             np.full(R, 100.0, np.float32))[0][0]
        assert blk.num_edges() == E and blk.segments()[1] is None
        col, row = blk.edges()
        x = torch.randn(E, H, device=dev, requires_grad=True)
        v = torch.randn(R + E, H * D, device=dev, requires_grad=True)
        gy = torch.randn(E, H, device=dev)
        gh = torch.randn(R, H * D, device=dev)

        def hip_softmax():
            ops.edge_softmax(blk, x).backward(gy)

        def torch_softmax():
            m = torch.full((R, H), -float("inf"), device=dev).scatter_reduce(
                0, row[:, None].expand(E, H), x, "amax")
            ex = torch.exp(x - m[row])
            s = torch.zeros(R, H, device=dev).index_add(0, row, ex)
            (ex / s[row]).backward(gy)

        def hip_reduce():
            ops.block_reduce(blk, v).backward(gh)

        def torch_reduce():
            torch.zeros(R, H * D, device=dev).index_add(0, row, v[col]).backward(gh)

        out = {"sweep": "blockops", "roots": R, "edges": E, "heads": H, "dim": H * D}
        out["edge_softmax_fwd_bwd_us"] = timed(hip_softmax)
        out["edge_softmax_torch_us"] = timed(torch_softmax)
        out["update_all_fwd_bwd_us"] = timed(hip_reduce)
        out["update_all_torch_us"] = timed(torch_reduce)
        # algorithmic bytes: softmax fwd reads x, writes y; bwd reads y, gy, writes gx (5 x 4 B
        # per edge-head); reduce fwd reads E rows, writes R rows; bwd reads R rows (re-read per
        # edge from cache), writes (R + E) rows (memset + adds on E of them)
        sm_bytes = 5 * 4 * E * H
        rd_bytes = 4 * H * D * (E + R + R + (R + E) + E)
        out["edge_softmax_GBps"] = sm_bytes / out["edge_softmax_fwd_bwd_us"] / 1e3
        out["update_all_GBps"] = rd_bytes / out["update_all_fwd_bwd_us"] / 1e3
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gather"
    if what == "blockops":
        blockops_sweep()
        sys.exit(0)
    gather_sweep() if what == "gather" else sampler_sweep()
