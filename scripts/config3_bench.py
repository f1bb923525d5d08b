"""BASELINE config 3: sampler roofline sweep on the synthetic power-law graph (10 M nodes /
200 M edges unless --nodes/--edges say otherwise).  One record per (policy, batch):
kernel times from dispatch-attached HIP-event scopes (gf_profile_*), algorithmic bytes per
SURVEY.md 8(d), fractions of the 8 TB/s HBM peak.  `sweep()` is what bench.py puts into its
line as "config3"; run as a script it prints one JSON line per record.  Run under
`rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes) for the counter traffic of the
search / emit kernels (scripts/rocprof_config3.sh).
Reference harness: benchmarks/benchmark_sampler.py:50-87 (root layout [src || dst || random],
timestamps = edge time x 3); kernels gnnflow/csrc/sampling_kernels.cu:109-273.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def sweep(N, E, batches, policies, dev, reps=10, emit=None):
    """Builds the graph and times sample() per (policy, batch).  Returns
    {"graph": {...}, "rows": [...]}; `emit(record)` is called per record if given."""
    import numpy as np
    import torch

    import gnnflow_amd
    from gnnflow_amd import _capi, synthetic
    lib = _capi.load()
    t0 = time.time()
    g = synthetic.powerlaw_device(N, E, dev, seed=42)
    gen_s = time.time() - t0
    graph = gnnflow_amd.DynamicGraph(1 << 30, 64 << 30, "cuda", 16, 1024, "insert",
                                     device=dev.index or 0)
    t0 = time.time()
    for lo in range(0, E, 10_000_000):
        hi = lo + 10_000_000
        graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
    ingest_s = time.time() - t0
    build = {"nodes": N, "edges": E, "gen_s": round(gen_s, 2), "ingest_s": round(ingest_s, 2),
             "ingest_Medges_per_s": E / ingest_s / 1e6}
    if emit:
        emit(dict(build, sweep="config3-build"))
    deg = torch.bincount(g["device"]["src"], minlength=N)
    del g["device"]["dst"], g["device"]["eid"]

    def prof(slot):
        ms, n = C.c_double(0), C.c_uint64(0)
        lib.gf_profile_get(_capi.PROFILE_SLOTS[slot], C.byref(ms), C.byref(n))
        return ms.value, n.value

    rows = []
    for policy in policies:
        s = gnnflow_amd.TemporalSampler(graph, [10, 10], policy)
        for B in batches:
            rng = np.random.RandomState(B)
            pick = rng.randint(int(E * 0.99), E, B)
            roots = np.concatenate([g["src"][pick], g["dst"][pick],
                                    rng.randint(0, N, B)]).astype(np.int64)
            ts = np.tile(g["ts"][pick], 3).astype(np.float32)
            r, t = torch.from_numpy(roots).to(dev), torch.from_numpy(ts).to(dev)
            m = None
            for _ in range(2):
                m = s.sample(r, t)
            torch.cuda.synchronize(dev)
            lib.gf_profile_reset()
            lib.gf_profile_enable(0b1011)
            t0 = time.perf_counter()
            for _ in range(reps):
                m = None            # the previous output goes back to the allocator first
                m = s.sample(r, t)
            torch.cuda.synchronize(dev)
            wall = (time.perf_counter() - t0) / reps
            lib.gf_profile_enable(0)
            # ... and as a stream of samples: the pipelined sample-only loop (ReplayPipeline with
            # no cache: two sampling lanes, four samples in flight, uniform draws keyed by call
            # number so the lanes reproduce the one sampler's stream) — the launch gaps and the
            # host wait of one sample hide behind the other lane's kernels
            from gnnflow_amd.pipeline import ReplayPipeline
            pipe = ReplayPipeline(s, None, [(r, t, None)], dev, pipelined=True)
            n_pipe = max(reps, min(400, int(2e-2 / max(wall, 1e-6))))
            pipe.run(0, min(n_pipe, 8))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            pipe.run(0, n_pipe)
            torch.cuda.synchronize(dev)
            wall_pipe = (time.perf_counter() - t0) / n_pipe
            blocks = [b for mfg in m for b in mfg]
            edges = sum(b.num_edges() for b in blocks)
            roots_total = sum(b.num_dst_nodes() for b in blocks)
            # per root 12 (id, ts) + 16 (table entry) + 8 * ceil(log2 deg) + 4
            # (count); per edge 20 read + 40 written
            search_bytes = 0.0
            per_layer = []
            for li, mfg in enumerate(reversed(m)):          # m[-1] is the roots' layer
                lb = 0.0
                for b in mfg:
                    ids = b.srcdata["ID"][:b.num_dst_nodes()]
                    d = deg[ids].clamp(min=2).to(torch.float64)
                    lb += float((28 + 4 + 8 * torch.ceil(torch.log2(d))).sum())
                search_bytes += lb
                per_layer.append({"layer": li, "roots": int(sum(b.num_dst_nodes() for b in mfg)),
                                  "edges": int(sum(b.num_edges() for b in mfg)),
                                  "search_alg_MB": lb / 1e6,
                                  "emit_alg_MB": sum(b.num_edges() for b in mfg) * 60.0 / 1e6})
            emit_bytes = edges * 60.0
            ms_s, _ = prof("search")
            ms_e, _ = prof("emit")
            ms_c, _ = prof("scan")

            def us(ms):
                return 1e3 * ms / reps
            kern_us = us(ms_s) + us(ms_e) + us(ms_c)
            rec = {
                "policy": policy, "batch": B, "roots": int(roots_total), "edges": int(edges),
                "wall_us": 1e6 * wall, "edges_per_s": edges / wall,
                "pipelined_wall_us": 1e6 * wall_pipe, "pipelined_edges_per_s": edges / wall_pipe,
                "pipelined_lanes": len(pipe.lanes),
                "search_us": us(ms_s), "emit_us": us(ms_e), "scan_us": us(ms_c),
                "search_alg_MB": search_bytes / 1e6, "emit_alg_MB": emit_bytes / 1e6,
                "layers": per_layer,
                "search_GBps": search_bytes / (us(ms_s) * 1e-6) / 1e9 if ms_s else None,
                "emit_GBps": emit_bytes / (us(ms_e) * 1e-6) / 1e9 if ms_e else None,
                # the whole sample(): against the kernels' summed time, and against the wall clock
                # (kernels + the boundaries and the one host wait between them)
                "kernel_sum_GBps": (search_bytes + emit_bytes) / (kern_us * 1e-6) / 1e9
                if kern_us else None,
                "all_GBps": (search_bytes + emit_bytes) / wall / 1e9}
            for k in ("search", "emit", "kernel_sum", "all"):
                v = rec[k + "_GBps"]
                rec[k + "_frac"] = v / HBM_PEAK_GBS if v else None
            rows.append(rec)
            if emit:
                emit(dict(rec, sweep="config3"))
            m = blocks = None
    return {"workload": "synthetic power-law temporal graph, {} nodes / {} edges (Zipf sources, "
                        "uniform destinations, sorted f32 times), fanout [10,10], roots = "
                        "[src || dst || random] of B edges drawn from the last 1 % of the "
                        "stream; algorithmic bytes per SURVEY 8(d); times from dispatch-attached "
                        "HIP events, mean of {} samples".format(N, E, reps),
            "peak_GBps": HBM_PEAK_GBS, "graph": build, "rows": rows}


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--edges", type=int, default=200_000_000)
    ap.add_argument("--batches", default="600,6000,60000,600000")
    ap.add_argument("--policies", default="uniform,recent")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from gnnflow_amd.utils import bind_to_device_cpus
    bind_to_device_cpus(0)   # host arrays and ingest threads on the GPU's NUMA node
    sweep(args.nodes, args.edges, [int(x) for x in args.batches.split(",")],
          args.policies.split(","), dev, reps=args.reps,
          emit=lambda rec: print(json.dumps(rec), flush=True))


if __name__ == "__main__":
    main()
