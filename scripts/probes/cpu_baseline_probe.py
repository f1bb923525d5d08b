"""How the CPU oracle baseline of bench.py behaves on this box: one chronological pass of the
REDDIT-shaped replay per thread count (diagnostic; prints edges/s and ms per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gnnflow_amd import synthetic
from oracle import oracle as O
import bench

g = synthetic.reddit_like(seed=42)
og = O.OracleGraph(minimum_block_size=62, insertion_policy="insert")
for lo in range(0, g["num_edges"], 100000):
    hi = lo + 100000
    og.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
batches = list(synthetic.replay_batches(g, 600, seed=42))
rng = np.random.RandomState(0)
ef = rng.rand(g["num_edges"], 172).astype(np.float32)
nf = rng.rand(g["num_nodes"], 172).astype(np.float32)
print("usable_cores", bench.usable_cores(), "affinity", len(os.sched_getaffinity(0)),
      "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
for threads in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,4,8,16,32").split(",")]:
    s = O.OracleSampler(og, [10, 10], "recent", seed=1234, threads=threads)
    for rep in range(2):
        ts, tg, edges = 0.0, 0.0, 0
        for r, t, e in batches:
            t0 = time.perf_counter()
            mfgs = s.sample(r, t)
            t1 = time.perf_counter()
            for blk in mfgs[0]:
                O.gather_rows(nf, blk.srcdata["ID"], threads)
            for mfg in mfgs:
                for blk in mfg:
                    if blk.num_edges():
                        O.gather_rows(ef, blk.edata["ID"], threads)
            O.gather_rows(ef, e, threads)
            t2 = time.perf_counter()
            ts += t1 - t0; tg += t2 - t1
            edges += sum(b.num_edges() for mfg in mfgs for b in mfg)
        print("threads %2d pass %d: %.2f M edges/s  sample %.3f ms/step  gather %.3f ms/step" % (
            threads, rep, edges / (ts + tg) / 1e6, 1e3 * ts / len(batches), 1e3 * tg / len(batches)), flush=True)
