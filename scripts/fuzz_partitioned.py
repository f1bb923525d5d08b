"""Fuzzer for the native partitioned chains (single and shared by 2-4 samples, slotted +
overflow redo) over the in-process loopback transport: random world sizes, graphs, fanouts, batch sizes (ragged, empty),
slot capacities, snapshots / windows / prop_time; every most-recent MFG of every rank is compared
bit for bit with the CPU oracle over the whole graph.  Beyond the test suite (run on the GPU box):

    python scripts/fuzz_partitioned.py [--seeds 200] [--first 0]
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gnnflow_amd import DynamicGraph, TemporalSampler
from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm, PartitionedGraph
from oracle import oracle as O
from tests import synth


def same(gb, wb):
    ok = np.array_equal(gb["ID"], wb.srcdata["ID"])
    ok &= np.array_equal(gb["ts"], wb.srcdata["ts"])
    ok &= np.array_equal(gb["eid"], wb.edata["ID"])
    ok &= np.array_equal(gb["dt"].view(np.uint8), np.asarray(wb.edata["dt"]).view(np.uint8))
    ok &= np.array_equal(gb["row"], wb.edges()[1]) and np.array_equal(gb["col"], wb.edges()[0])
    return bool(ok)


def one(seed, dev):
    rng = np.random.RandomState(seed)
    P = int(rng.choice([2, 3, 4, 5, 8]))
    N = int(rng.choice([50, 400, 3000]))
    E = int(rng.choice([2000, 12000, 60000]))
    L = int(rng.choice([1, 2, 3], p=[0.25, 0.6, 0.15]))
    fan = [int(rng.randint(1, 9)) for _ in range(L)]
    snaps = int(rng.choice([1, 1, 1, 2]))
    window = float(rng.choice([0.0, 40.0, 200.0])) if snaps == 1 else float(rng.choice([30.0, 120.0]))
    prop = bool(rng.randint(0, 2))
    slack = float(rng.choice([2.0, 2.0, 1.0, 0.3, 0.05]))
    chain = int(rng.choice([1, 2, 3, 4, 4]))
    narrow = bool(rng.randint(0, 2))
    adapt = bool(rng.randint(0, 2))
    # compact reply slots (shared chains): none / far too small (overflow -> redo) / roomy / all
    edge_fill = float(rng.choice([0.0, 0.02, 0.1, 0.4, 1.0]))
    reuse = bool(rng.randint(0, 2))        # do not request the previous layer's roots again
    if rng.randint(0, 2) and L > 1:        # ... which needs equal fanouts: make them so half the time
        fan = [fan[0]] * L
    minblk = int(rng.choice([4, 8, 62]))
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=seed, tie_levels=int(rng.choice([50, 500, 5000])))
    full = O.OracleGraph(minimum_block_size=minblk)
    shards = [DynamicGraph(1 << 20, 256 << 20, "cuda", minblk, 64, "insert") for _ in range(P)]
    parts = [PartitionedGraph(s, r, P) for r, s in enumerate(shards)]
    chunk = int(rng.choice([500, 2500, 100000]))
    rev = bool(rng.randint(0, 2))
    for lo in range(0, E, chunk):
        sl = slice(lo, lo + chunk)
        full.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=rev)
        for pg in parts:
            pg.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=rev)
    n_samples = int(rng.randint(3, 14))
    pool = [0, 1, 2, 17, 97, 300, 600, 1500]
    slot_roots = int(rng.choice([600, 1500]))
    batches = [[synth.random_roots(N, int(pool[rng.randint(0, len(pool))]), 1000.0,
                                   seed=seed * 1000 + 31 * r + it, extra_ids=[N + 3])
                for it in range(n_samples)] for r in range(P)]
    kw = dict(fanouts=fan, sample_strategy="recent", num_snapshots=snaps,
              snapshot_time_window=window, prop_time=prop)
    comms = NativeComm.loopback(P, dev)
    res, err = [None] * P, [None] * P
    inflight = int(rng.choice([1, 2, 3, 5, 8]))

    def body(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                part = DevicePartitionedSampler(TemporalSampler(shards[r], **kw), comm=comms[r],
                                                slack=slack, slot_roots=slot_roots,
                                                chain_samples=chain, narrow_ids=narrow,
                                                adapt_slack=adapt, edge_fill=edge_fill,
                                                reuse_roots=reuse)
                side = torch.cuda.Stream()
                got = []
                for lo in range(0, n_samples, inflight):
                    pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev),
                                              stream=side) for n, t in batches[r][lo:lo + inflight]]
                    for p in pend:
                        m = p.wait()
                        got.append([[{k: v.cpu().numpy() for k, v in (
                            ("ID", b.srcdata["ID"]), ("ts", b.srcdata["ts"]), ("eid", b.edata["ID"]),
                            ("dt", b.edata["dt"]), ("row", b.edges()[1]), ("col", b.edges()[0]))}
                            for b in mfg] for mfg in m])
                res[r] = (got, (part.overflows, part._slack_epoch), part.pairs)
        except BaseException as e:   # noqa: BLE001
            import traceback
            err[r] = "{}: {}\n{}".format(type(e).__name__, e, traceback.format_exc()[-800:])
    th = [threading.Thread(target=body, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    desc = dict(seed=seed, P=P, N=N, E=E, fan=fan, snaps=snaps, window=window, prop=prop, slack=slack,
                chain=chain, narrow=narrow, adapt=adapt, inflight=inflight, slot_roots=slot_roots,
                edge_fill=edge_fill, reuse=reuse)
    if any(t.is_alive() for t in th):
        return "HANG", desc
    if any(err):
        return "ERROR " + " | ".join(e for e in err if e), desc
    ref = O.OracleSampler(full, fan, "recent", num_snapshots=snaps, snapshot_time_window=window,
                          prop_time=prop, threads=4)
    for r in range(P):
        got, over, pairs = res[r]
        if over != res[0][1]:
            return "ranks disagree on the overflowed samples", desc
        for it, ((n, t), mfgs) in enumerate(zip(batches[r], got)):
            for gl, wl in zip(mfgs, ref.sample(n, t)):
                for gb, wb in zip(gl, wl):
                    if not same(gb, wb):
                        return "MISMATCH rank {} sample {} ({} roots)".format(r, it, len(n)), desc
    for c in comms:
        c.close()
    return None, dict(desc, overflows=res[0][1], pairs=res[0][2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    bad, over, pairs, t0 = 0, 0, 0, time.time()
    for seed in range(args.first, args.first + args.seeds):
        verdict, desc = one(seed, dev)
        if verdict:
            bad += 1
            print("FAIL", verdict, desc, flush=True)
            if verdict == "HANG":
                break
        else:
            over += desc["overflows"][0]
            pairs += desc["pairs"]
        if (seed + 1) % 20 == 0:
            print("... {} seeds, {} failures, {} overflowed samples redone, {} shared chains, {:.0f} s"
                  .format(seed + 1 - args.first, bad, over, pairs, time.time() - t0), flush=True)
    print("fuzz_partitioned: {} seeds, {} failures, {} overflowed samples (per rank 0), {} shared "
          "chains".format(args.seeds, bad, over, pairs))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
