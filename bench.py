#!/usr/bin/env python3
"""
bench.py — sampled edges/s of the GNNFlow hot path on MI355X.

One step = one pass of the hot path over one batch of the REDDIT-shaped synthetic
stream: TemporalSampler.sample() (2 layers, fanout [10,10], most-recent, 1800 roots =
[src || dst || random] of a 600-edge batch) followed by LRUCache.fetch_feature()
(edge + node cache ratio 0.2, 172-d features) — BASELINE.json configs[1].  Inputs
(roots, timestamps, feature tables, graph) are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N
ranks: the parent — before importing torch or touching HIP — starts N fresh children of this
file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's one JSON line and
exits non-zero if any child does (launch_ranks below).

The workload is the CHRONOLOGICAL REPLAY of the whole stream (1121 batches): early
batches sample almost nothing and late ones ~25k edges, and the LRU cache only behaves
like the real thing when consecutive batches follow each other.  So the timed region is
`repeats` back-to-back windows of K consecutive batches, starting at batch 0 on a freshly
initialised cache and wrapping around at the end of the stream.  Unless --repeats says
otherwise, `repeats` covers >= --min-replays (4) whole replays AND >= --min-seconds (2 s) of
wall time — the step time comes from an untimed calibration replay, the maximum over the ranks —
so whatever K is, edges_per_step is the full-replay mean and a clock outside the process sees
the region.  ms_per_step = elapsed / (K * repeats); `timed_steps` = K * repeats and
`timed_seconds` sit next to `steps` in the line.  The loop itself is
gnnflow_amd.pipeline.ReplayPipeline.run — the function tests/test_gpu_pipeline_parity.py
checks against the oracle.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     feature-gather kernel, algorithmic bytes / HIP-event time vs 8 TB/s
  "cpu_baseline": the CPU oracle (C port of the reference algorithm) timed on this
                  host on a bounded sample of the same batches (rank 0, N=1 only)
  "config3":      (N=1) BASELINE configs[2] — uniform sampling on the synthetic power-law graph
                  (10 M nodes / 200 M edges) at batch 600 / 6 000 / 60 000 / 600 000: edges/s,
                  search / emit GB/s from dispatch-attached events, algorithmic fractions.
Multi-GPU (N > 1): north_star's split is the headline — the graph is HASH-PARTITIONED by
source vertex over the N ranks (owner(v) = splitmix64(v) mod N), every rank replays its own
interleaved share of the batches (fixed work per GPU: "scaling": "weak"), and per layer
the roots travel to their owners and the sampled neighbours back as RCCL all-to-alls
(gnnflow_amd.dist.DevicePartitionedSampler, slotted exchange: no host synchronisation
inside a sample).  Feature tables (REDDIT-shaped: 463 MB) are replicated, the LRU cache is
per GPU.  The per-GPU-replica figure (no data-path collective; what the reference does
inside one machine) rides along as the extra key "replica"; `--partition replica` makes
it the main loop.  RCCL between ranks runs for the first time in the driver's scaling run, so
an N > 1 run walks a LADDER (supervise(), RUNGS), each rung in fresh worker processes: the hash
loop as above; should it FAIL or HANG (a collective some rank never joins: the worker's watchdog
fires after GNNFLOW_HASH_MAIN_TIMEOUT seconds), the hash loop on ONE lane (one communicator) with
shared chains; then on one lane with single chains and slot capacity 2.0; then the replica loop.
The one line says which rung produced it and what happened before ("ladder").
At N = 1 the main loop is the plain single-GPU path; the hash path rides along twice: "hash_partition"
(every root is the rank's own, no exchange) and "hash_partition_over_rccl_one_rank" (the chain an
N > 1 run issues — slotted exchange, lanes, up to four samples per chain — with every message
going through RCCL to the rank itself).
"""
import argparse
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
PMC_TRAFFIC_FILE = "r06_pmc_gather_traffic.json"


def workload_key(args):
    """What a committed PMC traffic measurement must have been taken with to be quoted (the
    per-launch traffic of the gather does not depend on how often the replay is repeated)."""
    return {"steps": args.steps, "warmup": args.warmup,
            "batch_size": args.batch_size, "fanouts": args.fanouts, "strategy": args.strategy,
            "cache_ratio": args.cache_ratio, "undirected": bool(args.undirected),
            "feature_placement": args.feature_placement, "partition": args.partition,
            "pipelined": not args.no_pipeline, "pipeline_depth": args.pipeline_depth,
            "sample_only": bool(args.sample_only)}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1121)   # one full chronological replay
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=0,
                    help="windows of --steps consecutive batches in the timed region "
                         "(0 = enough for --min-replays whole replays and --min-seconds)")
    ap.add_argument("--min-replays", type=float, default=4.0)
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="lower bound of the timed region's wall time (an outside clock and "
                         "rocm-smi's busy sampling must be able to see it)")
    ap.add_argument("--batch-size", type=int, default=600)
    ap.add_argument("--fanouts", type=str, default="10,10")
    ap.add_argument("--strategy", type=str, default="recent")
    ap.add_argument("--cache-ratio", type=float, default=0.2)
    ap.add_argument("--undirected", action="store_true",
                    help="add reverse edges (the reference's REDDIT config is directed)")
    ap.add_argument("--feature-placement", default="device", choices=["device", "pinned"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sample-only", action="store_true",
                    help="time sample() alone (no feature gather)")
    ap.add_argument("--partition", default=None, choices=["replica", "hash"],
                    help="the main loop's graph: hash-partitioned shards with an all-to-all "
                         "exchange per layer (gnnflow_amd/dist.py; the default with --gpus > 1) "
                         "or a full replica per GPU (the default with one GPU; what the "
                         "reference does inside one machine)")
    ap.add_argument("--no-hash-leg", "--no-second-leg", dest="no_second_leg", action="store_true",
                    help="skip the second timed run over the other kind of graph")
    ap.add_argument("--part-slack", type=float, default=None,
                    help="slot capacity factor of the partitioned exchange (0 = variable-size "
                         "all-to-all-v with one host sync per layer; default GNNFLOW_PART_SLACK, "
                         "else 1.2 + 0.1 x ranks: 1.4 / 1.6 / 2.0 at 2 / 4 / 8)")
    ap.add_argument("--part-lanes", type=int, default=None,
                    help="sampling lanes of the partitioned sampler: consecutive batches go "
                         "round-robin to lanes with their own stream, workspace and communicator, "
                         "so their exchange chains overlap (default GNNFLOW_PART_LANES, else 2; 3 from 4 ranks on)")
    ap.add_argument("--part-chain", type=int, default=None,
                    help="consecutive batches that share one chain of the partitioned sampler "
                         "(launches and exchanges): 1..4, default GNNFLOW_PART_CHAIN, else 4")
    ap.add_argument("--shard-features", action="store_true",
                    help="hash-partitioned run: shard the feature tables by owner too "
                         "(Cache(distributed=True): missed rows are pulled from their owners); "
                         "default: replicated tables")
    ap.add_argument("--always-exchange", action="store_true",
                    help="with one rank: still run the exchange (every message empty) — "
                         "prices the RCCL calls on a one-GPU box")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap batch i+1's sample() with batch i's fetch_feature()")
    ap.add_argument("--pipeline-depth", type=int, default=None,
                    help="batches whose sample() is in flight ahead of the fetch (default 2; "
                         "three per lane of the partitioned sampler)")
    ap.add_argument("--event-stride", type=int, default=61,
                    help="time every n-th gather launch with HIP events (1 = all)")
    ap.add_argument("--breakdown", action="store_true",
                    help="extra untimed pass with per-kernel-family HIP-event times")
    ap.add_argument("--no-placement-legs", action="store_true",
                    help="skip the sample_only / cache_free_device / pinned_placement legs")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the config-3 sweep (10 M nodes / 200 M edges, uniform) at N = 1")
    ap.add_argument("--config3-batches", default="600,6000,60000,600000")
    ap.add_argument("--config3-nodes", type=int, default=10_000_000)
    ap.add_argument("--config3-edges", type=int, default=200_000_000)
    return ap.parse_args(argv)


# ---- N > 1 without a launcher: this file starts its own ranks -------------------------------
def launch_ranks(args, argv):
    """The parent of `python bench.py --gpus N` (N > 1, no WORLD_SIZE): N fresh children of this
    file, one per GPU, rendezvous on 127.0.0.1.  Nothing here imports torch or touches HIP (a
    process that has initialised the GPU must not start workers by fork/exec).  Rank 0's stdout
    is relayed — it carries the one JSON line —, the other ranks' goes to stderr.  Exit status:
    0 only if every child exited 0; a child that dies takes the others with it after a grace
    period (their watchdogs fire first, so rank 0 still gets its error record out)."""
    import signal
    import socket
    import subprocess
    import threading
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n),
               LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    kids = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        kids.append(subprocess.Popen(cmd, env=e, stdin=subprocess.DEVNULL,
                                     stdout=subprocess.PIPE if r == 0 else sys.stderr))

    def stop(sig):
        for k in kids:
            if k.poll() is None:
                try:
                    k.send_signal(sig)
                except OSError:
                    pass

    def on_signal(signum, _frame):
        stop(signal.SIGTERM)
        time.sleep(2.0)
        stop(signal.SIGKILL)
        os._exit(128 + signum)
    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)

    lines = []

    def relay():
        for raw in kids[0].stdout:
            line = raw.decode(errors="replace")
            if line.startswith("{"):
                lines.append(line)
            else:
                sys.stderr.write(line)
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    limit = float(os.environ.get("GNNFLOW_BENCH_TIMEOUT", "900")) + 120.0
    t0 = time.time()
    bad_at = None
    while any(k.poll() is None for k in kids):
        time.sleep(0.1)
        if bad_at is None and any(k.poll() not in (None, 0) for k in kids):
            bad_at = time.time()
        if bad_at is not None and time.time() - bad_at > 30.0:
            stop(signal.SIGTERM)
            time.sleep(2.0)
            stop(signal.SIGKILL)
        if time.time() - t0 > limit:
            stop(signal.SIGKILL)
    t.join(timeout=10.0)
    rcs = [k.returncode for k in kids]
    rc = next((c for c in rcs if c), 0)
    if lines:
        sys.stdout.write(lines[-1])       # exactly one line
    else:
        sys.stdout.write(json.dumps({
            "metric": "sampled_edges_per_s", "value": 0.0, "unit": "edges/s", "n_gpus": n,
            "error": "no rank printed a record; exit codes {}".format(rcs)}) + "\n")
        rc = rc or 1
    sys.stdout.flush()
    return rc if rc >= 0 else 128 - rc


def guarded_aux(ctx, pending, key, limit, fn):
    """An AUXILIARY part of the line (another leg, a collective micro-measurement) that may hang
    on a first multi-GPU run: after `limit` seconds rank 0 prints the line it has, with a note
    under `key`, and every rank exits 0 — the main figure is measured and valid, and a non-zero
    status would send the rank's supervisor on to the next rung of the ladder and lose it."""
    import threading
    done = threading.Event()

    def watchdog():
        if not done.wait(limit):
            if ctx.rank == 0:
                pending[key] = {"error": "timed out after {:.0f} s (the main figure above is "
                                         "complete)".format(limit)}
                main.emit(pending)
            os._exit(0)
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        return fn()
    finally:
        done.set()


def all_to_all_record(ctx, sampler, rec):
    """DESIGN 6.2's two free parameters, measured: device time x and issue cost c of one
    equal-split all-to-all at the sizes a chain's layer-1 exchanges have (lane 0's
    communicator), and the projection evaluated on them.  Collective."""
    import torch
    import torch.distributed as dist
    comms = sampler.comms() if hasattr(sampler, "comms") else []
    # x and c, on lane 0's communicator, at the sizes a chain's layer-1 exchanges have
    if comms and hasattr(sampler, "_plan") and sampler._slack > 0:
        m = getattr(sampler, "chain_samples", 1)
        last = sampler.wire_bytes_per_sample()["per_layer"][-1]      # what really travels
        sizes = {"request_layer1": m * last["request_bytes_to_each_peer"],
                 "reply_layer1": m * last["reply_bytes_to_each_peer"]}
        torch.cuda.synchronize()
        dist.barrier()
        x = {}
        for name, nbytes in sizes.items():
            dev_us, host_us = comms[0].time_all_to_all(nbytes, 30, torch.cuda.current_stream(ctx.dev))
            x[name] = {"bytes_per_peer": nbytes, "device_us": dev_us, "issue_us": host_us}
        rec["all_to_all"] = x
        k = 7.0 * 13.5          # a chain of four: 7 kernels, ~95 us (measured at P = 1)
        layers = len(sampler._fanouts)
        # (every layer priced at the last layer's sizes: an upper bound — layer 0's slots are
        # a tenth of layer 1's at fanout 10)
        xs = layers * (x["request_layer1"]["device_us"] + x["reply_layer1"]["device_us"])
        rec["projection"] = {
            "formula": "a chain of m samples on one of L lanes = K us of kernels + the device time "
                       "of its 2 x layers all-to-alls; the lanes sustain one step per "
                       "(K + sum x) / (m L) us; the issuing thread needs (7 x 3 + 2 x layers x c) / m "
                       "+ 9 us per step",
            "K_us": k, "sum_x_us": xs, "m": m, "L": rec["lanes"],
            "chains_sustain_us_per_step": (k + xs) / (m * rec["lanes"]),
            "issuing_thread_us_per_step":
                (21.0 + 2 * layers * x["reply_layer1"]["issue_us"]) / m + 9.0}
    return rec


# ---- one rank -------------------------------------------------------------------------------
class Ctx:
    pass


# the N > 1 ladder (main()): arrangement, what it is
# (time limits per rung: rung_seconds(); the whole ladder fits the driver's 600 s)
# (rung 0 creates one RCCL communicator per lane inside its watched loop, on a box that may
# still be paging torch in: it gets the most; worst case 260 + 120 + 120 s + the replica rung)
RUNG_MAIN_SECONDS = (130, 70, 70)      # the hash loop's own watchdog
RUNG_SETUP_SECONDS = (130, 50, 50)     # on top of it before the supervisor kills the worker
RUNGS = [("hash", "graph hash-partitioned over the ranks; sampling lanes x shared chains of up to "
                  "four samples, one communicator per lane"),
         ("hash-one-lane", "graph hash-partitioned over the ranks; ONE lane — one communicator, "
                           "never two collectives in flight — with shared chains of up to four "
                           "samples"),
         ("hash-simple", "graph hash-partitioned over the ranks; ONE lane, single chains, slot "
                         "capacity 2.0 x the even share: one communicator, never two collectives "
                         "in flight"),
         ("replica", "a full replica of the graph per GPU, no data-path collective")]


def multi_gpu_record(ctx, sampler, cache):
    """What the N > 1 line says about the exchange beyond its rate (the reference all-gathers
    its per-rank sampling timers, gnnflow/distributed/dist_sampler.py:108-127): RCCL's own rank
    count per lane communicator, the ranks' devices, overflowed samples, bytes per step and rank
    on the links, and — DESIGN 6.2's two free parameters — the device time x of one all-to-all
    of the layer-1 reply / request size and the issuing thread's cost c per exchange."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    rec = {"lanes": getattr(sampler, "lanes", 1), "chain_samples": getattr(sampler, "chain_samples", 1),
           "overflowed_samples": int(getattr(sampler, "overflows", 0))}
    comms = sampler.comms() if hasattr(sampler, "comms") else []
    infos = [c.info() for c in comms]
    rec["transport"] = infos[0]["transport"] if infos else \
        "torch.distributed ({})".format(ctx.backend)
    rec["rccl_nranks"] = [i["nranks"] for i in infos] if infos and infos[0]["transport"] == "rccl" \
        else None
    rec["communicator_nranks"] = [i["nranks"] for i in infos]
    buf = C.create_string_buffer(64)
    bus = buf.value.decode() if ctx.lib.gf_device_pci_bus_id(ctx.local_rank, buf, 64) == 0 else "?"
    mine = {"rank": ctx.rank, "device": ctx.local_rank, "pci_bus_id": bus,
            "name": torch.cuda.get_device_name(ctx.dev),
            "overflowed_samples": rec["overflowed_samples"]}
    ranks = [None] * ctx.world
    dist.all_gather_object(ranks, mine)
    rec["ranks"] = ranks
    rec["distinct_devices"] = len({r["pci_bus_id"] for r in ranks})
    if cache is not None and getattr(cache, "distributed", False):
        # --shard-features: what the owner-sharded feature pull put on the links (rows of misses
        # and target edges pulled from their owners; gnnflow/distributed/kvstore.py:285-339)
        sh = cache._shards
        mine_pull = {"rows_pulled": int(sh.rows_pulled), "bytes_sent": int(sh.bytes_sent),
                     "count_read_backs": int(sh.host_syncs)}
        pulls = [None] * ctx.world
        dist.all_gather_object(pulls, mine_pull)
        steps_all = max(ctx.steps_issued, 1)
        rec["feature_pull"] = {
            "rows_pulled_per_step_and_rank": sum(p["rows_pulled"] for p in pulls) / ctx.world / steps_all,
            "bytes_sent_per_step_and_rank": sum(p["bytes_sent"] for p in pulls) / ctx.world / steps_all,
            "count_read_backs_per_step": pulls[0]["count_read_backs"] / steps_all,
            "ranks": pulls}
    if hasattr(sampler, "wire_bytes_per_sample"):
        wb = sampler.wire_bytes_per_sample()
        rec["wire"] = wb
        rec["request_bytes_per_step_and_rank"] = (ctx.world - 1) * wb["request_bytes_to_each_peer"]
        rec["reply_bytes_per_step_and_rank"] = (ctx.world - 1) * wb["reply_bytes_to_each_peer"]
    return rec


def build_leg(ctx, kind, always_exchange=None):
    """Graph + sampler of one kind over this rank's GPU: "replica" = the whole graph, "hash" =
    this rank's shard + the partitioned sampler."""
    import gnnflow_amd
    args, g = ctx.args, ctx.g
    if always_exchange is None:
        always_exchange = args.always_exchange
    MiB = 1 << 20
    # gnnflow/config.py:121-131 _reddit_default_config
    graph = gnnflow_amd.DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert",
                                     device=ctx.local_rank)
    ingest = graph
    if kind == "hash":
        from gnnflow_amd.dist import DevicePartitionedSampler, PartitionedGraph
        ingest = PartitionedGraph(graph, ctx.rank, ctx.world)
    t0 = time.time()
    for lo in range(0, g["num_edges"], 100000):   # benchmark_sampler.py:56-63
        hi = lo + 100000
        ingest.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi],
                         add_reverse=args.undirected)
    build_s = time.time() - t0
    sampler = gnnflow_amd.TemporalSampler(graph, ctx.fanouts, args.strategy, seed=1234)
    if kind == "hash":
        # every rank owns a shard; per layer the roots are bucketed by owner, requests and
        # replies travel as equal-split all-to-alls, and the rank's own share is sampled meanwhile
        slack = args.part_slack
        if slack is None and ctx.world == 1 and always_exchange:
            slack = 2.0     # one rank over RCCL prices the chain of an 8-GPU run: its capacity
        sampler = DevicePartitionedSampler(sampler, slack=slack,
                                           slot_roots=3 * args.batch_size,
                                           always_exchange=always_exchange,
                                           lanes=args.part_lanes,
                                           chain_samples=args.part_chain)
    ctx.live_sampler = sampler      # (a rung that hangs aborts its communicators: main())
    return graph, sampler, build_s


def time_leg(ctx, sampler, cache, main_leg, min_seconds, min_replays, probe=None):
    """Warm-up, calibration, then the timed region over this rank's share of the replay.
    Returns a dict with elapsed (max over ranks), edges (sum over ranks), this rank's edges,
    timed_steps, repeats and the pipeline."""
    import torch
    import torch.distributed as dist
    from gnnflow_amd.pipeline import ReplayPipeline
    args, world, nb = ctx.args, ctx.world, ctx.nb
    depth = args.pipeline_depth
    if depth is None:
        lanes = getattr(sampler, "lanes", 1)
        # partitioned sampler: a chain carries up to `chain_samples` batches; per lane one
        # chain in flight and half a chain being gathered
        chain = getattr(sampler, "chain_samples", 1)
        # (replica: ReplayPipeline's own default, 3 with two sampling lanes)
        depth = None if lanes == 1 and chain == 1 else max(3 * lanes, (3 * lanes * chain) // 2,
                                                           2 * chain)
    pipe = ReplayPipeline(sampler, cache, ctx.dev_batches, ctx.dev,
                          pipelined=not args.no_pipeline, depth=depth)
    ctx.live_pipe = pipe
    _run = pipe.run

    def counted_run(first, count, on_step=None):     # steps issued so far (per-step averages
        ctx.steps_issued = getattr(ctx, "steps_issued", 0) + max(count, 0)   # of cumulative counters)
        return _run(first, count, on_step)
    pipe.run = counted_run

    def reduce(value, op):
        if world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=ctx.dev if ctx.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t)

    pipe.run(0, args.warmup if main_leg else min(args.warmup, nb))
    steps = args.steps if main_leg else nb
    repeats = args.repeats if (main_leg and args.repeats > 0) else 0
    if repeats == 0:
        # calibration: the mean step time of a whole replay (15 % on top: the timed region
        # must not end up just short of --min-seconds)
        def timed(count):
            ctx.barrier()
            t0 = time.perf_counter()
            pipe.run(0, count)
            ctx.barrier()
            return reduce((time.perf_counter() - t0) / count, dist.ReduceOp.MAX)
        t_step = timed(min(nb, 64))          # a first look (the same value on every rank)
        if t_step * nb < 0.3:
            for _ in range(2):  # the first replay still grows buffers: the second one counts
                t_step = timed(nb)
        # else: slow steps (ranks sharing a GPU over a staged exchange): --min-replays decides
        need = max(min_replays * nb, 1.15 * min_seconds / max(t_step, 1e-7))
        repeats = max(1, -(-int(need) // max(steps, 1)))
    timed_steps = steps * repeats
    # the timed region replays from the first batch on a freshly initialised cache
    if cache is not None:
        cache.init_cache()
        cache.algorithmic_bytes = 0
        cache.rows_moved = 0
    if main_leg:
        ctx.lib.gf_profile_reset()
        # HIP events on the gather launches of the timed region, on their stream.  Every 61st
        # launch is timed: an event pair costs stream time, which at ~30 us per step distorts
        # both the throughput measured in the same pass and the launches it times (same box,
        # round 5: gather 13.4 us with every launch timed at 42 us per step, 13.9 at every 5th,
        # 15.3 at every 17th, 14.3 at every 64th at 28.1 us per step; rocprofv3: 12.7-13.1).
        ctx.lib.gf_profile_set_stride(args.event_stride)
        # ... and stream events around the LRU update behind every 61st gather
        ctx.lib.gf_profile_enable((1 << ctx.capi.PROFILE_SLOTS["gather"]) |
                                  (1 << ctx.capi.PROFILE_SLOTS["lru"]))
    probe0 = probe() if probe else None     # (may synchronise: outside the timed region)
    acc = {"edges": 0}

    def account(_i, mfgs):
        for mfg in mfgs:
            for b in mfg:
                acc["edges"] += b.num_edges()

    ctx.barrier()
    t0 = time.perf_counter()
    pipe.run(0, timed_steps, account)
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    if main_leg:
        ctx.lib.gf_profile_enable(0)
        ctx.lib.gf_profile_set_stride(1)
    if world > 1:
        elapsed_max = reduce(elapsed, dist.ReduceOp.MAX)
        edges_all = reduce(acc["edges"], dist.ReduceOp.SUM)
    else:
        elapsed_max, edges_all = elapsed, float(acc["edges"])
    return dict(elapsed=elapsed_max, edges_all=edges_all, edges=acc["edges"], steps=steps,
                repeats=repeats, timed_steps=timed_steps, pipe=pipe, probe0=probe0,
                probe1=probe() if probe else None)


def agree(ctx, ok):
    """min over the ranks of a success flag (the ranks must take the same branch)."""
    if ctx.world == 1:
        return bool(ok)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32,
                     device=ctx.dev if ctx.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t))


GIVE_UP = 75     # exit status of a rung's worker that hands over to the next rung


def rung_seconds(name, rung, defaults):
    """A time limit of the ladder from the environment: one value for every rung, or one per
    rung, comma-separated ("15,150": rung 0 gives up after 15 s, rung 1 after 150)."""
    v = os.environ.get(name)
    if not v:
        return float(defaults[min(rung, len(defaults) - 1)])
    parts = [p for p in v.split(",") if p.strip()]
    return float(parts[min(rung, len(parts) - 1)])


class RungBarrier:
    """The supervisors of one run leave a rung TOGETHER.  Without this a rank whose worker dies
    early starts the next rung at once — on the next port — while its peers sit in the old rung
    until their watchdogs fire; its new worker then waits in a rendezvous nobody joins, is killed
    before the others arrive, and from then on the ranks are on different rungs for good (rungs 1
    and 2 are never really tried).  All ranks of a run are on one node (the driver's contract:
    --nnodes=1) and children of one launcher, so the barrier is a directory under the temp dir
    named after the launcher's pid and the run's base port: a rank that has left rung r (whatever
    the outcome) drops a file there and waits until every rank's file is there — or until the
    longest a peer's rung can still last, after which it goes on alone, as before.  The
    supervisors never touch the GPU."""

    def __init__(self, rank, world, base_port):
        import tempfile
        self.rank, self.world = rank, world
        self.dir = os.path.join(tempfile.gettempdir(), "gnnflow_ladder_{}_{}".format(
            os.getppid(), base_port))
        try:
            os.makedirs(self.dir, exist_ok=True)
            for name in os.listdir(self.dir):            # this rank's files of an earlier run
                if name.endswith("_rank{}".format(rank)):
                    os.unlink(os.path.join(self.dir, name))
        except OSError:
            self.dir = None

    def leave(self, rung, status, wait_s):
        """Announces that this rank's worker of `rung` is gone; returns how many ranks had left
        the rung when the wait ended (world: everybody)."""
        if self.dir is None:
            return 0
        try:
            with open(os.path.join(self.dir, "left_{}_rank{}".format(rung, self.rank)), "w") as f:
                f.write(str(status))
        except OSError:
            return 0
        deadline = time.time() + max(wait_s, 0.0)
        while True:
            try:
                n = sum(1 for name in os.listdir(self.dir) if name.startswith("left_{}_".format(rung)))
            except OSError:
                return 0
            if n >= self.world or time.time() >= deadline:
                return n
            time.sleep(0.05)

    def close(self):
        if self.dir is None:
            return
        try:
            for name in os.listdir(self.dir):
                if name.endswith("_rank{}".format(self.rank)):
                    os.unlink(os.path.join(self.dir, name))
            os.rmdir(self.dir)            # (the last rank out succeeds)
        except OSError:
            pass


def supervise(args, argv):
    """One RANK of an N > 1 run (started by torchrun or by launch_ranks): this process never
    touches the GPU.  It runs the rungs of the ladder (RUNGS) one after the other, each as a
    WORKER process of its own on a new rendezvous port, until one of them prints the line:
      * a worker that finishes prints the ONE JSON line (rank 0) and exits 0;
      * a worker whose hash-partitioned loop RAISES (the ranks agree through an all-reduce) or
        HANGS (its watchdog fires after GNNFLOW_HASH_MAIN_TIMEOUT: RUNG_MAIN_SECONDS — the
        whole ladder fits the driver's 600 s) leaves a ladder entry and exits
        GIVE_UP; one that does not even do that is killed.  Either way the process is GONE —
        and with it its communicators and whatever kernels it still had on the GPU — before
        the next rung's worker starts: no figure shares the GPU with an abandoned run.
    RCCL between two ranks has never run on this code before the driver's first scaling run;
    the likeliest first failure is a collective that some rank never joins."""
    import subprocess
    import tempfile
    import threading
    rank = int(os.environ.get("RANK", "0"))
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    barrier = RungBarrier(rank, int(os.environ.get("WORLD_SIZE", "1")), base_port)
    history, rc, lines = [], 1, []
    for rung in range(len(RUNGS)):
        fd, path = tempfile.mkstemp(prefix="gnnflow_rung{}_rank{}_".format(rung, rank))
        os.close(fd)
        os.unlink(path)
        env = dict(os.environ)
        env.update(GNNFLOW_BENCH_WORKER="1", GNNFLOW_BENCH_RUNG=str(rung),
                   GNNFLOW_BENCH_LADDER=json.dumps(history), GNNFLOW_BENCH_LADDER_FILE=path,
                   MASTER_PORT=str(base_port + 17 * (rung + 1)))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)   # the worker's rank 0 hosts its own store
        limit = None
        if rung < len(RUNGS) - 1:
            # the worker's own watchdog covers its main loop; this covers everything else
            # (set-up included) should the worker be too wedged to exit by itself
            # (the first worker of a fresh box also pages torch in: 1-2 minutes)
            limit = rung_seconds("GNNFLOW_HASH_MAIN_TIMEOUT", rung, RUNG_MAIN_SECONDS) + \
                rung_seconds("GNNFLOW_RUNG_SETUP_ALLOWANCE", rung, RUNG_SETUP_SECONDS)
        t0 = time.time()
        kid = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                               stdin=subprocess.DEVNULL,
                               stdout=subprocess.PIPE if rank == 0 else sys.stderr)
        lines = []

        def relay(k=kid, out=lines):
            for raw in k.stdout:
                line = raw.decode(errors="replace")
                (out.append if line.startswith("{") else sys.stderr.write)(line)
        reader = None
        if rank == 0:
            reader = threading.Thread(target=relay, daemon=True)
            reader.start()
        killed = False
        try:
            rc = kid.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            killed = True
            kid.kill()
            rc = kid.wait()
        if reader is not None:
            reader.join(timeout=10.0)
        entry = None
        try:
            with open(path) as f:
                entry = json.load(f)
            os.unlink(path)
        except (OSError, ValueError):
            pass
        if rung < len(RUNGS) - 1:
            # leave the rung together with the peers: one of them may still be inside its
            # watchdog's time (they started the rung when this rank did)
            left = barrier.leave(rung, rc, (limit - (time.time() - t0) + 15.0) if rc != 0 else 0.0)
        if rc == 0:
            break
        if rung == len(RUNGS) - 1:
            break                               # the last rung's failure is the run's failure
        if entry is None:
            entry = {"rung": rung, "arrangement": RUNGS[rung][0], "hung": bool(killed),
                     "error": "the worker was killed after {:.0f} s without a word".format(
                         time.time() - t0) if killed else
                     "the worker exited with status {}".format(rc)}
        entry["worker_killed"] = bool(killed)
        entry["seconds"] = round(time.time() - t0, 1)
        entry["ranks_left_together"] = left
        history.append(entry)
        sys.stderr.write("bench.py rank {}: rung {} ({}) is over after {:.0f} s: {}; starting rung "
                         "{} ({})\n".format(rank, rung, RUNGS[rung][0], time.time() - t0,
                                            entry["error"], rung + 1, RUNGS[rung + 1][0]))
    barrier.close()
    if rank == 0:
        if lines:
            sys.stdout.write(lines[-1])         # exactly one line
        else:
            sys.stdout.write(json.dumps({
                "metric": "sampled_edges_per_s", "value": 0.0, "unit": "edges/s",
                "n_gpus": args.gpus, "ladder": history,
                "error": "no rung printed a record; last exit status {}".format(rc)}) + "\n")
            rc = rc or 1
        sys.stdout.flush()
    return rc if rc >= 0 else 128 - rc


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "GNNFLOW_BENCH_WORKER" not in os.environ \
            and args.partition in (None, "hash"):
        sys.exit(supervise(args, argv))
    fake = os.environ.get("GNNFLOW_BENCH_FAKE_WORKER")
    if fake and "GNNFLOW_BENCH_WORKER" in os.environ:
        # test hook (tests/test_bench_launcher.py, no GPU): this worker only ACTS its rung's
        # outcome — "ok" prints a line, "giveup" leaves a ladder entry and exits GIVE_UP, "hang"
        # never exits, "die" exits 9 — so that launcher + supervisor + ladder run end to end
        rung = int(os.environ["GNNFLOW_BENCH_RUNG"])
        # (GNNFLOW_BENCH_FAKE_WORKER_RANK<r>: that rank acts differently from the others)
        fake = os.environ.get("GNNFLOW_BENCH_FAKE_WORKER_RANK" + os.environ.get("RANK", "0"), fake)
        acts = fake.split(",")
        act = acts[min(rung, len(acts) - 1)]
        if act == "hang":
            time.sleep(1e6)
        if act == "die":
            os._exit(9)
        if act == "giveup":
            with open(os.environ["GNNFLOW_BENCH_LADDER_FILE"], "w") as f:
                json.dump({"rung": rung, "arrangement": RUNGS[rung][0], "hung": True,
                           "error": "acted: gave up"}, f)
            os._exit(GIVE_UP)
        if os.environ.get("RANK", "0") == "0":
            print(json.dumps({"metric": "sampled_edges_per_s", "value": 1.0, "unit": "edges/s",
                              "n_gpus": args.gpus, "master_port": os.environ.get("MASTER_PORT"),
                              "ladder": {"rung": rung, "arrangement": RUNGS[rung][0],
                                         "tried_before": json.loads(
                                             os.environ.get("GNNFLOW_BENCH_LADDER", "[]"))}}))
        sys.exit(0)
    # stdout carries exactly ONE line, the JSON record: everything else that libraries print
    # there (RCCL's version banner on communicator creation, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        if not getattr(main, "superseded", False):     # a fresh set of ranks took over (below)
            os.write(json_fd, (json.dumps(record) + "\n").encode())
    main.emit = emit
    ctx = Ctx()
    ctx.args = args
    rank = ctx.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # test hooks (single-GPU box): several ranks on one device, gloo for the bookkeeping
    # collectives.  The driver's multi-GPU runs use neither.
    if "GNNFLOW_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["GNNFLOW_BENCH_DEVICE"])
    ctx.local_rank = local_rank
    backend = ctx.backend = os.environ.get("GNNFLOW_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = ctx.dev = torch.device("cuda", local_rank)
    if world > 1 or args.partition == "hash" or args.always_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, "--gpus must match the launched world size"
    if world > 1:
        # a collective that never completes (RCCL between ranks has not run on this code before
        # the first scaling run) must not hang the job: after GNNFLOW_BENCH_TIMEOUT seconds rank
        # 0 prints an error record and every rank exits non-zero
        import threading
        limit = float(os.environ.get("GNNFLOW_BENCH_TIMEOUT", "900"))
        main.done = threading.Event()

        def bench_watchdog():
            if not main.done.wait(limit):
                if rank == 0:
                    emit({"metric": "sampled_edges_per_s", "value": 0.0, "unit": "edges/s",
                          "n_gpus": world, "error": "timed out after {} s".format(limit)})
                os._exit(3)
        threading.Thread(target=bench_watchdog, daemon=True).start()
    if args.partition is None:
        args.partition = "hash" if world > 1 else "replica"

    # Host threads per rank in the pipelined loop: Python + two polling enqueue lanes.  When the
    # ranks of this node have fewer than 4 usable cores each (a CPU-quota'd container), fall
    # back to one sleeping lane: slower steps, but no rank starves another.
    per_rank = usable_cores(cap=1 << 20) / max(int(os.environ.get("LOCAL_WORLD_SIZE", world)), 1)
    if per_rank < 4:
        os.environ.setdefault("GNNFLOW_ENQUEUE_LANES", "1")
        os.environ.setdefault("GNNFLOW_ENQUEUE_SPIN_US", "0")
    import gnnflow_amd  # noqa: F401
    from gnnflow_amd import _capi, synthetic
    from gnnflow_amd.cache import LRUCache
    from gnnflow_amd.utils import bind_to_device_cpus
    # one process per GPU, on the CPUs of that GPU's NUMA node (the launches are issued from
    # there; gnnflow_amd/utils.py).  The CPU baseline below runs inside the same mask.
    bound = bind_to_device_cpus(local_rank)

    lib = ctx.lib = _capi.load()
    ctx.capi = _capi
    fanouts = ctx.fanouts = [int(x) for x in args.fanouts.split(",")]
    g = ctx.g = synthetic.reddit_like(seed=42)
    d_e, d_n = synthetic.REDDIT["dim_edge"], synthetic.REDDIT["dim_node"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    ctx.barrier = barrier

    gen = torch.Generator(device=dev).manual_seed(42)
    edge_feats = torch.rand((g["num_edges"], d_e), generator=gen, device=dev)
    node_feats = torch.rand((g["num_nodes"], d_n), generator=gen, device=dev)
    cache = None
    sharded = (not args.sample_only) and args.shard_features and args.partition == "hash"
    if sharded:
        from gnnflow_amd.dist import FeatureShards, ShardedFeatures
        import numpy as np
        shards = ShardedFeatures(
            node=FeatureShards.from_full(node_feats, np.arange(g["num_nodes"]), rank, world, dev),
            edge=FeatureShards.from_full(edge_feats, g["src"], rank, world, dev))
        shards.always_exchange = args.always_exchange
        cache = LRUCache(args.cache_ratio, args.cache_ratio, g["num_nodes"], g["num_edges"], dev,
                         None, None, d_n, d_e, kvstore_client=shards, distributed=True)
        cache.init_cache()
    elif not args.sample_only:
        cache = LRUCache(args.cache_ratio, args.cache_ratio, g["num_nodes"], g["num_edges"],
                         dev, node_feats, edge_feats, d_n, d_e,
                         feature_placement=args.feature_placement)
        cache.init_cache()

    # this rank's share of the chronological replay, resident in HBM
    batches = list(synthetic.replay_batches(g, args.batch_size, seed=42))
    nb = ctx.nb = len(batches) // world   # same on every rank (a longer share loses its tail)
    mine = (batches[rank::world] if world > 1 else batches)[:nb]
    ctx.dev_batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev),
                        torch.from_numpy(e).to(dev)) for r, t, e in mine]

    # Software pipeline (the reference's training loop prefetches the next batch's
    # sample() on a Python thread, scripts/offline_edge_prediction.py:343-346,397-399):
    # batch i+1's sample() is enqueued on a side HIP stream before batch i's
    # fetch_feature() is issued on the main stream.  Every batch goes through the same
    # calls; nothing is skipped or cached.  gnnflow_amd/pipeline.py, parity-tested.
    #
    # The main loop.  A failing hash-partitioned loop (it runs over RCCL between ranks for the
    # first time in the driver's scaling run) must not cost the line: every rank then times
    # the replica loop instead and the record says so.
    def staging_probe():
        import ctypes as C
        busy, jobs = C.c_double(0), C.c_uint64(0)
        lib.gf_worker_stats(C.byref(busy), C.byref(jobs))
        st = cache.staging_state()
        st["_worker"] = (busy.value, jobs.value)
        return st

    def run_main(kind):
        graph, sampler, build_s = build_leg(ctx, kind)
        def worker_probe():      # issuing time of the library's enqueue threads so far
            import ctypes as C
            busy, jobs = C.c_double(0), C.c_uint64(0)
            lib.gf_worker_stats(C.byref(busy), C.byref(jobs))
            return {"_worker_only": (busy.value, jobs.value)}
        res = time_leg(ctx, sampler, cache, True, args.min_seconds, args.min_replays,
                       probe=staging_probe if (cache is not None and cache.staging)
                       else worker_probe)
        res.update(graph=graph, sampler=sampler, build_s=build_s, kind=kind)
        return res

    # ---- the N > 1 ladder (this process is the WORKER of one rung: supervise() above) ---------
    rung = int(os.environ.get("GNNFLOW_BENCH_RUNG", "0"))
    history = json.loads(os.environ.get("GNNFLOW_BENCH_LADDER", "[]"))
    laddered = "GNNFLOW_BENCH_WORKER" in os.environ and world > 1
    last_rung = len(RUNGS) - 1
    if not laddered:
        rung = last_rung if args.partition == "replica" else 0
    if laddered and rung == 1:
        args.part_lanes = 1
    if laddered and rung == 2:
        args.part_lanes, args.part_chain, args.part_slack = 1, 1, 2.0
    if laddered and rung >= last_rung:
        args.partition = "replica"
    main_kind, res = args.partition, None
    ctx.rung, ctx.history = rung, history

    def give_up(why, hung):
        """This rung is over on this rank: leave the ladder entry for the supervisor and exit
        with GIVE_UP — the supervisor starts the next rung once this process (and with it every
        kernel it still has on the GPU) is gone.  Never returns."""
        main.superseded = True
        sys.stderr.write("bench.py rank {}: rung {} ({}) gave up: {}\n".format(
            rank, rung, RUNGS[rung][0], why))
        entry = {"rung": rung, "arrangement": RUNGS[rung][0], "error": why, "hung": bool(hung)}
        path = os.environ.get("GNNFLOW_BENCH_LADDER_FILE")
        if path:
            try:
                with open(path, "w") as f:
                    json.dump(entry, f)
            except OSError:
                pass
        os._exit(GIVE_UP)

    hang_guard = None
    if laddered and rung < last_rung:
        import threading
        hang_guard = threading.Event()
        hang_limit = rung_seconds("GNNFLOW_HASH_MAIN_TIMEOUT", rung, RUNG_MAIN_SECONDS)

        def hash_hang_watchdog():
            if hang_guard.wait(hang_limit):
                return
            give_up("the hash-partitioned loop did not finish within {:.0f} s (a collective "
                    "that some rank never joined?)".format(hang_limit), hung=True)
        threading.Thread(target=hash_hang_watchdog, daemon=True).start()

    def hooked(name, rungs_default):
        # test hooks: GNNFLOW_BENCH_FAIL_HASH / GNNFLOW_BENCH_HANG_HASH act on the hash rungs
        # named by GNNFLOW_BENCH_FAIL_RUNGS / GNNFLOW_BENCH_HANG_RUNGS
        v = os.environ.get("GNNFLOW_BENCH_{}_HASH".format(name))
        on = os.environ.get("GNNFLOW_BENCH_{}_RUNGS".format(name), rungs_default).split(",")
        return v if (v is not None and main_kind == "hash" and str(rung) in on) else None
    try:
        if hooked("FAIL", "0,1,2"):
            raise RuntimeError("GNNFLOW_BENCH_FAIL_HASH is set")
        hang = hooked("HANG", "0,1,2")
        if hang is not None and hang in ("all", str(rank)):
            time.sleep(1e6)
        res = run_main(main_kind)
        failure = None
    except Exception as e:                      # noqa: BLE001 — reported in the record
        import traceback
        traceback.print_exc()
        failure = "{}: {}".format(type(e).__name__, e)
    agreed = agree(ctx, failure is None)
    if hang_guard is not None:
        hang_guard.set()
    if getattr(main, "superseded", False):      # the next rung's ranks have the job now
        time.sleep(1e6)
    if not agreed:
        failure = failure or "the loop failed on another rank"
        if laddered and rung < last_rung:
            give_up(failure, hung=False)
        if rank == 0:
            emit({"metric": "sampled_edges_per_s", "value": 0.0, "unit": "edges/s",
                  "n_gpus": world, "error": failure, "ladder": history})
        if getattr(main, "done", None) is not None:
            main.done.set()
        os._exit(1)
    sampler, pipe = res["sampler"], res["pipe"]
    elapsed_max, edges_all, edges = res["elapsed"], res["edges_all"], res["edges"]
    repeats, timed_steps = res["repeats"], res["timed_steps"]
    pipelined = pipe.pipelined
    args.partition = main_kind
    # rows x (8 B id + 2 x 4 x d B row read + write), summed by the cache
    # over the rows its gather launches really moved (a block served as a prefix of
    # another block's rows moves nothing and counts nothing)
    gather_bytes = cache.algorithmic_bytes if cache is not None else 0
    rows_moved = cache.rows_moved if cache is not None else 0

    import ctypes as C
    g_ms, g_n, g_all = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    lib.gf_profile_get(_capi.PROFILE_SLOTS["gather"], C.byref(g_ms), C.byref(g_n))
    lib.gf_profile_launches(_capi.PROFILE_SLOTS["gather"], C.byref(g_all))

    share = "" if world == 1 else " of this rank's share (every {}th batch)".format(world)
    window = "batches 0..{}{} in chronological order, {:.2f} times over".format(
        nb - 1, share, timed_steps / nb) if timed_steps >= nb else \
        "batches 0..{}{} in chronological order".format(timed_steps - 1, share)
    out = {
        "metric": "sampled_edges_per_s",
        "value": edges_all / elapsed_max,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "repeats": repeats,
        "timed_steps": timed_steps,
        "timed_seconds": elapsed_max,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed_max / timed_steps,
        # the harness's own unit (benchmarks/benchmark_sampler.py: target edges per second of the
        # replayed stream): batch size x ranks / step time
        "target_edges_per_s": args.batch_size * world * timed_steps / elapsed_max,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64/f32",
        "data": "synthetic",
        "config": {
            "workload": "REDDIT-shaped synthetic (10984 nodes, 672447 edges, directed), "
                        "2-layer fanout [{}] {} sampling, batch {} ({} roots), "
                        "LRUCache ratio {} + 172-d edge/node feature gather; "
                        "step = sample() + fetch_feature(); timed region = {} windows of {} "
                        "consecutive steps: {}".format(
                            args.fanouts, args.strategy, args.batch_size, 3 * args.batch_size,
                            args.cache_ratio, repeats, args.steps, window)
            if cache is not None else
            "REDDIT-shaped synthetic, 2-layer fanout [{}] {} sampling, batch {}; "
            "step = sample() only; timed region = {} windows of {} consecutive steps: "
            "{}".format(args.fanouts, args.strategy, args.batch_size, repeats, args.steps,
                        window),
            "batch_size": args.batch_size,
            "roots_per_step": 3 * args.batch_size,
            "timed_steps": timed_steps,
            "timed_seconds": elapsed_max,
            "edges_per_step": edges / max(timed_steps, 1),
            # feature rows gathered per step (node block + outer edge block + target rows; the
            # inner edge block is a prefix of the outer one's rows and moves nothing) — the CPU
            # baseline below moves the same rows
            "rows_per_step": (rows_moved / max(timed_steps, 1)) if cache is not None else 0,
            "feature_placement": args.feature_placement,
            "graph_build_s": round(res["build_s"], 3),
            "parallelism": "{}-dp{}".format(main_kind, world),
            "pipelined": bool(pipelined),
            "pipeline_depth": pipe.depth if pipelined else 0,
            "workload_key": workload_key(args),
            "cpus_bound_to_gpu_node": len(bound) if bound else None,
            "usable_cores_per_rank": per_rank,
            "host_logical_cpus": os.cpu_count(),
        },
    }
    if cache is not None and g_n.value:
        # dominant kernel by bytes: the fused feature gather (one launch per fetch round):
        # algorithmic bytes per launch (all launches) / average duration of the launches
        # that carried events (every event_stride-th one)
        n_launches = int(g_all.value)
        bytes_per_launch = gather_bytes / max(n_launches, 1)
        avg_us = 1e3 * g_ms.value / g_n.value
        achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
        lean = d_e % 4 == 0 and d_n % 4 == 0 and not any(
            k.lru_state()["queue_form"] for k in (cache._node, cache._edge) if k is not None)
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "gather_rows_any_kernel" if not lean else
            ("gather_rows_staged_kernel" if cache.staging else
             "gather_rows_mirror_kernel" if os.environ.get("GNNFLOW_CACHE_ROW_MIRROR") == "1"
             else "gather_rows_kernel"),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # PMC counters cannot be read inside this process: `traffic` stays null in the run,
            # and the committed rocprofv3 --pmc passes over THIS command line ride along as
            # `traffic_from_profile` when the arguments match (scripts/rocprof_pmc.sh; FETCH_SIZE /
            # WRITE_SIZE in separate passes, gfx950 FETCH x2 correction)
            "traffic": None,
            "launches": int(n_launches), "launches_timed": int(g_n.value),
            "avg_launch_us": avg_us,
            "algorithmic_bytes_per_launch": bytes_per_launch,
        }
        pmc = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
        if os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f)
            if rec.get("bench_args") == workload_key(args):
                out["roofline"]["traffic_from_profile"] = rec["hbm_traffic_bytes_per_dispatch"]
                out["roofline"]["traffic_source"] = "profiles/" + PMC_TRAFFIC_FILE
        out["cache_edge_ratio"] = float(cache.cache_edge_ratio)
        out["cache_node_ratio"] = float(cache.cache_node_ratio)
        if res.get("probe1") is not None and "_worker_only" in res["probe1"]:
            # how busy the enqueue threads (fetch launches; sampling launches) were: with the
            # kernels' times above it says whether the GPU or the host's issuing path sets the step
            w0, w1 = res["probe0"]["_worker_only"], res["probe1"]["_worker_only"]
            out["config"]["enqueue_threads_busy_us_per_step"] = (w1[0] - w0[0]) / max(timed_steps, 1)
        elif res.get("probe1") is not None:
            # host-resident tables: what crossed the host link (gnnflow/cache/cache.py:288-313,
            # 381-388 move every miss host -> pinned -> device inside fetch_feature)
            s0, s1 = res["probe0"], res["probe1"]
            w0, w1 = s0.pop("_worker"), s1.pop("_worker")
            steps_t = max(timed_steps, 1)
            pulled = sum(s1[k]["rows_pulled"] - s0[k]["rows_pulled"] for k in s1) / steps_t
            host = sum(s1[k]["rows_read_from_host"] - s0[k]["rows_read_from_host"]
                       for k in s1) / steps_t
            nbytes = (pulled + host) * 4.0 * d_e
            src = cache._edge.table.view(-1)[:1 << 26]
            dst = torch.empty_like(src, device=dev)
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            link = 3 * src.numel() * 4 / (time.perf_counter() - t0) / 1e9
            del dst
            out["staging"] = {
                "what": "tables in pinned host memory, LRUCache ratio {} in HBM, staging ring {} "
                        "generations x {} rows per kind ({:.0f} MB of HBM with its index); batch "
                        "i+1's misses pulled over the host link on a side stream beside "
                        "fetch_feature(batch i)".format(
                            args.cache_ratio, s1["edge"]["generations"],
                            s1["edge"]["rows_per_generation"],
                            sum(s1[k]["ring_bytes"] for k in s1) / 1e6),
                "rows_pulled_into_ring_per_step": pulled,
                "rows_read_from_host_by_gather_per_step": host,
                "host_link_bytes_per_step": nbytes,
                "host_link_GBps_over_the_step": nbytes / (elapsed_max / steps_t) / 1e9,
                "host_link_GBps_memcpy_256MB": link,
                "host_link_floor_us_per_step": nbytes / (link * 1e9) * 1e6,
                "generations_dropped": sum(s1[k]["dropped"] - s0[k]["dropped"] for k in s1),
                "issue_wait_us_per_step": sum(s1[k]["issue_wait_us"] - s0[k]["issue_wait_us"]
                                              for k in s1) / steps_t,
                "fetches_that_waited_for_a_pull": (s1["edge"]["stream_waits"] -
                                                   s0["edge"]["stream_waits"]) / steps_t,
                "enqueue_threads_busy_us_per_step": (w1[0] - w0[0]) / steps_t}
        # The other kernel of a step's fetch chain: the LRU list update (one launch:
        # lru_list_fused_kernel).  Algorithmic bytes per update: the list rewrite (per slot:
        # entry + mark read, entry + qpos written = 16 B, both caches), the block's rows (id +
        # claim = 12 B each) and the installed rows (2 x 4 x d + 24 B of map / slot_id each;
        # counted from the miss RATIOS, so repeated ids count twice: an upper bound).  A chain of
        # dependent hops, not a bandwidth kernel: the fraction says how far from one it is.
        l_ms, l_n = C.c_double(0), C.c_uint64(0)
        lib.gf_profile_get(_capi.PROFILE_SLOTS["lru"], C.byref(l_ms), C.byref(l_n))
        if l_n.value:
            rows = rows_moved / max(n_launches, 1)
            miss = 1.0 - 0.5 * (out["cache_edge_ratio"] + out["cache_node_ratio"])
            # which kernels the update really ran (gf_cache_lru_state + GNNFLOW_LRU_FUSED): the
            # byte model and the clock follow the form
            forms = [k.lru_state()["queue_form"] for k in (cache._node, cache._edge) if k is not None]
            fused = os.environ.get("GNNFLOW_LRU_FUSED", "1") != "0"
            installs = miss * rows * (8.0 * d_e + 24.0)
            if not any(forms):
                lru_bytes = 16.0 * (cache.edge_capacity + cache.node_capacity) + 12.0 * rows + installs
                kernel = "lru_list_fused_kernel" if fused else \
                    "lru_list_scan_kernel + lru_list_install_kernel"
                clock = "dispatch begin / end of the one launch (the clock rocprofv3 reads)" \
                    if fused else "stream events around the two launches"
            else:
                # queue form: no O(capacity) pass — per block row id + claim + slot + mark +
                # appended entry + qpos (28 B), plus the installed rows
                lru_bytes = 28.0 * rows + installs
                kernel = "lru_list_scan_kernel + lru_queue_walk_kernel + lru_queue_install_kernel" + \
                    ("" if all(forms) else " (+ lru_list_fused_kernel of the list-form cache)")
                clock = "stream events around the update's launches"
            lru_us = 1e3 * l_ms.value / l_n.value
            out["roofline_lru"] = {
                "bound": "hbm", "kernel": kernel, "unit": "GB/s",
                "algorithmic_bytes_per_launch": lru_bytes, "avg_launch_us": lru_us,
                "achieved": lru_bytes / (lru_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                "frac": lru_bytes / (lru_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "launches_timed": int(l_n.value), "clock": clock,
                "note": "latency chain of ~7 dependent hops, not a bandwidth kernel: see DESIGN 3.4"}

    if main_kind == "hash":
        out["config"]["exchange"] = exchange_note(sampler, world, backend)
        out["config"]["features"] = (
            "sharded by owner: {} rows pulled, {} count read-backs in {} steps".format(
                cache._shards.rows_pulled, cache._shards.host_syncs, timed_steps + args.warmup)
            if cache is not None and cache.distributed else "replicated on every GPU")
    if world > 1:
        out["ladder"] = {"rung": rung, "arrangement": RUNGS[rung][0], "what": RUNGS[rung][1],
                         "tried_before": history}
        # (every rung runs in processes of its own and the next one starts only when they are
        # gone — killed if need be — so no abandoned kernel shares the GPU with this figure)
    if main_kind == "hash" and world > 1:
        out["multi_gpu"] = multi_gpu_record(ctx, sampler, cache)
    # (before the hash legs: their lanes, clones and communicators create streams, and which
    # hardware queue a stream lands on depends on how many the process created before it)
    if rank == 0 and world == 1 and main_kind == "replica" and cache is not None \
            and not args.no_placement_legs and not sharded:
        placement_legs(ctx, out, res["sampler"], node_feats, edge_feats)
    if history and main_kind == "replica":
        out["hash_partition"] = {"error": history[0]["error"], "ladder": history,
                                 "fallback": "replica loop timed instead"}
    elif cache is not None and not args.no_second_leg:
        # The other kind of graph in the same line, over the same batches, cache and pipeline:
        # the hash-partitioned one when the main loop ran on replicas (at P = 1 every root is
        # the rank's own: the figure then prices the bucketing / fixed-slot / merge kernels
        # against the plain sampler above), the replicas when the main loop ran on the
        # partitioned graph (N > 1).
        other = "hash" if main_kind == "replica" else "replica"
        second_leg.pending_line = out
        out["hash_partition" if other == "hash" else "replica"] = second_leg(ctx, other, cache)
        if other == "replica" and out["replica"].get("value"):
            # weak-scaling efficiency can only be formed against N = 1 by the driver; what this
            # run can say itself: what the exchange costs against replicas on the same GPUs
            out["efficiency_vs_replica_same_run"] = out["value"] / out["replica"]["value"]
        if world == 1 and other == "hash" and not args.always_exchange and backend == "nccl":
            # ... and the chain the ranks of an N > 1 run really issue — slotted exchange, lanes,
            # two samples per chain — with every message travelling through RCCL to this rank
            # itself: what the multi-rank path costs before any real peer exists
            out["hash_partition_over_rccl_one_rank"] = second_leg(ctx, "hash", cache,
                                                                  always_exchange=True)
    if main_kind == "hash" and world > 1 and "multi_gpu" in out:
        # last of all, and guarded: a collective micro-measurement on the lanes' communicators
        guarded_aux(ctx, out, "all_to_all_measurement", 45.0,
                    lambda: all_to_all_record(ctx, sampler, out["multi_gpu"]))

    if args.breakdown and rank == 0:
        lib.gf_profile_reset()
        lib.gf_profile_enable(0x1F)
        pipe.run(0, min(timed_steps, 200))
        torch.cuda.synchronize()
        lib.gf_profile_enable(0)
        bd = {}
        for name, slot in _capi.PROFILE_SLOTS.items():
            ms, n = C.c_double(0), C.c_uint64(0)
            lib.gf_profile_get(slot, C.byref(ms), C.byref(n))
            bd[name] = {"total_ms": ms.value, "intervals": int(n.value)}
        out["kernel_breakdown_200_steps"] = bd

    if rank == 0 and world == 1 and not args.no_config3:
        out["config3"] = config3_leg(args, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(g, batches[:min(timed_steps, nb)], fanouts, args,
                                           edge_feats.cpu().numpy(), node_feats.cpu().numpy(),
                                           cache is not None)
    if rank == 0:
        emit(out)
    if getattr(main, "done", None) is not None:
        main.done.set()
    if world > 1:
        # the record is out: a teardown that hangs (communicators being destroyed while a peer
        # has already left) must not turn the run into a failure
        import threading
        t = threading.Timer(30.0, lambda: os._exit(0))
        t.daemon = True
        t.start()
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if world > 1:
        sys.stderr.flush()
        os._exit(0)     # no interpreter-exit finalisers racing the other ranks' teardown


def config3_leg(args, dev):
    """BASELINE configs[2] in the same line: uniform sampling on the synthetic power-law graph
    (10 M nodes / 200 M edges; 7.2 GB of edge store, far beyond the Infinity Cache) at batch
    600 ... 600 000.  scripts/config3_bench.py does the work; skipped on a GPU with < 64 GB."""
    import torch
    try:
        _free, total = torch.cuda.mem_get_info(dev)
        if total < (64 << 30):
            return {"skipped": "GPU has {:.0f} GB of memory (< 64 GB)".format(total / 2**30)}
        from scripts import config3_bench
        return config3_bench.sweep(
            args.config3_nodes, args.config3_edges,
            [int(b) for b in args.config3_batches.split(",")], ["uniform"], dev, reps=5)
    except Exception as e:       # noqa: BLE001 — the headline must survive this leg
        import traceback
        traceback.print_exc()
        return {"error": "{}: {}".format(type(e).__name__, e)}


def exchange_note(sampler, world, backend):
    if world == 1 and not getattr(sampler, "_always_exchange", False):
        return "none (one rank: every root is its own)"
    via = "RCCL" if backend == "nccl" else backend
    lanes = getattr(sampler, "lanes", 1)
    lane_note = "; {} sampling lanes (stream + workspace + communicator each), consecutive " \
                "batches round-robin".format(lanes) if lanes > 1 else ""
    try:
        import ctypes as C
        from gnnflow_amd import _capi
        n = C.c_uint64(0)
        _capi.load().gf_debug_merge_recounts(C.byref(n))
        if n.value:
            lane_note += "; {} merge tiles recounted by a look-back that stopped waiting " \
                         "(oversubscribed GPU)".format(n.value)
    except Exception:      # noqa: BLE001 — diagnostics only
        pass
    if getattr(sampler, "pairs", 0):
        lane_note += "; {} chains carried {:.2f} samples each on average (shared launches " \
                     "and exchanges), {}-byte reply slots".format(
                         sampler.pairs, getattr(sampler, "chained", 0) / sampler.pairs,
                         12 if getattr(sampler, "_narrow", False) else 24)
    comm = getattr(sampler, "_comm", None)
    if getattr(sampler, "_slack", 0) > 0 and comm is not None and comm.transport == "ipc":
        return ("2 equal-split exchanges per layer over the library's hipIpc transport (ranks "
                "sharing a GPU; host-synchronising: a test transport) — one native call per "
                "sample, slot capacity {} x the even share; {} overflowed samples redone{}".format(
                    sampler._slack, sampler.overflows, lane_note))
    if getattr(sampler, "_slack", 0) > 0 and comm is not None:
        return ("2 equal-split all-to-alls per layer over the library's own RCCL communicator "
                "(one native call per sample, issued by the enqueue thread; slot capacity {} x "
                "the even share; exchanges {}; no host sync inside a sample; {} overflowed "
                "samples redone{})".format(
                    sampler._slack, "on the communicator's stream, requests overlapped with "
                    "the own share" if sampler._overlap else "in the sampling stream",
                    sampler.overflows, lane_note))
    if getattr(sampler, "_slack", 0) > 0:
        return ("2 equal-split all-to-alls per layer over {} (slot capacity {} x the even share, "
                "no host sync inside a sample; {} overflowed samples redone)".format(
                    via, sampler._slack, sampler.overflows))
    return "2 all-to-all-v per layer over {} (one host sync per layer)".format(via)


def placement_legs(ctx, out, sampler, node_feats, edge_feats):
    """The other feature placements and the sampler alone, beside the headline (N = 1, replica
    main loop), over the same replay and the same pipelined loop (ReplayPipeline.run, the function
    tests/test_gpu_pipeline_parity.py checks in each of these arrangements):
      sample_only        sample() alone — what benchmarks/benchmark_sampler.py:71-87 times and
                         what north_star's ">= 10x the reference CPU sampler" is written on;
      cache_free_device  tables in HBM, cache ratio 0: the gather without any LRU bookkeeping —
                         the headline minus this is what the reference's cache protocol costs on
                         a placement where a hit and a miss read the same memory;
      pinned_placement   the reference's placement (gnnflow/cache/cache.py:288-313,381-388,
                         gnnflow/utils.py:284-297): tables in pinned HOST memory, LRU 0.2 in HBM,
                         batch i+1's misses pulled over the host link into the staging ring on a
                         side stream while batch i is fetched; rows that crossed the link, bytes
                         and GB/s per step ride along."""
    import torch
    from gnnflow_amd import synthetic
    from gnnflow_amd.cache import LRUCache
    args, g, dev = ctx.args, ctx.g, ctx.dev
    d_e, d_n = synthetic.REDDIT["dim_edge"], synthetic.REDDIT["dim_node"]
    secs = min(args.min_seconds, 1.0)

    def line(res):
        ms = 1e3 * res["elapsed"] / res["timed_steps"]
        return {"value": res["edges_all"] / res["elapsed"], "unit": "edges/s", "ms_per_step": ms,
                "target_edges_per_s": args.batch_size / (ms * 1e-3),
                "steps": res["steps"], "repeats": res["repeats"],
                "timed_seconds": res["elapsed"], "pipelined": bool(res["pipe"].pipelined)}

    def sample_only():
        rec = line(time_leg(ctx, sampler, None, False, secs, 1.0))
        rec["what"] = "sample() alone, {} samples in flight over {} lanes".format(
            max(2 * len(ctx.live_pipe.lanes), ctx.live_pipe.depth), len(ctx.live_pipe.lanes))
        return rec

    def cache_free_device():
        cache = LRUCache(0, 0, g["num_nodes"], g["num_edges"], dev, node_feats, edge_feats,
                         d_n, d_e, feature_placement="device")
        cache.init_cache()
        rec = line(time_leg(ctx, sampler, cache, False, secs, 1.0))
        rec["what"] = "tables in HBM, cache ratio 0: sample() + cache-free gather of the same rows"
        rec["rows_per_step"] = cache.rows_moved / max(rec["steps"] * rec["repeats"], 1)
        return rec

    def pinned_placement():
        # In a FRESH process: which hardware queue a stream lands on depends on how many streams
        # the process created before it (HIP spreads them over four queues in creation order), and
        # after the legs above the pull stream of this one shares a queue with busy streams —
        # 58 us per step here against 38 in a process of its own, same build, same box.
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--feature-placement", "pinned",
               "--no-config3", "--no-hash-leg", "--no-cpu-baseline", "--no-placement-legs",
               "--min-seconds", str(secs), "--steps", str(ctx.nb), "--warmup", str(args.warmup),
               "--batch-size", str(args.batch_size), "--fanouts", args.fanouts,
               "--strategy", args.strategy, "--cache-ratio", str(args.cache_ratio)]
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
            env.pop(k, None)
        env["GNNFLOW_BENCH_DEVICE"] = str(ctx.local_rank)

        def child(extra_env):
            e = dict(env)
            e.update(extra_env)
            p = subprocess.run(cmd, env=e, stdin=subprocess.DEVNULL, capture_output=True,
                               text=True, timeout=110)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                raise RuntimeError("the pinned-placement process failed: " + p.stderr[-600:])
            return json.loads(lines[-1])
        d = child({})
        rec = {k: d[k] for k in ("value", "unit", "ms_per_step", "target_edges_per_s", "steps",
                                 "repeats", "timed_seconds")}
        rec["pipelined"] = d["config"]["pipelined"]
        rec["process"] = "a fresh process (python bench.py --feature-placement pinned ...)"
        rec["rows_per_step"] = d["config"]["rows_per_step"]
        rec.update(d.get("staging", {}))
        for k in ("roofline", "roofline_lru"):
            if k in d:
                rec[k] = {x: d[k][x] for x in ("kernel", "avg_launch_us", "frac") if x in d[k]}
        rec["cache_edge_ratio"], rec["cache_node_ratio"] = d["cache_edge_ratio"], d["cache_node_ratio"]
        # ... and the same without the ring: every missed row is read from the host table by the
        # gather itself, inside the launch every hit also waits in (round 2-5's pinned path)
        d0 = child({"GNNFLOW_STAGING": "0"})
        rec["without_staging_ring"] = {k: d0[k] for k in ("value", "unit", "ms_per_step",
                                                          "target_edges_per_s")}
        return rec

    for key, fn, limit in (("sample_only", sample_only, 60.0),
                           ("cache_free_device", cache_free_device, 60.0),
                           ("pinned_placement", pinned_placement, 120.0)):
        def guarded(fn=fn):
            try:
                return fn()
            except Exception as e:       # noqa: BLE001 — the headline must survive this leg
                import traceback
                traceback.print_exc()
                return {"error": "{}: {}".format(type(e).__name__, e)}
        out[key] = guarded_aux(ctx, out, key, limit, guarded)


def second_leg(ctx, kind, cache, always_exchange=None):
    import threading
    # RCCL with more than one rank has never run on this code path before the first scaling
    # run: a watchdog ends every rank if a collective hangs — rank 0 first prints the line it
    # has, with an error note for THIS leg, then every rank exits 0: the main figure is complete,
    # and a non-zero status would make the rank's supervisor try the next rung and lose it.
    limit = float(os.environ.get("GNNFLOW_HASH_LEG_TIMEOUT", "180"))
    done = threading.Event()
    key = "replica" if kind != "hash" else (
        "hash_partition_over_rccl_one_rank" if always_exchange and ctx.world == 1
        else "hash_partition")

    def watchdog():
        if not done.wait(limit):
            if ctx.rank == 0 and second_leg.pending_line is not None:
                second_leg.pending_line[key] = {"error": "timed out after {} s".format(limit)}
                main.emit(second_leg.pending_line)
            os._exit(0)
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        if always_exchange and ctx.world == 1:
            import socket
            import torch.distributed as dist
            if not dist.is_initialized():      # one rank: RCCL to this rank itself
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    port = so.getsockname()[1]
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ["MASTER_PORT"] = str(port)
                dist.init_process_group("nccl", device_id=ctx.dev, rank=0, world_size=1)
        _graph, sampler, _ = build_leg(ctx, kind, always_exchange)
        res = time_leg(ctx, sampler, cache, False, min(ctx.args.min_seconds, 1.0), 1.0)
        return {"value": res["edges_all"] / res["elapsed"], "unit": "edges/s",
                "ms_per_step": 1e3 * res["elapsed"] / res["timed_steps"],
                "target_edges_per_s": ctx.args.batch_size * ctx.world * res["timed_steps"] / res["elapsed"],
                "steps": res["steps"], "repeats": res["repeats"],
                "timed_seconds": res["elapsed"], "world_size": ctx.world,
                "pipelined": bool(res["pipe"].pipelined),
                "pipeline_depth": res["pipe"].depth,
                # what RCCL itself says about every lane's communicator (ncclCommCount)
                "rccl_nranks": [c.info()["nranks"] for c in sampler.comms()
                                if c.transport == "rccl"] if hasattr(sampler, "comms") else [],
                "exchange": exchange_note(sampler, ctx.world, ctx.backend) if kind == "hash" else
                "none (a full replica of the graph per GPU, no data-path collective)"}
    except Exception as e:   # the main figure above must survive a failing second leg
        import traceback
        traceback.print_exc()
        return {"error": "{}: {}".format(type(e).__name__, e)}
    finally:
        done.set()


second_leg.pending_line = None


def usable_cores(cap=32):
    """Host threads this process can really run at once: the affinity mask, cut down to the
    container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) — a GPU box exposes all of the
    host's logical CPUs to a container that may only use a few of them."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota)))
    return max(1, min(n, cap))


def cpu_baseline(g, batches, fanouts, args, edge_feats, node_feats, with_gather):
    """The CPU oracle (oracle/gnnflow_oracle.c: C port of the reference's block-walking
    sampler + cache-free gather) on exactly the batches of the GPU's timed region, in whole
    passes — once on one thread and once on all the host cores this process may use (OpenMP
    over the roots / rows; same routine, identical output).  `value` is the faster of the two
    with its `cores`.  Reported baseline, not the target."""
    from oracle import oracle as O
    import numpy as np
    # fresh host copies: the arrays that came back from the GPU sit in pages first touched by
    # the runtime's copy threads (measured 1.5x / 3x slower for the gathers at 1 / 16 threads
    # on the 2-socket GPU box); the CPU side gets its tables in memory it touched itself
    edge_feats, node_feats = np.array(edge_feats, copy=True), np.array(node_feats, copy=True)
    og = O.OracleGraph(minimum_block_size=62, insertion_policy="insert")
    for lo in range(0, g["num_edges"], 100000):
        hi = lo + 100000
        og.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi],
                     add_reverse=args.undirected)
    all_cores = usable_cores()

    # Like for like with the GPU leg (gnnflow_amd/cache/cache.py fetch_feature): under
    # most-recent sampling with equal fanouts and one snapshot, the roots' layer's edge ids
    # are a PREFIX of the outer layer's (same roots, same timestamps, root-major output), so
    # its rows are the first rows of the outer block's gather and are not gathered again.
    alias = args.strategy == "recent" and len(set(fanouts)) == 1

    def run(threads, budget_s, with_gather=with_gather):
        """Whole passes over `batches` (the batch sequence of the GPU's timed region, so
        edges per step are the same on both sides) until the budget is spent; >= 1 pass."""
        osamp = O.OracleSampler(og, fanouts, args.strategy, seed=1234, threads=threads)
        t_total, passes, best = 0.0, 0, None
        while passes == 0 or t_total < budget_s:
            edges, rows, t_pass = 0, 0, 0.0
            for r, t, e in batches:
                t0 = time.perf_counter()
                mfgs = osamp.sample(r, t)
                if with_gather:
                    for blk in mfgs[0]:
                        blk.srcdata["h"] = O.gather_rows(node_feats, blk.srcdata["ID"], threads)
                        rows += len(blk.srcdata["ID"])
                    prev = None      # (edge ids, rows) of the previous edge block
                    for mfg in mfgs:
                        for blk in mfg:
                            n = blk.num_edges()
                            if not n:
                                prev = None
                                continue
                            ids = blk.edata["ID"]
                            if alias and len(mfg) == 1 and prev is not None and n <= len(prev[0]):
                                blk.edata["f"] = prev[1][:n]      # a view: nothing moves
                                prev = (prev[0], prev[1])
                                continue
                            blk.edata["f"] = O.gather_rows(edge_feats, ids, threads)
                            rows += n
                            prev = (ids, blk.edata["f"])
                    O.gather_rows(edge_feats, e, threads)     # target_edge_features
                    rows += len(e)
                t_pass += time.perf_counter() - t0
                edges += sum(b.num_edges() for mfg in mfgs for b in mfg)
            passes += 1
            t_total += t_pass
            # the GPU box is shared with other tenants: the FASTEST pass is the baseline
            if best is None or t_pass < best[1]:
                best = (edges, t_pass, rows)
        return dict(value=best[0] / best[1], cores=threads, batches=len(batches), passes=passes,
                    seconds=t_total, ms_per_step=1e3 * best[1] / max(len(batches), 1),
                    edges_per_step=best[0] / max(len(batches), 1),
                    rows_per_step=best[2] / max(len(batches), 1))

    # the box is shared: more passes for the all-core run, whose time varies most
    one = run(1, args.cpu_seconds / 3)
    runs = [one]
    if all_cores > 1:
        runs.append(run(all_cores, 2 * args.cpu_seconds / 3))
    best = max(runs, key=lambda r: r["value"])
    # the sampler alone (benchmarks/benchmark_sampler.py:71-87 times only `sample`; north_star's
    # ">= 10x the reference CPU sampler" is written on this): same oracle, same batches, no gather
    so = None
    if with_gather:
        so_runs = [run(1, args.cpu_seconds / 6, with_gather=False)]
        if all_cores > 1:
            so_runs.append(run(all_cores, args.cpu_seconds / 3, with_gather=False))
        so_best = max(so_runs, key=lambda r: r["value"])
        so = {"value": so_best["value"], "unit": "edges/s", "cores": so_best["cores"],
              "ms_per_step": so_best["ms_per_step"],
              "target_edges_per_s": args.batch_size / (so_best["ms_per_step"] * 1e-3),
              "single_thread_value": so_runs[0]["value"],
              "sample": "oracle sample() alone on the same batches, fastest whole pass: " +
                        "; ".join("{} thread{}: {:.2f} M edges/s".format(
                            r["cores"], "" if r["cores"] == 1 else "s", r["value"] / 1e6)
                            for r in so_runs)}
    return {
        "sample_only": so,
        "target_edges_per_s": args.batch_size / (best["ms_per_step"] * 1e-3),
        "value": best["value"], "unit": "edges/s", "cores": best["cores"], "kind": "port",
        "sample": "batches 0..{} of the chronological replay — the batch sequence of the GPU's "
                  "timed region — in whole passes, fastest pass reported ({:.1f} s of CPU work in "
                  "all; oracle sample() + cache-free gather of the rows the GPU leg gathers (the "
                  "roots' layer's edge rows alias the outer block's), no LRU bookkeeping; "
                  "gcc -O2 -fopenmp): {}".format(
                      len(batches) - 1, sum(r["seconds"] for r in runs),
                      "; ".join("{} thread{}: best of {} pass{} = {:.2f} M edges/s".format(
                          r["cores"], "" if r["cores"] == 1 else "s", r["passes"],
                          "" if r["passes"] == 1 else "es", r["value"] / 1e6) for r in runs)),
        "ms_per_step": best["ms_per_step"],
        "edges_per_step": best["edges_per_step"],
        "rows_per_step": best["rows_per_step"],
        "single_thread_value": one["value"],
        "host_logical_cpus": os.cpu_count(),
    }


if __name__ == "__main__":
    main()
