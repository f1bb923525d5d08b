"""bench.py's output contract on a real GPU: one JSON line from rank 0 with the fields the
driver reads, the roofline / cpu_baseline / config3 objects, a 2-rank launch through torchrun
and the same through bench.py's own launcher (`python bench.py --gpus 2`, the shape of the
driver's command; both ranks on the one GPU of the test box, gloo carries the bookkeeping
collectives), and the fall-back to the replica loop when the hash-partitioned one fails."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int,
            "warmup": int, "ms_per_step": float, "higher_is_better": bool, "scaling": str,
            "dtype": str, "data": str, "config": dict}


def _run(cmd, env=None, rc=0):
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):   # a launcher's leftovers
        e.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == rc, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]       # exactly ONE JSON line
    return json.loads(lines[0])


def _check_common(d, n_gpus, steps, warmup):
    for k, t in REQUIRED.items():
        assert k in d and isinstance(d[k], t), k
    assert d["metric"] == "sampled_edges_per_s" and d["unit"] == "edges/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (n_gpus, steps, warmup)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_single_gpu_line_with_roofline_and_cpu_baseline():
    d = _run([sys.executable, "bench.py", "--steps", "60", "--warmup", "5", "--cpu-seconds", "1",
              "--min-seconds", "0.1", "--config3-nodes", "200000", "--config3-edges", "4000000",
              "--config3-batches", "600,6000"])
    _check_common(d, 1, 60, 5)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert 0 < r["frac"] < 1 and r["launches_timed"] > 0 and r["avg_launch_us"] > 0
    # PMC counters are not readable in-process: `traffic` is null, the committed profile of
    # the driver's command line rides along under another name, only for that command line
    assert r["traffic"] is None and "traffic_from_profile" not in r
    # the timed region covers whole chronological replays whatever --steps is ...
    assert d["repeats"] >= 75 and d["timed_steps"] == 60 * d["repeats"]
    assert d["config"]["timed_steps"] == d["timed_steps"]
    assert d["ms_per_step"] == pytest.approx(1e3 * d["timed_seconds"] / d["timed_steps"])
    # issuing time of the library's enqueue threads per step: GPU-bound or host-bound?
    assert 0 < d["config"]["enqueue_threads_busy_us_per_step"] < 2e3 * d["ms_per_step"]
    assert d["timed_seconds"] >= 0.1 and d["config"]["timed_seconds"] == d["timed_seconds"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "edges/s" and c["value"] > 0
    assert c["single_thread_value"] > 0
    assert isinstance(c["sample"], str) and c["sample"]
    assert d["value"] > c["value"]
    # ... and the CPU baseline ran the same batches: same sampled edges per step
    assert c["edges_per_step"] == pytest.approx(d["config"]["edges_per_step"], rel=0.02)
    # ... and moved the same feature rows (both legs serve the roots' layer's edge block from
    # the outer block's rows)
    assert c["rows_per_step"] == pytest.approx(d["config"]["rows_per_step"], rel=0.02)
    assert c["host_logical_cpus"] >= c["cores"]
    # north_star's hash-partitioned split rides in the same line (one rank: no exchange)
    h = d["hash_partition"]
    assert "error" not in h and h["world_size"] == 1 and h["value"] > 0 and h["steps"] == 1121
    assert h["value"] > 0.5 * d["value"]         # the hash path is not a second-class citizen
    # ... and so does the chain an N > 1 run issues, every message through RCCL to this rank
    hr = d["hash_partition_over_rccl_one_rank"]
    assert "error" not in hr and hr["value"] > 0.4 * d["value"]
    assert "RCCL communicator" in hr["exchange"] and "samples each on average" in hr["exchange"]
    assert hr["rccl_nranks"] and set(hr["rccl_nranks"]) == {1}    # ncclCommCount of every lane
    # the harness's own unit and the sampler-only comparison north_star's ">= 10x the reference
    # CPU sampler" is written on (benchmarks/benchmark_sampler.py:71-87), on both legs
    assert d["target_edges_per_s"] == pytest.approx(600 / (d["ms_per_step"] * 1e-3))
    assert h["target_edges_per_s"] > 0 and hr["target_edges_per_s"] > 0
    so, cso = d["sample_only"], c["sample_only"]
    assert "error" not in so and so["pipelined"] and so["value"] > d["value"]
    assert so["target_edges_per_s"] == pytest.approx(600 / (so["ms_per_step"] * 1e-3))
    assert cso["value"] >= c["value"] and cso["single_thread_value"] > 0 and cso["cores"] >= 1
    assert c["target_edges_per_s"] > 0 and cso["target_edges_per_s"] > 0
    assert so["value"] > 10 * cso["value"]                 # north_star's target, sampler on sampler
    # the other feature placements beside the headline: the gather without a cache (tables in
    # HBM), and the reference's placement — tables in pinned host memory behind LRU 0.2
    cf, pp = d["cache_free_device"], d["pinned_placement"]
    assert "error" not in cf and cf["value"] > 0
    assert cf["rows_per_step"] == pytest.approx(d["config"]["rows_per_step"], rel=0.02)
    assert "error" not in pp and pp["value"] > 0
    assert pp["rows_per_step"] == pytest.approx(d["config"]["rows_per_step"], rel=0.02)
    assert 0 < pp["cache_edge_ratio"] <= 1 and 0 < pp["cache_node_ratio"] <= 1
    assert pp["rows_pulled_into_ring_per_step"] > 0
    assert pp["host_link_bytes_per_step"] == pytest.approx(
        688 * (pp["rows_pulled_into_ring_per_step"] + pp["rows_read_from_host_by_gather_per_step"]))
    assert pp["host_link_GBps_memcpy_256MB"] > 1
    assert pp["value"] > pp["without_staging_ring"]["value"]    # the ring pays
    # BASELINE configs[2] (here on a small graph) rides in the same line
    c3 = d["config3"]
    assert "error" not in c3 and [r["batch"] for r in c3["rows"]] == [600, 6000]
    for row in c3["rows"]:
        assert row["policy"] == "uniform" and row["edges"] > 0 and row["edges_per_s"] > 0
        assert 0 < row["search_frac"] < 1 and 0 < row["emit_frac"] < 1


def test_driver_command_line_is_representative():
    """`--steps 20 --warmup 5` (what the driver runs): full-replay mean edges per step, and a
    timed region an outside clock can see (>= 2 s)."""
    d = _run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
              "--no-config3"])
    _check_common(d, 1, 20, 5)
    assert d["repeats"] >= 225 and d["timed_seconds"] >= 1.9
    assert d["roofline"].get("traffic_source", "").startswith("profiles/") or \
        "traffic_from_profile" not in d["roofline"]
    full = _run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-config3",
                 "--min-seconds", "0.05"])      # (four replays take ~0.1 s)
    assert full["repeats"] == 4
    assert d["config"]["edges_per_step"] == pytest.approx(full["config"]["edges_per_step"],
                                                         rel=0.02)


def test_two_ranks_through_torchrun():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2",
              "--steps", "40", "--warmup", "5", "--min-replays", "1"],
             env={"GNNFLOW_BENCH_DEVICE": "0", "GNNFLOW_BENCH_BACKEND": "gloo"})
    _check_common(d, 2, 40, 5)
    # N > 1: the headline is north_star's split — the graph hash-partitioned over the ranks,
    # both ranks exchanging roots / replies per layer (staged through gloo on this one-GPU box)
    assert d["config"]["parallelism"] == "hash-dp2"
    assert "equal-split" in d["config"]["exchange"] and "no host sync" in d["config"]["exchange"]
    assert "cpu_baseline" not in d       # rank 0 at N = 1 only
    r = d["replica"]                     # the per-GPU-replica figure rides along
    assert "error" not in r and r["world_size"] == 2 and r["value"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 --steps 20 --warmup 5` — no torchrun, no WORLD_SIZE: the parent
    starts the ranks itself (before touching HIP) and relays rank 0's one line."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5",
              "--min-replays", "1", "--min-seconds", "0.2"],
             env={"GNNFLOW_BENCH_DEVICE": "0", "GNNFLOW_BENCH_BACKEND": "gloo"})
    _check_common(d, 2, 20, 5)
    assert d["config"]["parallelism"] == "hash-dp2"
    assert "error" not in d["replica"] and d["replica"]["world_size"] == 2
    # the N > 1 line describes itself: which rung of the ladder produced it, the ranks' devices,
    # what travelled, and the figure against replicas on the same GPUs
    assert d["ladder"]["rung"] == 0 and d["ladder"]["arrangement"] == "hash" and \
        d["ladder"]["tried_before"] == []
    m = d["multi_gpu"]
    assert [r["rank"] for r in m["ranks"]] == [0, 1] and all(r["pci_bus_id"] for r in m["ranks"])
    assert m["distinct_devices"] == 1           # both ranks share the test box's one GPU
    assert m["overflowed_samples"] == 0 and m["rccl_nranks"] is None   # staged through gloo here
    assert m["request_bytes_per_step_and_rank"] > 0 and \
        m["reply_bytes_per_step_and_rank"] > m["request_bytes_per_step_and_rank"]
    assert d["efficiency_vs_replica_same_run"] == pytest.approx(d["value"] / d["replica"]["value"])


def test_failing_hash_loop_falls_back_to_the_replica_figure():
    """The hash-partitioned loop raising on every rank (on all three hash rungs of the ladder) must
    not cost the line: fresh ranks time the replica loop and the record says so."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5",
              "--min-replays", "1", "--min-seconds", "0.2"],
             env={"GNNFLOW_BENCH_DEVICE": "0", "GNNFLOW_BENCH_BACKEND": "gloo",
                  "GNNFLOW_BENCH_FAIL_HASH": "1"})
    _check_common(d, 2, 20, 5)
    assert d["config"]["parallelism"] == "replica-dp2"
    assert "GNNFLOW_BENCH_FAIL_HASH" in d["hash_partition"]["error"]
    lad = d["ladder"]
    assert lad["rung"] == 3 and lad["arrangement"] == "replica"
    assert [h["arrangement"] for h in lad["tried_before"]] == ["hash", "hash-one-lane", "hash-simple"]
    assert all(h["hung"] is False for h in lad["tried_before"])


@pytest.mark.parametrize("who,rungs,final", [("1", "0", "hash-one-lane"), ("1", "0,1", "hash-simple"),
                                             ("all", "0,1,2", "replica")])
def test_hanging_hash_loop_walks_down_the_ladder(who, rungs, final):
    """A collective that some rank never joins does not raise, it HANGS: after
    GNNFLOW_HASH_MAIN_TIMEOUT every rank's worker process gives up and exits, and the rank's
    supervisor (which never touches the GPU) starts a fresh worker for the NEXT rung on a new
    rendezvous port — the hash loop on ONE lane (one communicator) with shared chains, then in
    its simplest arrangement (1 lane, single chains), then the replica loop — and the ONE line
    says what happened on the way (own launcher and torchrun alike)."""
    env = {"GNNFLOW_BENCH_DEVICE": "0", "GNNFLOW_BENCH_BACKEND": "gloo",
           "GNNFLOW_BENCH_HANG_HASH": who, "GNNFLOW_BENCH_HANG_RUNGS": rungs,
           # the rungs that hang give up after 15 s; a rung that runs gets the time a slow,
           # shared box may need (the whole suite has taken 1.8 x its usual time on one)
           "GNNFLOW_HASH_MAIN_TIMEOUT": {"0": "15,150", "0,1": "15,15,150"}.get(rungs, "15"),
           "GNNFLOW_PART_TRANSPORT": "ipc"}
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--min-replays", "1", "--min-seconds", "0.2"]
    if who == "all":        # the driver's torchrun form: the agent's store must not be reused
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29577"] + cmd[1:]
    d = _run(cmd, env=env)
    _check_common(d, 2, 20, 5)
    lad = d["ladder"]
    assert lad["arrangement"] == final
    tried = lad["tried_before"]
    assert [h["arrangement"] for h in tried] == ["hash", "hash-one-lane", "hash-simple"][:len(tried)]
    assert all(h["hung"] and "did not finish within 15 s" in h["error"] for h in tried)
    # every rung's ranks were gone (here: by themselves, after their watchdog fired) before the
    # next rung's started
    assert all(h["worker_killed"] is False and 15 <= h["seconds"] < 240 for h in tried)
    if final == "replica":
        assert d["config"]["parallelism"] == "replica-dp2" and len(tried) == 3
        assert "did not finish within 15 s" in d["hash_partition"]["error"]
    else:
        # the native chains over the library's communicator on ONE lane: shared chains of four
        # (rung 1), single chains with slot capacity 2.0 (rung 2)
        simple = final == "hash-simple"
        assert d["config"]["parallelism"] == "hash-dp2" and len(tried) == (2 if simple else 1)
        m = d["multi_gpu"]
        assert m["lanes"] == 1 and m["chain_samples"] == (1 if simple else 4) and m["transport"] == "ipc"
        assert m["communicator_nranks"] == [2]
        if simple:
            assert m["wire"]["slack"] == 2.0
        a = m["all_to_all"]["reply_layer1"]
        assert a["device_us"] > 0 and a["issue_us"] > 0 and a["bytes_per_peer"] > 0
        assert m["projection"]["chains_sustain_us_per_step"] > 0


def test_a_dying_rank_ends_the_launcher_non_zero_with_one_line():
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"],
             env={"GNNFLOW_BENCH_DEVICE": "7", "GNNFLOW_BENCH_BACKEND": "gloo"}, rc=1)
    assert d["value"] == 0.0 and "error" in d


def test_four_ranks_sharing_the_gpu_finish_the_hash_loop():
    """Four rank processes time-slicing ONE GPU, native chains over the hipIpc transport: the
    fused merge's look-back must not depend on the other tiles being dispatched (before the
    recount fall-back this run ended in "a look-back granule did not arrive" and the bench fell
    back to the replica loop)."""
    d = _run([sys.executable, "bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5",
              "--min-replays", "1", "--min-seconds", "0.2", "--no-second-leg"],
             env={"GNNFLOW_BENCH_DEVICE": "0", "GNNFLOW_BENCH_BACKEND": "gloo",
                  "GNNFLOW_PART_TRANSPORT": "ipc"})
    _check_common(d, 4, 20, 5)
    assert d["config"]["parallelism"] == "hash-dp4" and "hash_partition" not in d


def test_two_ranks_native_chains_over_the_ipc_transport():
    """The closest a one-GPU box gets to the driver's multi-GPU run: bench.py's own launcher, two
    rank processes, and the NATIVE partitioned chains (sampling lanes, up to four samples per chain,
    issued by the enqueue thread) with every message real — carried by the library's hipIpc
    transport instead of RCCL, which refuses ranks that share a GPU."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5",
              "--min-replays", "1", "--min-seconds", "0.2", "--no-second-leg"],
             env={"GNNFLOW_BENCH_DEVICE": "0", "GNNFLOW_BENCH_BACKEND": "gloo",
                  "GNNFLOW_PART_TRANSPORT": "ipc"})
    _check_common(d, 2, 20, 5)
    assert d["config"]["parallelism"] == "hash-dp2"
    assert "hipIpc" in d["config"]["exchange"] and \
        "samples each on average" in d["config"]["exchange"]
    assert "0 overflowed" in d["config"]["exchange"]
