"""bench.py's own launcher (`python bench.py --gpus N` without torchrun) on a box without a GPU:
the children fail their GPU assertion, the parent — which must not import torch or touch HIP —
still prints exactly one JSON line and exits non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_reports_failing_ranks_with_one_line():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = ""          # also on a GPU box: no rank finds a device
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    # every rank is a supervisor that walks the ladder (hash, hash-one-lane, hash-simple, replica) with a fresh
    # worker per rung; here every worker dies at once, and rank 0's supervisor says so
    assert d["value"] == 0.0 and d["n_gpus"] == 2 and "no rung printed a record" in d["error"]
    assert [h["arrangement"] for h in d["ladder"]] == ["hash", "hash-one-lane", "hash-simple"]
    assert all("exited with status" in h["error"] and not h["worker_killed"] for h in d["ladder"])
    assert "bench.py needs a GPU" in p.stderr          # the workers' own message got through


def test_launcher_parent_does_not_load_torch():
    """The parent's work happens in launch_ranks(): importing bench and calling parse() must
    not pull torch in (a process that initialised the GPU must not start the ranks)."""
    code = ("import sys; sys.argv=['bench.py','--gpus','2']; import bench; "
            "a = bench.parse(sys.argv[1:]); assert a.gpus == 2; "
            "assert 'torch' not in sys.modules and 'gnnflow_amd' not in sys.modules")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True,
                       timeout=60)
    assert p.returncode == 0, p.stderr


def _acted(outcomes, extra_env=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env["GNNFLOW_BENCH_FAKE_WORKER"] = outcomes
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])      # exactly ONE line, whatever happens
    return p.returncode, json.loads(lines[0]), p.stderr


def test_ladder_first_rung_succeeds():
    rc, d, _ = _acted("ok,ok,ok,ok")
    assert rc == 0 and d["ladder"]["rung"] == 0 and d["ladder"]["tried_before"] == []


def test_ladder_walks_down_when_rungs_give_up():
    """Launcher -> two supervisors (never a GPU call) -> a worker per rung (here: acting its
    outcome): the three hash rungs give up on both ranks one after the other, the replica rung
    prints the line; every rung rendezvous on its own port.  A rung in the middle that works ends
    the walk there."""
    rc, d, err = _acted("giveup,giveup,giveup,ok")
    assert rc == 0
    lad = d["ladder"]
    assert lad["rung"] == 3 and lad["arrangement"] == "replica"
    assert [h["arrangement"] for h in lad["tried_before"]] == ["hash", "hash-one-lane", "hash-simple"]
    assert all(h["hung"] and not h["worker_killed"] for h in lad["tried_before"])
    assert "starting rung 1 (hash-one-lane)" in err and "starting rung 2 (hash-simple)" in err
    assert "starting rung 3 (replica)" in err
    rc, d, _ = _acted("giveup,giveup,ok,ok")
    assert rc == 0 and d["ladder"]["rung"] == 2 and d["ladder"]["arrangement"] == "hash-simple"


def test_ladder_kills_a_worker_that_neither_finishes_nor_gives_up():
    rc, d, _ = _acted("hang,ok,ok,ok", {"GNNFLOW_HASH_MAIN_TIMEOUT": "2",
                                     "GNNFLOW_RUNG_SETUP_ALLOWANCE": "1"})
    assert rc == 0 and d["ladder"]["rung"] == 1
    (h,) = d["ladder"]["tried_before"]
    assert h["worker_killed"] is True and h["hung"] is True and "killed after" in h["error"]
    assert 3 <= h["seconds"] < 30


def test_ladder_ranks_leave_a_rung_together_when_only_one_worker_dies():
    """Rank 1's rung-0 worker dies at once, rank 0's hangs until the supervisor kills it (3 s):
    rank 1 must WAIT for rank 0 to leave rung 0 instead of racing ahead to rung 1 on the next
    port — where its worker would wait alone, be killed before rank 0 arrives, and leave the ranks
    on different rungs for good.  Both then meet in rung 1, which works."""
    rc, d, err = _acted("hang,ok,ok,ok", {"GNNFLOW_BENCH_FAKE_WORKER_RANK1": "die,ok,ok,ok",
                                          "GNNFLOW_HASH_MAIN_TIMEOUT": "2",
                                          "GNNFLOW_RUNG_SETUP_ALLOWANCE": "1"})
    assert rc == 0 and d["ladder"]["rung"] == 1
    (h,) = d["ladder"]["tried_before"]            # rank 0's view: its worker was killed ...
    assert h["worker_killed"] is True and h["ranks_left_together"] == 2
    # ... and rank 1, whose worker was gone within a second, started rung 1 only after ~3 s
    import re
    m = re.search(r"rank 1: rung 0 \(hash\) is over after (\d+) s", err)
    assert m and int(m.group(1)) >= 2, err[-1500:]
    # the other way round (rank 0 dies, rank 1 hangs): the line still comes from rung 1
    rc, d, _ = _acted("die,ok,ok,ok", {"GNNFLOW_BENCH_FAKE_WORKER_RANK1": "hang,ok,ok,ok",
                                       "GNNFLOW_HASH_MAIN_TIMEOUT": "2",
                                       "GNNFLOW_RUNG_SETUP_ALLOWANCE": "1"})
    assert rc == 0 and d["ladder"]["rung"] == 1
    assert d["ladder"]["tried_before"][0]["ranks_left_together"] == 2


def test_ladder_ports_differ_per_rung_and_a_dying_last_rung_fails_the_run():
    rc0, d0, _ = _acted("ok,ok,ok,ok", {"MASTER_PORT": "29900"})
    rc1, d1, _ = _acted("die,ok,ok,ok", {"MASTER_PORT": "29900"})
    assert d0["master_port"] != d1["master_port"]            # rung 0 and rung 1 rendezvous apart
    rc, d, _ = _acted("die,die,die,die")
    assert rc != 0 and d["value"] == 0.0 and "no rung printed a record" in d["error"]
    assert [h["arrangement"] for h in d["ladder"]] == ["hash", "hash-one-lane", "hash-simple"]


def test_rung_time_limits_parse_one_value_or_one_per_rung(monkeypatch):
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.delenv("GNNFLOW_HASH_MAIN_TIMEOUT", raising=False)
    assert [bench.rung_seconds("GNNFLOW_HASH_MAIN_TIMEOUT", r, bench.RUNG_MAIN_SECONDS)
            for r in range(3)] == [float(x) for x in bench.RUNG_MAIN_SECONDS]
    monkeypatch.setenv("GNNFLOW_HASH_MAIN_TIMEOUT", "15")
    assert [bench.rung_seconds("GNNFLOW_HASH_MAIN_TIMEOUT", r, (1, 2, 3)) for r in range(3)] == [15.0] * 3
    monkeypatch.setenv("GNNFLOW_HASH_MAIN_TIMEOUT", "15,150")
    assert [bench.rung_seconds("GNNFLOW_HASH_MAIN_TIMEOUT", r, (1, 2, 3)) for r in range(3)] == [15.0, 150.0, 150.0]
    # the whole ladder fits the driver's 600 s with a minute left for the replica rung
    assert sum(bench.RUNG_MAIN_SECONDS) + sum(bench.RUNG_SETUP_SECONDS) <= 540
    assert len(bench.RUNGS) == len(bench.RUNG_MAIN_SECONDS) + 1
