"""
Transcribes the literal known-answer expectations of the reference's own unit
tests into data (tests/golden/reference_tests.json).

Only *data* is taken from the reference: the toy graphs fed to DynamicGraph /
TemporalSampler and the arrays the reference's assertions expect back.  Sources:
  /root/reference/tests/test_dynamic_graph.py   (cases "graph/*")
  /root/reference/tests/test_temporal_sampler.py (cases "sampler/*")
Each case records the reference file:line range it was transcribed from.

Run:  python tests/golden/make_reference_tests_json.py
"""
import json
import os

S9 = [0, 0, 0, 1, 1, 1, 2, 2, 2]
D9 = [1, 2, 3, 1, 2, 3, 1, 2, 3]
T9 = [0, 1, 2, 0, 1, 2, 0, 1, 2]
T9b = [3, 4, 5, 3, 4, 5, 3, 4, 5]
S18 = [0] * 6 + [1] * 6 + [2] * 6
D18 = [1, 2, 3, 4, 5, 6] * 3
T18 = [0, 1, 2, 3, 4, 5] * 3


def add(src, dst, ts, eids=None, add_reverse=False):
    return {"op": "add_edges", "src": src, "dst": dst, "ts": ts, "eids": eids,
            "add_reverse": add_reverse}


def nbrs(**kw):
    return {k: {"dst": v[0], "ts": v[1], "eid": v[2]} for k, v in kw.items()}


def check(num_edges, num_vertices, out_degree=None, neighbors=None):
    d = {"op": "check_graph", "num_edges": num_edges, "num_vertices": num_vertices}
    if out_degree is not None:
        d["out_degree"] = {"nodes": [0, 1, 2, 3], "expect": out_degree}
    if neighbors is not None:
        d["neighbors"] = neighbors
    return d


def blk(ID, ts, dt, eid, num_src, num_dst, e0, e1):
    return {"ID": ID, "ts": ts, "dt": dt, "eid": eid, "num_src_nodes": num_src,
            "num_dst_nodes": num_dst, "edges0": e0, "edges1": e1}


EMPTY = [[], [], []]
N321 = lambda e: [[3, 2, 1], [2, 1, 0], e]  # noqa: E731

cases = []

# ------------------------------------------------------------------ graph cases
cases.append({
    "name": "graph/add_edges_sorted_by_timestamp",
    "ref": "tests/test_dynamic_graph.py:24-70",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9),
              check(9, 4, [3, 3, 3, 0], nbrs(n0=N321([2, 1, 0]), n1=N321([5, 4, 3]),
                                             n2=N321([8, 7, 6]), n3=EMPTY))]})
cases.append({
    "name": "graph/add_edges_sorted_add_reverse",
    "ref": "tests/test_dynamic_graph.py:72-118",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9, add_reverse=True),
              check(9, 4, [3, 6, 6, 3], nbrs(
                  n0=N321([2, 1, 0]),
                  n1=[[3, 2, 2, 1, 0, 1], [2, 1, 0, 0, 0, 0], [5, 4, 6, 3, 0, 3]],
                  n2=[[3, 2, 1, 0, 2, 1], [2, 1, 1, 1, 1, 0], [8, 7, 4, 1, 7, 6]],
                  n3=[[2, 1, 0], [2, 2, 2], [8, 5, 2]]))]})
cases.append({
    "name": "graph/add_edges_unsorted",
    "ref": "tests/test_dynamic_graph.py:120-163",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, [2, 1, 0, 2, 1, 0, 2, 1, 0]),
              check(9, 4, [3, 3, 3, 0], nbrs(
                  n0=[[1, 2, 3], [2, 1, 0], [0, 1, 2]],
                  n1=[[1, 2, 3], [2, 1, 0], [3, 4, 5]],
                  n2=[[1, 2, 3], [2, 1, 0], [6, 7, 8]], n3=EMPTY))]})

N6 = lambda e: [[3, 2, 1, 3, 2, 1], [5, 4, 3, 2, 1, 0], e]  # noqa: E731
two_batches_checks = [
    check(9, 4, [3, 3, 3, 0], nbrs(n0=N321([2, 1, 0]), n1=N321([5, 4, 3]),
                                   n2=N321([8, 7, 6]), n3=EMPTY)),
]
after_second = check(18, 4, [6, 6, 6, 0], nbrs(
    n0=N6([11, 10, 9, 2, 1, 0]), n1=N6([14, 13, 12, 5, 4, 3]),
    n2=N6([17, 16, 15, 8, 7, 6]), n3=EMPTY))
for pol, ref in (("insert", "tests/test_dynamic_graph.py:165-242"),
                 ("replace", "tests/test_dynamic_graph.py:244-322")):
    cases.append({
        "name": "graph/add_edges_multiple_times_" + pol, "ref": ref,
        "graph": {"minimum_block_size": 4, "insertion_policy": pol},
        "steps": [add(S9, D9, T9)] + two_batches_checks + [add(S9, D9, T9b), after_second]})
cases.append({
    "name": "graph/insertion_policy_replace",
    "ref": "tests/test_dynamic_graph.py:346-403",
    "graph": {"minimum_block_size": 4, "insertion_policy": "replace"},
    "steps": [add(S9, D9, T9), add(S9, D9, T9b), after_second]})
cases.append({
    "name": "graph/add_edges_with_eids",
    "ref": "tests/test_dynamic_graph.py:405-455",
    "graph": {"minimum_block_size": 4, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9, eids=list(range(9))),
              add(S9, D9, T9b, eids=list(range(9, 18))), after_second]})
E1 = [0, 2, 4, 6, 8, 10, 12, 14, 16]
E2 = [17, 19, 21, 23, 25, 27, 29, 31, 33]
cases.append({
    "name": "graph/add_edges_with_noncontinuous_eids",
    "ref": "tests/test_dynamic_graph.py:457-515",
    "graph": {"minimum_block_size": 4, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9, eids=E1), add(S9, D9, T9b, eids=E2),
              check(18, 4, [6, 6, 6, 0], nbrs(
                  n0=N6([21, 19, 17, 4, 2, 0]), n1=N6([27, 25, 23, 10, 8, 6]),
                  n2=N6([33, 31, 29, 16, 14, 12]), n3=EMPTY))]})
cases.append({
    "name": "graph/offload_old_blocks",
    "ref": "tests/test_dynamic_graph.py:517-573",
    "graph": {"minimum_block_size": 4, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9, eids=E1), add(S9, D9, T9b, eids=E2),
              {"op": "offload", "ts": 3.5},
              check(6, 4, None, nbrs(
                  n0=[[3, 2], [5, 4], [21, 19]], n1=[[3, 2], [5, 4], [27, 25]],
                  n2=[[3, 2], [5, 4], [33, 31]], n3=EMPTY))]})

# ---------------------------------------------------------------- sampler cases
L1 = blk([0, 1, 2, 2, 1, 2, 1, 2, 1], [1.5, 1.5, 1.5, 1, 0, 1, 0, 1, 0],
         [0.5, 1.5, 0.5, 1.5, 0.5, 1.5], [1, 0, 4, 3, 7, 6], 9, 3,
         [3, 4, 5, 6, 7, 8], [0, 0, 1, 1, 2, 2])
R3 = [0, 1, 2]
T15 = [1.5, 1.5, 1.5]


def sample(sampler, nodes, ts, expect):
    return {"op": "sample", "sampler": sampler, "nodes": nodes, "ts": ts, "expect": expect}


def sample_layer(sampler, nodes, ts, layer, snapshot, expect):
    return {"op": "sample_layer", "sampler": sampler, "nodes": nodes, "ts": ts,
            "layer": layer, "snapshot": snapshot, "expect": expect}


cases.append({
    "name": "sampler/sample_layer", "ref": "tests/test_temporal_sampler.py:27-78",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9),
              sample({"fanouts": [2]}, R3, T15, {"0,0": L1}),
              sample_layer({"fanouts": [2]}, R3, T15, 0, 0, L1)]})
cases.append({
    "name": "sampler/sample_layer_uniform_shapes",
    "ref": "tests/test_temporal_sampler.py:80-110",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9),
              sample({"fanouts": [2], "sample_strategy": "uniform"}, R3, [3, 3, 3],
                     {"0,0": {"num_src_nodes": 9, "num_dst_nodes": 3}}),
              sample_layer({"fanouts": [2], "sample_strategy": "uniform"}, R3, [3, 3, 3],
                           0, 0, {"num_src_nodes": 9, "num_dst_nodes": 3})]})
cases.append({
    "name": "sampler/sample_layer_with_multiple_blocks",
    "ref": "tests/test_temporal_sampler.py:112-172",
    "graph": {"minimum_block_size": 4, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9), add(S9, D9, T9b),
              sample({"fanouts": [2]}, R3, T15, {"0,0": L1}),
              sample_layer({"fanouts": [2]}, R3, T15, 0, 0, L1)]})
cases.append({
    "name": "sampler/sample_layer_with_multiple_blocks_offload",
    "ref": "tests/test_temporal_sampler.py:174-238",
    "graph": {"minimum_block_size": 4, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9), add(S9, D9, T9b), {"op": "offload", "ts": 3.5},
              sample({"fanouts": [2]}, R3, T15,
                     {"0,0": blk([0, 1, 2], T15, [], [], 3, 3, [], [])}),
              sample({"fanouts": [2]}, R3, [4.5, 4.5, 4.5],
                     {"0,0": blk([0, 1, 2, 2, 2, 2], [4.5, 4.5, 4.5, 4, 4, 4],
                                 [0.5, 0.5, 0.5], [10, 13, 16], 6, 3,
                                 [3, 4, 5], [0, 1, 2])})]})
DUP = blk([0, 1, 2, 0, 2, 1, 2, 1, 2, 1, 2, 1],
          [1.5, 1.5, 1.5, 1.5, 1, 0, 1, 0, 1, 0, 1, 0],
          [0.5, 1.5, 0.5, 1.5, 0.5, 1.5, 0.5, 1.5], [1, 0, 4, 3, 7, 6, 1, 0], 12, 4,
          [4, 5, 6, 7, 8, 9, 10, 11], [0, 0, 1, 1, 2, 2, 3, 3])
cases.append({
    "name": "sampler/sample_layer_with_duplicate_vertices",
    "ref": "tests/test_temporal_sampler.py:240-293",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9),
              sample({"fanouts": [2]}, [0, 1, 2, 0], [1.5] * 4, {"0,0": DUP}),
              sample_layer({"fanouts": [2]}, [0, 1, 2, 0], [1.5] * 4, 0, 0, DUP)]})
L2 = blk([0, 1, 2, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 1, 1, 1],
         [1.5, 1.5, 1.5, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0],
         [0.5, 1.5, 0.5, 1.5, 0.5, 1.5, 1, 1, 1], [1, 0, 4, 3, 7, 6, 6, 6, 6], 18, 9,
         [9, 10, 11, 12, 13, 14, 15, 16, 17], [0, 0, 1, 1, 2, 2, 3, 5, 7])
cases.append({
    "name": "sampler/sample_multi_layers",
    "ref": "tests/test_temporal_sampler.py:295-386",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9),
              # blocks[1][0] = roots' layer, blocks[0][0] = last-sampled layer
              sample({"fanouts": [2, 2]}, R3, T15, {"1,0": L1, "0,0": L2}),
              sample_layer({"fanouts": [2, 2]}, R3, T15, 0, 0, L1),
              sample_layer({"fanouts": [2, 2]}, L1["ID"], L1["ts"], 1, 0, L2)]})
SNAP1 = blk([0, 1, 2, 5, 5, 5], [5, 5, 5, 4, 4, 4], [1, 1, 1], [4, 10, 16], 6, 3,
            [3, 4, 5], [0, 1, 2])  # window [4, 5)
SNAP0 = blk([0, 1, 2, 4, 4, 4], [5, 5, 5, 3, 3, 3], [2, 2, 2], [3, 9, 15], 6, 3,
            [3, 4, 5], [0, 1, 2])  # window [3, 4)
MS = {"fanouts": [2], "num_snapshots": 2, "snapshot_time_window": 1}
cases.append({
    "name": "sampler/sample_multi_snapshots",
    "ref": "tests/test_temporal_sampler.py:388-489",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S18, D18, T18),
              sample(MS, R3, [5, 5, 5], {"0,1": SNAP1, "0,0": SNAP0}),
              sample_layer(MS, R3, [5, 5, 5], 0, 1, SNAP1),
              sample_layer(MS, R3, [5, 5, 5], 0, 0, SNAP0)]})
SNAP1_L2 = blk([0, 1, 2, 5, 5, 5, 5, 5, 5], [5, 5, 5, 4, 4, 4, 4, 4, 4], [1, 1, 1],
               [4, 10, 16], 9, 6, [6, 7, 8], [0, 1, 2])
SNAP0_L2 = blk([0, 1, 2, 4, 4, 4, 4, 4, 4], [5, 5, 5, 3, 3, 3, 3, 3, 3], [2, 2, 2],
               [3, 9, 15], 9, 6, [6, 7, 8], [0, 1, 2])
MS2 = {"fanouts": [2, 2], "num_snapshots": 2, "snapshot_time_window": 1}
cases.append({
    "name": "sampler/sample_multi_layers_multi_snapshots",
    "ref": "tests/test_temporal_sampler.py:491-656",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S18, D18, T18),
              sample(MS2, R3, [5, 5, 5], {"1,1": SNAP1, "1,0": SNAP0,
                                          "0,1": SNAP1_L2, "0,0": SNAP0_L2}),
              sample_layer(MS2, R3, [5, 5, 5], 0, 1, SNAP1),
              sample_layer(MS2, SNAP1["ID"], SNAP1["ts"], 1, 1, SNAP1_L2),
              sample_layer(MS2, R3, [5, 5, 5], 0, 0, SNAP0),
              sample_layer(MS2, SNAP0["ID"], SNAP0["ts"], 1, 0, SNAP0_L2)]})
cases.append({
    "name": "sampler/sample_layer_with_different_batch_size",
    "ref": "tests/test_temporal_sampler.py:658-682",
    "graph": {"minimum_block_size": 64, "insertion_policy": "insert"},
    "steps": [add(S9, D9, T9),
              {"op": "sample_random_batches", "sampler": {"fanouts": [2]},
               "batch_sizes": list(range(0, 100, 10)), "node_high": 3, "ts_high": 3}]})

out = {
    "source": "jasperzhong/GNNFlow tests/test_dynamic_graph.py + tests/test_temporal_sampler.py "
              "(literal assertion values; transcribed by tests/golden/make_reference_tests_json.py)",
    "block_key": "expect keys are '<layer index in the returned (reversed) list>,<snapshot>'",
    "cases": cases,
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tests.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(cases), "cases")
