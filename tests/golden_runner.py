"""
Runs the transcribed reference test cases (tests/golden/reference_tests.json)
against any implementation of the reference's Python API:

    make_graph(**graph_kwargs)            -> object with gnnflow.DynamicGraph methods
    make_sampler(graph, **sampler_kwargs) -> object with gnnflow.TemporalSampler methods

Used with the CPU oracle (tests/test_oracle_golden.py) and with the HIP product
path (tests/test_gpu_golden.py), so both are pinned to the same golden vectors.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                      "reference_tests.json")


def load_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


def _tolist(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x).tolist()


def check_block(block, exp, where):
    if "ID" in exp:
        assert _tolist(block.srcdata["ID"]) == exp["ID"], where
        assert _tolist(block.srcdata["ts"]) == exp["ts"], where
        assert _tolist(block.edata["dt"]) == exp["dt"], where
        assert _tolist(block.edata["ID"]) == exp["eid"], where
        e0, e1 = block.edges()
        assert _tolist(e0) == exp["edges0"], where
        assert _tolist(e1) == exp["edges1"], where
    assert block.num_src_nodes() == exp["num_src_nodes"], where
    assert block.num_dst_nodes() == exp["num_dst_nodes"], where


def run_case(case, make_graph, make_sampler, default_graph_kwargs=None):
    kw = dict(default_graph_kwargs or {})
    kw.update(case["graph"])
    g = make_graph(**kw)
    samplers = {}

    def sampler_for(cfg):
        key = json.dumps(cfg, sort_keys=True)
        if key not in samplers:
            samplers[key] = make_sampler(g, **cfg)
        return samplers[key]

    for si, step in enumerate(case["steps"]):
        where = "{} step {} ({})".format(case["name"], si, case["ref"])
        op = step["op"]
        if op == "add_edges":
            eids = None if step["eids"] is None else np.array(step["eids"])
            g.add_edges(np.array(step["src"]), np.array(step["dst"]),
                        np.array(step["ts"]), eids, add_reverse=step["add_reverse"])
        elif op == "offload":
            g.offload_old_blocks(step["ts"])
        elif op == "check_graph":
            assert g.num_edges() == step["num_edges"], where
            assert g.num_vertices() == step["num_vertices"], where
            if "out_degree" in step:
                od = g.out_degree(np.array(step["out_degree"]["nodes"]))
                assert _tolist(od) == step["out_degree"]["expect"], where
            for k, v in step.get("neighbors", {}).items():
                d, t, e = g.get_temporal_neighbors(int(k[1:]))
                assert _tolist(d) == v["dst"], where + " node " + k
                assert _tolist(t) == v["ts"], where + " node " + k
                assert _tolist(e) == v["eid"], where + " node " + k
        elif op == "sample":
            s = sampler_for(step["sampler"])
            mfgs = s.sample(np.array(step["nodes"]), np.array(step["ts"]))
            for key, exp in step["expect"].items():
                li, sn = (int(x) for x in key.split(","))
                check_block(mfgs[li][sn], exp, where + " block " + key)
        elif op == "sample_layer":
            s = sampler_for(step["sampler"])
            b = s.sample_layer(np.array(step["nodes"]), np.array(step["ts"]),
                               step["layer"], step["snapshot"])
            check_block(b, step["expect"], where)
        elif op == "sample_random_batches":
            s = sampler_for(step["sampler"])
            rng = np.random.RandomState(0)
            for bs in step["batch_sizes"]:
                nodes = rng.randint(0, step["node_high"], bs)
                ts = rng.randint(0, step["ts_high"], bs)
                mfgs = s.sample(nodes, ts)
                assert mfgs[0][0].num_dst_nodes() == bs, where
                b = s.sample_layer(nodes, ts, 0, 0)
                assert b.num_dst_nodes() == bs, where
        else:
            raise AssertionError("unknown op " + op)
    return g
