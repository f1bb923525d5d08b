"""A second, independent pin for the oracle: the block-walking C restatement
(oracle/gnnflow_oracle.c, which follows the reference's linked blocks and per-slot
LowerBound walk) against a direct numpy statement of what the reference's sampler *means*
(SURVEY.md 8(a) rows 3, 8, 9, 11 and the equivalence note under 8(c)):

  * a node's edges form one chronological sequence: batches in arrival order, inside a batch
    the node's edges stable-sorted by timestamp (dynamic_graph.cu:105-128);
  * candidates of a root = the edges with start <= ts < end of its window
    (sampling_kernels.cu:28-40); slot j takes the j-th most recent candidate;
  * results are root-major, newest first; all_nodes = roots ++ neighbours; the next layer's
    roots are this layer's all_nodes / all_timestamps.

Written without looking at blocks at all, so agreement on random multi-block graphs with
ties, reverse edges, windows, snapshots and prop_time checks the walk, the LowerBound case
analysis and the block policy's invisibility in one go."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import synth


class FlatGraph:
    def __init__(self):
        self.seq = {}          # node -> list of (ts, dst, eid) in sequence order

    def add_edges(self, src, dst, ts, eid, add_reverse=False):
        if add_reverse:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
            ts, eid = np.concatenate([ts, ts]), np.concatenate([eid, eid])
        for v in np.unique(src):
            idx = np.flatnonzero(src == v)                       # input order
            idx = idx[np.argsort(ts[idx], kind="stable")]        # stable by timestamp
            self.seq.setdefault(int(v), []).extend(
                (np.float32(ts[i]), int(dst[i]), int(eid[i])) for i in idx)


def window(t, snapshot, num_snapshots, w):
    t, w = np.float32(t), np.float32(w)
    if num_snapshots == 1:
        return (np.float32(0) if abs(float(w)) < 1e-6 else np.float32(t - w)), t
    k = np.float32(num_snapshots - snapshot - 1)
    # the reference's `t - k*w` is contracted to one fused multiply-add by nvcc
    end = np.float32(np.float64(t) - np.float64(k) * np.float64(w))
    return np.float32(end - w), end


def sample_layer(g, nodes, ts, fanout, snapshot, num_snapshots, w, prop_time):
    all_nodes, all_ts = list(nodes), list(ts)
    dt, eids, row = [], [], []
    for r, (v, t) in enumerate(zip(nodes, ts)):
        start, end = window(t, snapshot, num_snapshots, w)
        cand = [e for e in g.seq.get(int(v), []) if start <= e[0] < end]
        for e in reversed(cand[-fanout:] if fanout < len(cand) else cand):   # newest first
            all_nodes.append(e[1])
            all_ts.append(np.float32(t) if prop_time else e[0])
            dt.append(np.float32(np.float32(t) - e[0]))
            eids.append(e[2])
            row.append(r)
    return (np.array(all_nodes, np.int64), np.array(all_ts, np.float32),
            np.array(dt, np.float32), np.array(eids, np.int64), np.array(row, np.int64))


CFGS = [
    dict(fanouts=[5], snapshots=1, w=0.0, prop_time=False, reverse=False, min_block=4),
    dict(fanouts=[4, 3], snapshots=1, w=0.0, prop_time=False, reverse=True, min_block=2),
    dict(fanouts=[6, 2], snapshots=1, w=150.0, prop_time=True, reverse=False, min_block=64),
    dict(fanouts=[3, 3], snapshots=3, w=40.0, prop_time=False, reverse=True, min_block=8),
]


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("policy", ["insert", "replace"])
def test_oracle_equals_the_definition(cfg, policy):
    N, E = 60, 1500
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=17, tie_levels=90)
    og = O.OracleGraph(minimum_block_size=cfg["min_block"], insertion_policy=policy)
    fg = FlatGraph()
    for lo in range(0, E, 170):                         # uneven multi-batch ingestion
        sl = slice(lo, lo + 170)
        og.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=cfg["reverse"])
        fg.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=cfg["reverse"])
    s = O.OracleSampler(og, cfg["fanouts"], "recent", num_snapshots=cfg["snapshots"],
                        snapshot_time_window=cfg["w"], prop_time=cfg["prop_time"])
    nodes, t = synth.random_roots(N, 80, 1000.0, seed=3, extra_ids=[N + 5])
    got = s.sample(nodes, t)            # [layer reversed][snapshot]
    got = got[::-1]                     # back to sampling order
    for snap in range(cfg["snapshots"]):
        cur_nodes, cur_ts = nodes, t
        for layer, fanout in enumerate(cfg["fanouts"]):
            want = sample_layer(fg, cur_nodes, cur_ts, fanout, snap, cfg["snapshots"], cfg["w"],
                                cfg["prop_time"])
            b = got[layer][snap]
            assert np.array_equal(b.srcdata["ID"], want[0]), (layer, snap)
            assert np.array_equal(b.srcdata["ts"].view(np.uint32), want[1].view(np.uint32))
            assert np.array_equal(b.edata["dt"].view(np.uint32), want[2].view(np.uint32))
            assert np.array_equal(b.edata["ID"], want[3])
            assert np.array_equal(b.edges()[1], want[4])
            assert np.array_equal(b.edges()[0], len(cur_nodes) + np.arange(len(want[3])))
            cur_nodes, cur_ts = want[0], want[1]
