"""GPU: the reference's harness flow through the mirrored helpers —
load_dataset -> build_dynamic_graph -> get_batch -> TemporalSampler.sample -> prepare_input
— against the CPU oracle on the same batches."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_offline_harness_flow_matches_oracle(tmp_path):
    import pandas as pd
    import torch
    from gnnflow_amd import TemporalSampler
    from gnnflow_amd import utils as U
    from oracle import oracle as O
    rng = np.random.RandomState(5)
    n, N = 6000, 400
    src = rng.randint(0, 250, n)
    dst = rng.randint(250, N, n)
    time = np.sort(rng.rand(n) * 5000)
    roll = np.zeros(n, np.int64)
    roll[4200:5100] = 1
    roll[5100:] = 2
    os.makedirs(tmp_path / "TOY")
    pd.DataFrame({"src": src, "dst": dst, "time": time, "ext_roll": roll}).to_csv(
        tmp_path / "TOY" / "edges.csv")
    nf = rng.rand(N, 12).astype(np.float32)
    ef = rng.rand(n, 20).astype(np.float32)
    np.save(tmp_path / "TOY" / "node_features.npy", nf)
    np.save(tmp_path / "TOY" / "edge_features.npy", ef)

    train, val, test, full = U.load_dataset("TOY", data_dir=str(tmp_path))
    node_feats, edge_feats = U.load_feat("TOY", data_dir=str(tmp_path))
    g = U.build_dynamic_graph(1 << 20, 64 << 20, "cuda", 16, 64, "insert", undirected=True,
                              device=0, dataset_df=full, unused_model_kwarg=1)
    assert g.num_edges() == n and g.max_vertex_id() == max(src.max(), dst.max())
    sampler = TemporalSampler(g, fanouts=[5, 5], sample_strategy="recent")

    og = O.OracleGraph(minimum_block_size=16)
    og.add_edges(full["src"].values.astype(np.int64), full["dst"].values.astype(np.int64),
                 full["time"].values.astype(np.float32), full["eid"].values.astype(np.int64),
                 add_reverse=True)
    osamp = O.OracleSampler(og, [5, 5], "recent")

    edge_dev = edge_feats.cuda()          # device table; node table stays a host tensor
    neg = U.DstRandEdgeSampler(full["dst"].to_numpy(dtype=np.int32), seed=3)
    nb = 0
    for roots, ts, eid in U.get_batch(val, 300, 0, neg):
        mfgs = sampler.sample(roots, ts)
        mfgs = U.mfgs_to_cuda(mfgs, "cuda:0")
        U.prepare_input(mfgs, node_feats, edge_dev)
        want = osamp.sample(roots, ts)
        for gl, wl in zip(mfgs, want):
            for gb, wb in zip(gl, wl):
                ids = gb.srcdata["ID"].cpu().numpy()
                eids = gb.edata["ID"].cpu().numpy()
                assert np.array_equal(ids, wb.srcdata["ID"])
                assert np.array_equal(eids, wb.edata["ID"])
                assert np.array_equal(gb.edata["f"].cpu().numpy(), ef[eids])
        h = mfgs[0][0].srcdata["h"]
        assert h.dtype == torch.float32 and h.is_cuda
        assert np.array_equal(h.cpu().numpy(), nf[mfgs[0][0].srcdata["ID"].cpu().numpy()])
        assert "h" not in mfgs[1][0].srcdata        # node features only for mfgs[0]
        nb += 1
    assert nb == 3     # val rows 4200..5099: index-aligned batches of 300
