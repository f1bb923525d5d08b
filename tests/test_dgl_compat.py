"""The dgl subset (gnnflow_amd.dgl_compat): registration, create_block, shared-memory arrays;
and — in the build container only — that the reference's own model files import against it."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"


@pytest.fixture()
def dgl():
    from gnnflow_amd import dgl_compat
    saved = {k: v for k, v in sys.modules.items() if k == "dgl" or k.startswith("dgl.")}
    mod = dgl_compat.install(force=True)
    yield mod
    for k in [k for k in sys.modules if k == "dgl" or k.startswith("dgl.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_namespace_and_create_block(dgl):
    import dgl as d
    import dgl.function as fn
    import dgl.nn as dglnn
    from dgl.heterograph import DGLBlock
    from dgl.utils.shared_mem import create_shared_mem_array, get_shared_mem_array  # noqa: F401
    assert d is dgl and callable(d.ops.edge_softmax)
    assert fn.copy_src('v', 'm').out == 'm' and fn.sum('m', 'h').kind == "sum"
    assert dglnn.SAGEConv and dglnn.GATConv
    col, row = torch.tensor([3, 4, 5]), torch.tensor([0, 0, 2])
    b = d.create_block((col, row), num_src_nodes=6, num_dst_nodes=3)
    assert isinstance(b, DGLBlock)
    assert (b.num_src_nodes(), b.num_dst_nodes(), b.num_edges()) == (6, 3, 3)
    assert torch.equal(b.edges()[0], col) and torch.equal(b.edges()[1], row)
    b.srcdata['ID'] = torch.arange(6)
    assert b.srcdata['ID'].shape[0] == 6 and b.device == torch.device("cpu")
    assert d.create_block((col, row)).num_src_nodes() == 6


def test_shared_mem_arrays(dgl):
    from dgl.utils.shared_mem import create_shared_mem_array, get_shared_mem_array
    name = "test_{}".format(os.getpid())
    path = "/dev/shm/gnnflow_amd_" + name
    try:
        a = create_shared_mem_array(name, (4, 3), torch.float32)
        a[:] = torch.arange(12, dtype=torch.float32).view(4, 3)
        b = get_shared_mem_array(name, (4, 3), torch.float32)
        assert torch.equal(a, b)
        b[1, 1] = 99.0
        assert a[1, 1] == 99.0           # same memory
        with pytest.raises(FileNotFoundError):
            get_shared_mem_array(name + "_missing", (1,), torch.float32)
    finally:
        if os.path.exists(path):
            os.remove(path)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_reference_model_files_import_against_the_subset(dgl):
    """gnnflow/models/{modules/layers,graphsage,gat}.py import dgl, dgl.function, dgl.nn and
    DGLBlock at module level; they must load and construct with the subset alone."""
    sys.dont_write_bytecode = True
    mods = {}
    for name, rel in (("layers", "gnnflow/models/modules/layers.py"),
                      ("graphsage", "gnnflow/models/graphsage.py"),
                      ("gat", "gnnflow/models/gat.py")):
        spec = importlib.util.spec_from_file_location("_ref_" + name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    sage = mods["graphsage"].SAGE(16, 8, num_layers=2)
    gat = mods["gat"].GAT(16, 8, num_layers=2, attn_head=[2, 2])
    att = mods["layers"].TransfomerAttentionLayer(16, 4, 8, 8, 2, dropout=0.1, att_dropout=0.1)
    from gnnflow_amd import nn as gnn
    assert isinstance(sage.layers['l0h0'], gnn.SAGEConv)
    assert isinstance(gat.layers['l1h0'], gnn.GATConv)
    assert att.w_q.in_features == 16 + 8
