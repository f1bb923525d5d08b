"""The slice of the `dgl` namespace that the reference imports, mapped onto
this package so that GNNFlow's model files run on MFGBlocks without DGL:

    dgl.create_block((col, row), num_src_nodes=, num_dst_nodes=)   temporal_sampler.py:153-157
    dgl.heterograph.DGLBlock                                       type annotations
    dgl.ops.edge_softmax, dgl.function.{copy_src, copy_u, u_mul_e, sum, mean}   layers.py:153-159
    dgl.nn.{SAGEConv, GATConv}                                     graphsage.py:27, gat.py:28
    dgl.utils.shared_mem.{create,get}_shared_mem_array             utils.py:12

`install()` registers these as `sys.modules['dgl' ...]` when the real DGL is not importable
(or `force=True`)."""
import os
import sys
import types

import numpy as np
import torch

from . import function, nn, ops
from .mfg import MFGBlock


def create_block(data_dict, num_src_nodes=None, num_dst_nodes=None, idtype=None, device=None):
    """Block from (source index, destination index) edge tensors — dgl.create_block for the
    single-relation form the reference uses."""
    col, row = data_dict
    col, row = torch.as_tensor(col, dtype=torch.int64), torch.as_tensor(row, dtype=torch.int64)
    if device is not None:
        col, row = col.to(device), row.to(device)
    if num_src_nodes is None:
        num_src_nodes = int(col.max()) + 1 if col.numel() else 0
    if num_dst_nodes is None:
        num_dst_nodes = int(row.max()) + 1 if row.numel() else 0
    return MFGBlock(num_src_nodes, num_dst_nodes, col, row)


_SHM_DIR = "/dev/shm"


def _shm_path(name):
    return os.path.join(_SHM_DIR, "gnnflow_amd_" + name)


def create_shared_mem_array(name, shape, dtype):
    """Tensor in POSIX shared memory that other local ranks open with
    get_shared_mem_array(name, shape, dtype)."""
    n = int(np.prod(shape))
    return torch.from_file(_shm_path(name), shared=True, size=n, dtype=dtype).view(*shape)


def get_shared_mem_array(name, shape, dtype):
    if not os.path.exists(_shm_path(name)):
        raise FileNotFoundError("shared array '{}' has not been created".format(name))
    n = int(np.prod(shape))
    return torch.from_file(_shm_path(name), shared=True, size=n, dtype=dtype).view(*shape)


def install(force: bool = False):
    """Makes `import dgl` resolve to this subset.  Returns the module registered as `dgl`."""
    if not force:
        try:
            import dgl   # noqa: F401
            return sys.modules["dgl"]
        except ImportError:
            pass

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    dgl = module("dgl", create_block=create_block, __path__=[])
    dgl.heterograph = module("dgl.heterograph", DGLBlock=MFGBlock)
    dgl.ops = module("dgl.ops", edge_softmax=ops.edge_softmax)
    dgl.function = module("dgl.function", copy_src=function.copy_src, copy_u=function.copy_u,
                          u_mul_e=function.u_mul_e, sum=function.sum, mean=function.mean,
                          max=function.max)
    dgl.nn = module("dgl.nn", SAGEConv=nn.SAGEConv, GATConv=nn.GATConv)
    dgl.utils = module("dgl.utils", __path__=[])
    dgl.utils.shared_mem = module("dgl.utils.shared_mem",
                                  create_shared_mem_array=create_shared_mem_array,
                                  get_shared_mem_array=get_shared_mem_array)
    dgl.DGLGraph = MFGBlock
    return dgl
