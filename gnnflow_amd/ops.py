"""Message passing on an MFGBlock: the DGL calls the reference's layers
make on a sampled block, on HIP segment kernels (csrc/block_ops.hip) with autograd.

    edge_softmax(block, logits)            dgl.ops.edge_softmax      (layers.py:153)
    block.update_all(fn.copy_src('v','m'), fn.sum('m','h'))          (layers.py:159)
    copy_u / u_mul_e messages, sum / mean reducers                   (dgl.nn.SAGEConv / GATConv)

A block's edges are grouped by destination (the sampler emits them that way); blocks built by
hand with unordered edges are handled through a stable permutation.
"""
import ctypes as C

import torch

from . import _capi


def _stream(device):
    return _capi.current_stream(device)


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError("block ops compute in float32, got {}".format(t.dtype))
    return t.contiguous()


def _ptr(t):
    return t.data_ptr() if t is not None and t.numel() else None


class _EdgeSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, offsets, num_dst):
        x = _f32(logits)
        E = x.shape[0]
        heads = x.numel() // E if E else 0
        y = torch.empty_like(x)
        if E:
            with torch.cuda.device(x.device):
                _capi.check(_capi.load().gf_block_edge_softmax(
                    offsets.data_ptr(), num_dst, E, heads, x.data_ptr(), y.data_ptr(),
                    x.device.index, _stream(x.device)))
        ctx.save_for_backward(y, offsets)
        ctx.num_dst = num_dst
        return y

    @staticmethod
    def backward(ctx, grad):
        y, offsets = ctx.saved_tensors
        g = _f32(grad)
        E = y.shape[0]
        gx = torch.empty_like(y)
        if E:
            with torch.cuda.device(y.device):
                _capi.check(_capi.load().gf_block_edge_softmax_backward(
                    offsets.data_ptr(), ctx.num_dst, E, y.numel() // E, y.data_ptr(),
                    g.data_ptr(), gx.data_ptr(), y.device.index, _stream(y.device)))
        return gx, None, None


class _BlockReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, weight, offsets, col, num_dst, mean, num_edges):
        # col is None for the sampler's layout (source of edge k = row num_dst + k)
        s = _f32(src)
        num_src = s.shape[0]
        dim = s.numel() // num_src if num_src else 0
        w, heads = None, 1
        if weight is not None:
            w = _f32(weight)
            E = w.shape[0]
            heads = w.numel() // E if E else 1
            if dim % max(heads, 1):
                raise ValueError("feature size {} is not a multiple of the {} edge-weight heads"
                                 .format(dim, heads))
        out = torch.zeros((num_dst,) + tuple(s.shape[1:]), dtype=torch.float32, device=s.device)
        if num_dst and dim and num_edges:
            with torch.cuda.device(s.device):
                _capi.check(_capi.load().gf_block_reduce(
                    offsets.data_ptr(), num_dst, _ptr(col), s.data_ptr(), dim, _ptr(w), heads,
                    1 if mean else 0, out.data_ptr(), s.device.index, _stream(s.device)))
        ctx.save_for_backward(s, w, offsets, col)
        ctx.meta = (num_dst, mean, heads, dim, num_edges)
        return out

    @staticmethod
    def backward(ctx, grad):
        s, w, offsets, col = ctx.saved_tensors
        num_dst, mean, heads, dim, num_edges = ctx.meta
        g = _f32(grad)
        need_src, need_w = ctx.needs_input_grad[0], w is not None and ctx.needs_input_grad[1]
        gs = torch.empty_like(s) if need_src else None
        gw = torch.zeros_like(w) if need_w else None
        if (need_src or need_w) and dim:
            if num_edges == 0 or num_dst == 0:
                if gs is not None:
                    gs.zero_()
            else:
                with torch.cuda.device(s.device):
                    _capi.check(_capi.load().gf_block_reduce_backward(
                        offsets.data_ptr(), num_dst, _ptr(col), s.data_ptr(), dim, _ptr(w),
                        heads, 1 if mean else 0, g.data_ptr(), _ptr(gs), s.shape[0], _ptr(gw),
                        s.device.index, _stream(s.device)))
        return gs, gw, None, None, None, None, None


class _BlockMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, offsets, col, num_dst, num_edges):
        s = _f32(src)
        num_src = s.shape[0]
        dim = s.numel() // num_src if num_src else 0
        out = torch.zeros((num_dst,) + tuple(s.shape[1:]), dtype=torch.float32, device=s.device)
        arg = torch.full((num_dst, max(dim, 1)), -1, dtype=torch.int64, device=s.device)
        if num_dst and dim and num_edges:
            with torch.cuda.device(s.device):
                _capi.check(_capi.load().gf_block_reduce_max(
                    offsets.data_ptr(), num_dst, _ptr(col), s.data_ptr(), dim, out.data_ptr(),
                    arg.data_ptr(), s.device.index, _stream(s.device)))
        ctx.save_for_backward(arg, col)
        ctx.meta = (num_dst, dim, tuple(s.shape))
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    def backward(ctx, grad):
        arg, col = ctx.saved_tensors
        num_dst, dim, shape = ctx.meta
        g = _f32(grad)
        gs = torch.empty(shape, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _capi.check(_capi.load().gf_block_reduce_max_backward(
                num_dst, _ptr(col), dim, _ptr(g), arg.data_ptr(), gs.data_ptr(), shape[0],
                g.device.index, _stream(g.device)))
        return gs, None, None, None, None


def block_max(block, src: torch.Tensor) -> torch.Tensor:
    """out[d] = element-wise max over the edges into d of src[source(k)] (0 without in-edges):
    update_all(copy_src, max)."""
    if src.shape[0] != block.num_src_nodes():
        raise ValueError("src must have one row per source node")
    offsets, col, _ = block.segments()
    return _BlockMax.apply(src, offsets, col, block.num_dst_nodes(), block.num_edges())


def edge_softmax(block, logits: torch.Tensor) -> torch.Tensor:
    """Softmax of `logits[num_edges, ...]` over the edges that share a destination node
    (dgl.ops.edge_softmax with the default norm_by='dst')."""
    if logits.shape[0] != block.num_edges():
        raise ValueError("logits must have one row per edge")
    offsets, _, perm = block.segments()
    if perm is None:
        return _EdgeSoftmax.apply(logits, offsets, block.num_dst_nodes())
    y = _EdgeSoftmax.apply(logits[perm], offsets, block.num_dst_nodes())
    return torch.empty_like(y).index_copy(0, perm, y)


def block_reduce(block, src: torch.Tensor, edge_weight=None, mean: bool = False) -> torch.Tensor:
    """out[d] = sum (mean) over the edges k into d of edge_weight[k] * src[source(k)].
    src: [num_src_nodes, ...]; edge_weight: None or [num_edges, heads(, 1)], each head
    scaling `feature_size / heads` consecutive values."""
    if src.shape[0] != block.num_src_nodes():
        raise ValueError("src must have one row per source node")
    offsets, col, perm = block.segments()
    if edge_weight is not None and perm is not None:
        edge_weight = edge_weight[perm]
    return _BlockReduce.apply(src, edge_weight, offsets, col, block.num_dst_nodes(), mean,
                              block.num_edges())
