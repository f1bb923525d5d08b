"""GPU numerics of the block message-passing ops (csrc/block_ops.hip) against a plain
PyTorch fp32 reference of the same op (index_add / scatter formulations), forward and
backward, on sampler-shaped blocks (edges grouped by destination, <= fanout per node), on a
long-segment block (wave kernels) and on a hand-built block with unordered edges.
Tolerance: 1e-5 absolute on O(1) values (fp32 sums of <= a few hundred terms)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


def _block(num_dst, degs, seed, shuffle=False, sampler_like=True):
    import torch
    from gnnflow_amd import MFGBlock
    rng = np.random.RandomState(seed)
    row = np.repeat(np.arange(num_dst), degs).astype(np.int64)
    E = len(row)
    if sampler_like:
        col = num_dst + np.arange(E, dtype=np.int64)       # roots ++ one new node per edge
        num_src = num_dst + E
    else:
        num_src = num_dst + max(E // 3, 1)
        col = rng.randint(0, num_src, E).astype(np.int64)  # sources feed several edges
    if shuffle:
        p = rng.permutation(E)
        row, col = row[p], col[p]
    dev = torch.device("cuda", 0)
    return MFGBlock(num_src, num_dst, torch.from_numpy(col).to(dev), torch.from_numpy(row).to(dev))


def ref_edge_softmax(row, x, num_dst):
    import torch
    m = torch.full((num_dst,) + x.shape[1:], -float("inf"), device=x.device)
    m = m.scatter_reduce(0, row.view(-1, *[1] * (x.dim() - 1)).expand_as(x), x, "amax")
    ex = torch.exp(x - m[row])
    s = torch.zeros_like(m).index_add(0, row, ex)
    return ex / s[row]


def ref_reduce(col, row, src, w, num_dst, mean):
    import torch
    msg = src[col]
    if w is not None and len(col):
        E, H = w.shape[0], w.shape[1]
        msg = (msg.reshape(E, H, -1) * w.reshape(E, H, 1)).reshape(msg.shape)
    out = torch.zeros((num_dst,) + src.shape[1:], device=src.device).index_add(0, row, msg)
    if mean:
        deg = torch.zeros(num_dst, device=src.device).index_add(
            0, row, torch.ones(len(row), device=src.device)).clamp(min=1)
        out = out / deg.view(-1, *[1] * (src.dim() - 1))
    return out


CASES = [
    dict(num_dst=1800, maxdeg=10, shuffle=False, sampler_like=True),    # a sampled layer
    dict(num_dst=37, maxdeg=300, shuffle=False, sampler_like=False),    # long segments (wave path)
    dict(num_dst=200, maxdeg=6, shuffle=True, sampler_like=False),      # unordered hand-built block
    dict(num_dst=5, maxdeg=0, shuffle=False, sampler_like=True),        # no edges at all
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("heads", [1, 2])
def test_edge_softmax_forward_backward(case, heads):
    import torch
    from gnnflow_amd import ops
    rng = np.random.RandomState(1)
    degs = rng.randint(0, case["maxdeg"] + 1, case["num_dst"])
    b = _block(case["num_dst"], degs, 2, case["shuffle"], case["sampler_like"])
    col, row = b.edges()
    E = b.num_edges()
    x = (torch.randn(E, heads, device="cuda") * 3).requires_grad_()
    xr = x.detach().clone().requires_grad_()
    y = ops.edge_softmax(b, x)
    yr = ref_edge_softmax(row, xr, b.num_dst_nodes())
    assert torch.allclose(y, yr, **TOL)
    g = torch.randn_like(y)
    y.backward(g)
    yr.backward(g)
    assert torch.allclose(x.grad, xr.grad, **TOL)
    if E:
        sums = torch.zeros(b.num_dst_nodes(), heads, device="cuda").index_add(0, row, y.detach())
        assert torch.allclose(sums[degs > 0], torch.ones_like(sums[degs > 0]), atol=1e-5)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["copy_sum", "copy_mean", "mul_sum"])
def test_update_all_forward_backward(case, mode):
    import torch
    import gnnflow_amd.function as fn
    rng = np.random.RandomState(3)
    degs = rng.randint(0, case["maxdeg"] + 1, case["num_dst"])
    b = _block(case["num_dst"], degs, 4, case["shuffle"], case["sampler_like"])
    col, row = b.edges()
    H, D = 2, 50
    v = torch.randn(b.num_src_nodes(), H, D, device="cuda").requires_grad_()
    vr = v.detach().clone().requires_grad_()
    w = wr = None
    if mode == "mul_sum":
        w = torch.rand(b.num_edges(), H, 1, device="cuda").requires_grad_()
        wr = w.detach().clone().requires_grad_()
        b.srcdata["v"], b.edata["a"] = v, w
        b.update_all(fn.u_mul_e("v", "a", "m"), fn.sum("m", "h"))
    else:
        b.srcdata["v"] = v
        red = fn.mean if mode == "copy_mean" else fn.sum
        b.update_all(fn.copy_src("v", "m"), red("m", "h"))
    out = b.dstdata["h"]
    want = ref_reduce(col, row, vr, wr, b.num_dst_nodes(), mode == "copy_mean")
    assert out.shape == want.shape and torch.allclose(out, want, rtol=1e-5, atol=2e-5)
    g = torch.randn_like(out)
    out.backward(g)
    want.backward(g)
    assert torch.allclose(v.grad, vr.grad, rtol=1e-5, atol=2e-5)
    if w is not None and b.num_edges():
        assert torch.allclose(w.grad, wr.grad, rtol=1e-4, atol=1e-4)


def test_attention_layer_pattern_on_sampled_blocks():
    """The reference's TransformerAttentionLayer tail (layers.py:153-159) on real sampler
    output: edge_softmax over Q.K scores, V weighting, copy_src/sum into the roots."""
    import torch
    import gnnflow_amd.function as fn
    from gnnflow_amd import DynamicGraph, TemporalSampler, ops
    rng = np.random.RandomState(8)
    N, E = 300, 5000
    src, dst = rng.randint(0, N, E), rng.randint(0, N, E)
    ts = np.sort(rng.rand(E)).astype(np.float32)
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 16, 64, "insert")
    g.add_edges(src.astype(np.int64), dst.astype(np.int64), ts, add_reverse=True)
    sampler = TemporalSampler(g, [10, 10], "recent")
    mfgs = sampler.sample(rng.randint(0, N, 500).astype(np.int64), np.full(500, 2.0, np.float32))
    for b in (mfgs[0][0], mfgs[1][0]):
        col, row = b.edges()
        nd, ne, H = b.num_dst_nodes(), b.num_edges(), 2
        scores = torch.randn(ne, H, device="cuda")
        V = torch.randn(ne, H * 25, device="cuda")
        att = ops.edge_softmax(b, scores)
        Vw = torch.reshape(V.reshape(ne, H, -1) * att[:, :, None], (ne, -1))
        b.srcdata['v'] = torch.cat((torch.zeros((nd, Vw.shape[1]), device="cuda"), Vw), dim=0)
        b.update_all(fn.copy_src('v', 'm'), fn.sum('m', 'h'))
        att_ref = ref_edge_softmax(row, scores, nd)
        want = torch.zeros(nd, H * 25, device="cuda").index_add(
            0, row, (V.reshape(ne, H, -1) * att_ref[:, :, None]).reshape(ne, -1))
        assert torch.allclose(b.dstdata['h'], want, rtol=1e-5, atol=2e-5)
        assert bool((b.in_degrees() <= 10).all())


def test_sage_and_gat_layers_match_torch_reference():
    import torch
    import torch.nn.functional as F
    from gnnflow_amd import nn as gnn
    torch.manual_seed(0)
    degs = np.random.RandomState(5).randint(0, 8, 300)
    b = _block(300, degs, 6)
    col, row = b.edges()
    x = torch.randn(b.num_src_nodes(), 48, device="cuda")
    for agg, out_dim in (("mean", 32), ("mean", 64), ("gcn", 32)):
        layer = gnn.SAGEConv(48, out_dim, agg).cuda()
        got = layer(b, x)
        deg = torch.from_numpy(degs).cuda().float()
        if agg == "mean":
            neigh = ref_reduce(col, row, x, None, 300, True)
            want = layer.fc_self(x[:300]) + layer.fc_neigh(neigh) + layer.bias
        else:
            total = ref_reduce(col, row, x, None, 300, False)
            want = layer.fc_neigh((total + x[:300]) / (deg[:, None] + 1)) + layer.bias
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4), agg
    with pytest.raises(NotImplementedError):
        gnn.SAGEConv(8, 8, "lstm")

    gat = gnn.GATConv(48, 16, 3, activation=F.elu, allow_zero_in_degree=True).cuda()
    got, att = gat(b, x, get_attention=True)
    ft = gat.fc(x).view(-1, 3, 16)
    el, er = (ft * gat.attn_l).sum(-1), (ft[:300] * gat.attn_r).sum(-1)
    a = ref_edge_softmax(row, F.leaky_relu(el[col] + er[row], 0.2), 300)
    want = torch.zeros(300, 3, 16, device="cuda").index_add(0, row, ft[col] * a[:, :, None])
    want = F.elu(want + gat.bias.view(1, 3, 16))
    assert got.shape == (300, 3, 16) and torch.allclose(got, want, rtol=1e-4, atol=1e-4)
    assert torch.allclose(att.squeeze(-1), a, rtol=1e-5, atol=1e-5)
    strict = gnn.GATConv(48, 16, 3).cuda()
    with pytest.raises(RuntimeError):
        strict(b, x)       # some destinations have no in-edge
    # gradients flow to every parameter through the HIP ops
    got.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in gat.parameters())


@pytest.mark.parametrize("case", CASES)
def test_max_reducer_forward_backward(case):
    """update_all(copy_src, max) — the reducer behind SAGEConv('pool'): element-wise max over
    in-edges, 0 for destinations without any, gradient to the winning source only."""
    import torch
    import gnnflow_amd.function as fn
    rng = np.random.RandomState(9)
    degs = rng.randint(0, case["maxdeg"] + 1, case["num_dst"])
    b = _block(case["num_dst"], degs, 10, case["shuffle"], case["sampler_like"])
    col, row = b.edges()
    nd, D = b.num_dst_nodes(), 24
    v = torch.randn(b.num_src_nodes(), D, device="cuda").requires_grad_()
    vr = v.detach().clone().requires_grad_()
    b.srcdata["v"] = v
    b.update_all(fn.copy_src("v", "m"), fn.max("m", "h"))
    out = b.dstdata["h"]
    want = torch.zeros(nd, D, device="cuda")
    if b.num_edges():
        want = want.scatter_reduce(0, row[:, None].expand(-1, D), vr[col], "amax",
                                   include_self=False)
    assert torch.equal(out, want)
    g = torch.randn_like(out)
    out.backward(g)
    if b.num_edges():
        want.backward(g)
        assert torch.allclose(v.grad, vr.grad, **TOL)
    else:
        assert torch.count_nonzero(v.grad) == 0


def test_sage_pool_layer_matches_torch_reference():
    import torch
    import torch.nn.functional as F
    from gnnflow_amd import nn as gnn
    torch.manual_seed(1)
    degs = np.random.RandomState(11).randint(0, 8, 200)
    b = _block(200, degs, 12)
    col, row = b.edges()
    x = torch.randn(b.num_src_nodes(), 20, device="cuda")
    layer = gnn.SAGEConv(20, 16, "pool").cuda()
    got = layer(b, x)
    pooled = torch.zeros(200, 20, device="cuda").scatter_reduce(
        0, row[:, None].expand(-1, 20), F.relu(layer.fc_pool(x))[col], "amax", include_self=False)
    want = layer.fc_self(x[:200]) + layer.fc_neigh(pooled) + layer.bias
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)
    got.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())
