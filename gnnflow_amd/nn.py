"""dgl.nn.SAGEConv / dgl.nn.GATConv for MFGBlocks — the two DGL layers the
reference's static models instantiate (gnnflow/models/graphsage.py:27-31, gat.py:28-46), with
dgl's constructor arguments, parameter names and formulas, on the block ops of
gnnflow_amd.ops.  dgl (requirements.txt: dgl >= 0.7) is not vendored in the reference; these
follow its documented layer definitions."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class SAGEConv(nn.Module):
    """GraphSAGE layer: h_i' = W_self h_i + W_neigh * AGG_{j in N(i)} h_j + b.
    aggregator_type 'mean', 'gcn' or 'pool' ('lstm' is not built)."""

    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0., bias=True, norm=None,
                 activation=None):
        super().__init__()
        if aggregator_type not in ('mean', 'gcn', 'pool'):
            raise NotImplementedError(
                "SAGEConv aggregator '{}' (only 'mean', 'gcn' and 'pool')".format(aggregator_type))
        self._in_src_feats = self._in_dst_feats = in_feats
        self._out_feats = out_feats
        self._aggre_type = aggregator_type
        self.norm = norm
        self.feat_drop = nn.Dropout(feat_drop)
        self.activation = activation
        if aggregator_type == 'pool':
            self.fc_pool = nn.Linear(in_feats, in_feats)
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        if aggregator_type != 'gcn':
            self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_feats))
        else:
            self.register_buffer('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain('relu')
        if self._aggre_type == 'pool':
            nn.init.xavier_uniform_(self.fc_pool.weight, gain=gain)
        if self._aggre_type != 'gcn':
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, graph, feat, edge_weight=None):
        feat_src = self.feat_drop(feat)
        feat_dst = feat_src[:graph.num_dst_nodes()]
        h_self = feat_dst
        if graph.num_edges() == 0:
            h_neigh = torch.zeros((graph.num_dst_nodes(), self._in_src_feats),
                                  dtype=feat_dst.dtype, device=feat_dst.device)
            if self._aggre_type != 'gcn':
                h_neigh = self.fc_neigh(h_neigh)
        else:
            w = None if edge_weight is None else edge_weight.reshape(graph.num_edges(), 1)
            # linear before message passing when it shrinks the rows
            lin_before_mp = self._in_src_feats > self._out_feats
            if self._aggre_type == 'mean':
                src = self.fc_neigh(feat_src) if lin_before_mp else feat_src
                h_neigh = ops.block_reduce(graph, src, w, mean=True)
                if not lin_before_mp:
                    h_neigh = self.fc_neigh(h_neigh)
            elif self._aggre_type == 'pool':   # max over relu(fc_pool(h_j))
                if w is not None:
                    raise NotImplementedError("SAGEConv('pool') with edge weights")
                h_neigh = self.fc_neigh(ops.block_max(graph, F.relu(self.fc_pool(feat_src))))
            else:   # gcn: (sum of neighbours + self) / (degree + 1)
                src = self.fc_neigh(feat_src) if lin_before_mp else feat_src
                dst = src[:graph.num_dst_nodes()]
                total = ops.block_reduce(graph, src, w, mean=False)
                degs = graph.in_degrees().to(total.dtype)
                h_neigh = (total + dst) / (degs.unsqueeze(-1) + 1)
                if not lin_before_mp:
                    h_neigh = self.fc_neigh(h_neigh)
        rst = h_neigh if self._aggre_type == 'gcn' else self.fc_self(h_self) + h_neigh
        if self.bias is not None:
            rst = rst + self.bias
        if self.activation is not None:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst


class GATConv(nn.Module):
    """Graph attention layer: e_ij = LeakyReLU(a_l . W h_j + a_r . W h_i),
    alpha = edge_softmax(e), h_i' = sum_j alpha_ij W h_j (+ residual, bias, activation);
    returns [num_dst, num_heads, out_feats]."""

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0.,
                 negative_slope=0.2, residual=False, activation=None,
                 allow_zero_in_degree=False, bias=True):
        super().__init__()
        self._num_heads = num_heads
        self._in_src_feats = self._in_dst_feats = in_feats
        self._out_feats = out_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        if bias:
            self.bias = nn.Parameter(torch.zeros(num_heads * out_feats))
        else:
            self.register_buffer('bias', None)
        if residual:
            if in_feats != out_feats * num_heads:
                self.res_fc = nn.Linear(in_feats, num_heads * out_feats, bias=False)
            else:
                self.res_fc = nn.Identity()
        else:
            self.register_buffer('res_fc', None)
        self.activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain('relu')
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def forward(self, graph, feat, get_attention=False):
        if not self._allow_zero_in_degree and graph.num_dst_nodes() and \
                bool((graph.in_degrees() == 0).any()):
            raise RuntimeError(
                "There are 0-in-degree nodes in the graph, output for those nodes will be "
                "invalid. Set allow_zero_in_degree=True to suppress this check.")
        num_dst, H, D = graph.num_dst_nodes(), self._num_heads, self._out_feats
        h_src = self.feat_drop(feat)
        feat_src = self.fc(h_src).view(-1, H, D)
        feat_dst = feat_src[:num_dst]
        el = (feat_src * self.attn_l).sum(dim=-1)        # [num_src, H]
        er = (feat_dst * self.attn_r).sum(dim=-1)        # [num_dst, H]
        col, row = graph.edges()
        e = self.leaky_relu(el[col] + er[row])           # u_add_v
        a = self.attn_drop(ops.edge_softmax(graph, e))   # [E, H]
        rst = ops.block_reduce(graph, feat_src, a)       # u_mul_e + sum -> [num_dst, H, D]
        if self.res_fc is not None:
            rst = rst + self.res_fc(h_src[:num_dst]).view(num_dst, H, D)
        if self.bias is not None:
            rst = rst + self.bias.view(1, H, D)
        if self.activation:
            rst = self.activation(rst)
        if get_attention:
            return rst, a.unsqueeze(-1)
        return rst
