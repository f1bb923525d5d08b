"""The message / reduce descriptors of dgl.function that the reference's models use with
`block.update_all` (gnnflow/models/modules/layers.py:159; dgl.nn.SAGEConv / GATConv)."""
import collections

Message = collections.namedtuple("Message", "kind src edge out")
Reduce = collections.namedtuple("Reduce", "kind msg out")


def copy_u(u, out):
    """message = source feature `u`"""
    return Message("copy_u", u, None, out)


copy_src = copy_u      # the name older dgl (and the reference) uses


def u_mul_e(u, e, out):
    """message = source feature `u` times edge feature `e` ([E, H, 1] against [N, H, D])"""
    return Message("u_mul_e", u, e, out)


def sum(msg, out):   # noqa: A001 - dgl's name
    return Reduce("sum", msg, out)


def mean(msg, out):
    return Reduce("mean", msg, out)


def max(msg, out):   # noqa: A001 - dgl's name
    return Reduce("max", msg, out)
