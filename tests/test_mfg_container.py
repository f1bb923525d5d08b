"""The MFG container on CPU tensors: the DGLBlock surface the reference's callers use
(gnnflow/utils.py:477-481, gnnflow/cache/cache.py:272-400, models/modules/layers.py:100-144)
and the lazy dictionaries behind it."""
import pytest
import torch

from gnnflow_amd import MFGBlock
from gnnflow_amd.mfg import LazyTensorDict


def test_lazy_dict_builds_entries_on_first_access_only():
    calls = []
    d = LazyTensorDict()
    d.set_lazy("ID", lambda: calls.append("ID") or torch.arange(3))
    d.set_lazy("ts", lambda: calls.append("ts") or torch.zeros(3))
    assert "ID" in d and "ts" in d and "h" not in d and len(d) == 2 and calls == []
    assert d["ID"].tolist() == [0, 1, 2] and calls == ["ID"]
    assert d["ID"] is d["ID"] and calls == ["ID"]             # cached after the first access
    assert d.get("h") is None and d.get("ts").shape == (3,) and calls == ["ID", "ts"]
    d["h"] = torch.ones(3, 2)
    assert sorted(d.keys()) == ["ID", "h", "ts"] and len(list(d.items())) == 3
    with pytest.raises(KeyError):
        d["missing"]
    e = LazyTensorDict()
    e.set_lazy("a", lambda: 1)
    assert dict(e.materialize()) == {"a": 1}


def test_block_surface_and_lazy_registration():
    col, row = torch.tensor([2, 3, 4, 5]), torch.tensor([0, 0, 1, 1])
    b = MFGBlock(6, 2, col, row)
    assert (b.num_src_nodes(), b.num_dst_nodes(), b.num_edges()) == (6, 2, 4)
    assert torch.equal(b.edges()[0], col) and torch.equal(b.edges()[1], row)
    assert b.device == torch.device("cpu") and b.to("cpu") is b
    assert b.raw_ids("src") is None and b.raw_ids("e") is None       # not a sampler block
    # a producer registered before the dict exists, and one after
    b.set_lazy("src", "h", lambda: torch.full((6, 2), 7.0))
    b.srcdata["ID"] = torch.arange(6)
    b.set_lazy("src", "g", lambda: torch.zeros(6))
    assert b.srcdata["h"].shape == (6, 2) and b.srcdata["g"].shape == (6,)
    b.edata["f"] = torch.ones(4, 3)
    b.dstdata["out"] = torch.zeros(2)
    assert "f" in b.edata and "out" in b.dstdata
    b.srcdata = {"x": 1}                                               # plain dicts are accepted
    assert b.srcdata == {"x": 1}
    assert "MFGBlock(num_src_nodes=6" in repr(b)
    with pytest.raises(AttributeError):
        b.not_a_field = 1                                             # __slots__: no stray state


def test_callable_edges_are_resolved_once():
    n = []
    b = MFGBlock(3, 1, lambda: n.append(1) or torch.tensor([1, 2]), lambda: torch.tensor([0, 0]),
                 num_edges=2, device=torch.device("cpu"))
    assert b.num_edges() == 2 and n == []
    b.edges()
    b.edges()
    assert n == [1]
