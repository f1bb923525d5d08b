"""
Generates tests/golden/memory_reference.npz by running the REFERENCE's own TGN memory
module (gnnflow/models/modules/memory.py, imported read-only from /root/reference with
empty stubs for the `dgl` / `gnnflow.utils` / kvstore imports it only uses for type names)
on CPU tensors with seeded inputs.  Build-container only; the .npz is what travels.

Run:  python -B tests/golden/make_memory_fixtures.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference_memory():
    dgl = _stub("dgl")
    dgl.heterograph = _stub("dgl.heterograph", DGLBlock=type("DGLBlock", (), {}))
    dgl.utils = _stub("dgl.utils")
    dgl.utils.shared_mem = _stub("dgl.utils.shared_mem", create_shared_mem_array=None,
                                 get_shared_mem_array=None)
    pkg = _stub("gnnflow")
    pkg.__path__ = []
    pkg.utils = _stub("gnnflow.utils", local_rank=lambda: 0)
    dist = _stub("gnnflow.distributed")
    dist.__path__ = []
    _stub("gnnflow.distributed.kvstore", KVStoreClient=type("KVStoreClient", (), {}))
    spec = importlib.util.spec_from_file_location(
        "gnnflow_ref_memory", os.path.join(REF, "gnnflow/models/modules/memory.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Memory


class Blk:
    def __init__(self, ids):
        import torch
        self.srcdata = {"ID": torch.from_numpy(ids)}
        self.device = torch.device("cpu")


def main():
    import torch
    Memory = import_reference_memory()
    rng = np.random.RandomState(7)
    N, de, dm, B, steps = 60, 6, 8, 25, 6
    mem = Memory(N, de, dm, device="cpu")
    out = {"meta": np.array([N, de, dm, B, steps], np.int64)}
    for s in range(steps):
        neg = 1 if s % 2 == 0 else 2
        n = (2 + neg) * B
        nid = rng.randint(0, N, n).astype(np.int64)          # plenty of duplicates
        memory = (rng.randint(0, 64, (n, dm)) / 8.0).astype(np.float32)
        ts = (s * 100 + np.arange(n)).astype(np.float32)
        ef = None if s == 3 else (rng.randint(0, 64, (B, de)) / 8.0).astype(np.float32)
        mem.update_mem_mail(torch.from_numpy(nid), torch.from_numpy(memory),
                            torch.from_numpy(ts),
                            None if ef is None else torch.from_numpy(ef), neg_sample_ratio=neg)
        q = rng.randint(0, N, 40).astype(np.int64)
        b = Blk(q)
        mem.prepare_input(b)
        out["s%d/neg" % s] = np.array([neg], np.int64)
        out["s%d/nid" % s] = nid
        out["s%d/memory" % s] = memory
        out["s%d/ts" % s] = ts
        if ef is not None:
            out["s%d/ef" % s] = ef
        out["s%d/query" % s] = q
        out["s%d/mem" % s] = b.srcdata["mem"].numpy()
        out["s%d/mem_ts" % s] = b.srcdata["mem_ts"].numpy()
        out["s%d/mail_ts" % s] = b.srcdata["mail_ts"].numpy()
        out["s%d/mem_input" % s] = b.srcdata["mem_input"].numpy()
        out["s%d/node_memory" % s] = mem.node_memory.numpy().copy()
        out["s%d/mailbox" % s] = mem.mailbox.numpy().copy()
        out["s%d/node_memory_ts" % s] = mem.node_memory_ts.numpy().copy()
        out["s%d/mailbox_ts" % s] = mem.mailbox_ts.numpy().copy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "memory_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
