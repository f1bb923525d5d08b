"""TGN memory gather / scatter: the numpy oracle is pinned to the
reference's own module (tests/golden/memory_reference.npz, produced by running
gnnflow/models/modules/memory.py on CPU tensors); the HIP path is compared with both."""
import os

import numpy as np
import pytest

from oracle.memory_oracle import OracleMemory

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "memory_reference.npz")


def _replay(make, update, query, state):
    z = np.load(FIX)
    N, de, dm, B, steps = (int(x) for x in z["meta"])
    m = make(N, de, dm)
    for s in range(steps):
        ef = z["s%d/ef" % s] if ("s%d/ef" % s) in z.files else None
        update(m, z["s%d/nid" % s], z["s%d/memory" % s], z["s%d/ts" % s], ef,
               int(z["s%d/neg" % s][0]))
        mem, mem_ts, mail_ts, mem_input = query(m, z["s%d/query" % s])
        assert np.array_equal(mem, z["s%d/mem" % s]), s
        assert np.array_equal(mem_ts, z["s%d/mem_ts" % s]), s
        assert np.array_equal(mail_ts, z["s%d/mail_ts" % s]), s
        assert np.array_equal(mem_input, z["s%d/mem_input" % s]), s
        nm, mb, nmt, mbt = state(m)
        assert np.array_equal(nm, z["s%d/node_memory" % s]), s
        assert np.array_equal(mb, z["s%d/mailbox" % s]), s
        assert np.array_equal(nmt, z["s%d/node_memory_ts" % s]), s
        assert np.array_equal(mbt, z["s%d/mailbox_ts" % s]), s


def test_oracle_matches_reference_module():
    _replay(lambda N, de, dm: OracleMemory(N, de, dm),
            lambda m, nid, mem, ts, ef, neg: m.update_mem_mail(nid, mem, ts, ef, neg),
            lambda m, q: m.prepare_input(q),
            lambda m: (m.node_memory, m.mailbox, m.node_memory_ts, m.mailbox_ts))


class _Blk:
    def __init__(self, ids):
        self.srcdata = {"ID": ids}


@pytest.mark.gpu
def test_hip_matches_reference_module():
    import torch
    from gnnflow_amd.memory import Memory

    def update(m, nid, mem, ts, ef, neg):
        m.update_mem_mail(torch.from_numpy(nid), torch.from_numpy(mem), torch.from_numpy(ts),
                          None if ef is None else torch.from_numpy(ef), neg_sample_ratio=neg)

    def query(m, q):
        b = _Blk(torch.from_numpy(q).cuda())
        m.prepare_input(b)
        return tuple(b.srcdata[k].cpu().numpy() for k in ("mem", "mem_ts", "mail_ts", "mem_input"))

    _replay(lambda N, de, dm: Memory(N, de, dm, device="cuda:0"), update, query,
            lambda m: tuple(t.cpu().numpy() for t in
                            (m.node_memory, m.mailbox, m.node_memory_ts, m.mailbox_ts)))


@pytest.mark.gpu
def test_hip_matches_oracle_tgn_sized():
    """TGN-config sizes (dim_memory 100, dim_edge 172, batch 600 x 3 roots, REDDIT node
    count) over a sequence of batches with heavy id reuse; backup / restore / resize / reset."""
    import torch
    from gnnflow_amd.memory import Memory
    N, de, dm, B = 10984, 172, 100, 600
    rng = np.random.RandomState(0)
    hip = Memory(N, de, dm, device="cuda:0")
    ora = OracleMemory(N, de, dm)
    for s in range(5):
        nid = rng.randint(0, 400, 3 * B).astype(np.int64)
        mem = rng.rand(3 * B, dm).astype(np.float32)
        ts = np.sort(rng.rand(3 * B).astype(np.float32)) + s
        ef = rng.rand(B, de).astype(np.float32)
        hip.update_mem_mail(torch.from_numpy(nid).cuda(), torch.from_numpy(mem).cuda(),
                            torch.from_numpy(ts).cuda(), torch.from_numpy(ef).cuda())
        ora.update_mem_mail(nid, mem, ts, ef)
        q = rng.randint(0, N, 20000).astype(np.int64)
        b = _Blk(torch.from_numpy(q).cuda())
        hip.prepare_input(b)
        want = ora.prepare_input(q)
        for k, w in zip(("mem", "mem_ts", "mail_ts", "mem_input"), want):
            assert np.array_equal(b.srcdata[k].cpu().numpy(), w), (s, k)
    bk = hip.backup()
    hip.reset()
    assert float(hip.mailbox.abs().sum()) == 0.0
    hip.restore(bk)
    assert np.array_equal(hip.node_memory.cpu().numpy(), ora.node_memory)
    hip.resize(N + 100)
    assert hip.node_memory.shape[0] == N + 100
    assert np.array_equal(hip.mailbox[:N].cpu().numpy(), ora.mailbox)
    assert float(hip.mailbox[N:].abs().sum()) == 0.0
