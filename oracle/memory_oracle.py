"""
memory_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's TGN memory module
(gnnflow/models/modules/memory.py:156-269).  Duplicate node ids in update_mem_mail: the
reference selects the surviving row with `new_empty(n).scatter_(0, inv, perm)`, which is
nondeterministic by torch's contract and "last occurrence wins" in the CPU kernel; this
oracle (and the HIP path) make the last occurrence win.  Pinned against the reference module
itself run on CPU tensors (tests/golden/make_memory_fixtures.py -> memory_reference.npz).
"""
import numpy as np


class OracleMemory:
    def __init__(self, num_nodes, dim_edge, dim_memory):
        self.num_nodes, self.dim_edge, self.dim_memory = num_nodes, dim_edge, dim_memory
        self.dim_raw_message = 2 * dim_memory + dim_edge
        self.node_memory = np.zeros((num_nodes, dim_memory), np.float32)
        self.node_memory_ts = np.zeros(num_nodes, np.float32)
        self.mailbox = np.zeros((num_nodes, self.dim_raw_message), np.float32)
        self.mailbox_ts = np.zeros(num_nodes, np.float32)

    # memory.py:156-190
    def prepare_input(self, ids):
        ids = np.asarray(ids, np.int64)
        return (self.node_memory[ids], self.node_memory_ts[ids], self.mailbox_ts[ids],
                self.mailbox[ids])

    @staticmethod
    def _last_occurrence(nid):
        """positions of the last occurrence of every distinct id (any order)."""
        last = {}
        for p, v in enumerate(nid.tolist()):
            last[v] = p
        return np.array(sorted(last.values()), np.int64)

    # memory.py:192-269
    def update_mem_mail(self, nid, memory, ts, edge_feats=None, neg_sample_ratio=1):
        nid = np.asarray(nid, np.int64)
        memory = np.asarray(memory, np.float32)
        ts = np.asarray(ts, np.float32)
        chunks = 2 + neg_sample_ratio
        B = len(nid) // chunks
        if edge_feats is None:
            edge_feats = np.zeros((B, self.dim_edge), np.float32)
        src, dst = nid[:B], nid[B:2 * B]
        mem_src, mem_dst = memory[:B], memory[B:2 * B]
        src_mail = np.concatenate([mem_src, mem_dst, edge_feats], axis=1)
        dst_mail = np.concatenate([mem_dst, mem_src, edge_feats], axis=1)
        mail = np.concatenate([src_mail, dst_mail], axis=1).reshape(-1, src_mail.shape[1])
        mnid = np.stack([src, dst], axis=1).reshape(-1)
        mail_ts = ts[:len(mnid)]
        perm = self._last_occurrence(mnid)
        self.mailbox[mnid[perm]] = mail[perm]
        self.mailbox_ts[mnid[perm]] = mail_ts[perm]
        nid2, mem2, ts2 = nid[:2 * B], memory[:2 * B], ts[:2 * B]
        perm = self._last_occurrence(nid2)
        self.node_memory[nid2[perm]] = mem2[perm]
        self.node_memory_ts[nid2[perm]] = ts2[perm]
