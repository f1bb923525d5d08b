"""TGN memory module on MI355X — `gnnflow.models.modules.memory.Memory`
(gnnflow/models/modules/memory.py:17-269) with the same constructor, attributes
(`node_memory`, `node_memory_ts`, `mailbox`, `mailbox_ts`) and methods (`reset`, `resize`,
`backup`, `restore`, `prepare_input`, `update_mem_mail`); the four tables live in HBM and the
gather / last-writer-wins scatter run as HIP kernels (gnnflow_amd/csrc/memory_ops.hip) instead
of CPU `torch.unique` + index ops."""
import ctypes as C
from typing import Dict, Optional, Union

import torch

from . import _capi


class Memory:
    """
    Memory module proposed by TGN
    """

    def __init__(self, num_nodes: int, dim_edge: int, dim_memory: int,
                 device: Union[torch.device, str] = 'cuda',
                 shared_memory: bool = False, kvstore_client=None):
        if kvstore_client is not None:
            raise NotImplementedError('the multi-machine KVStore memory path is out of scope')
        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError('gnnflow_amd.Memory keeps the memory in HBM; device must be a GPU')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        # shared_memory (one host copy for all local ranks in the reference) is accepted and
        # ignored: every rank keeps its tables in its own HBM
        self.num_nodes = num_nodes
        self.dim_edge = dim_edge
        self.dim_memory = dim_memory
        self.dim_raw_message = 2 * dim_memory + dim_edge
        self.device = device
        self.kvstore_client = None
        self.partition = False
        self._lib = _capi.load()
        f32 = dict(dtype=torch.float32, device=device)
        self.node_memory = torch.zeros((num_nodes, dim_memory), **f32)
        self.node_memory_ts = torch.zeros(num_nodes, **f32)
        self.mailbox = torch.zeros((num_nodes, self.dim_raw_message), **f32)
        self.mailbox_ts = torch.zeros((num_nodes,), **f32)
        self._win = torch.zeros((2, num_nodes), dtype=torch.int64, device=device)
        self._epoch = 0

    def _stream(self):
        return _capi.current_stream(self.device)

    def reset(self):
        """Reset the memory and the mailbox."""
        self.node_memory.fill_(0)
        self.node_memory_ts.fill_(0)
        self.mailbox.fill_(0)
        self.mailbox_ts.fill_(0)

    def resize(self, num_nodes):
        """Resize the memory and the mailbox (new nodes start at zero)."""
        if num_nodes <= self.num_nodes:
            return
        def grow(t, shape):
            n = torch.zeros(shape, dtype=t.dtype, device=self.device)
            n[:t.shape[0]] = t
            return n
        self.node_memory = grow(self.node_memory, (num_nodes, self.dim_memory))
        self.node_memory_ts = grow(self.node_memory_ts, (num_nodes,))
        self.mailbox = grow(self.mailbox, (num_nodes, self.dim_raw_message))
        self.mailbox_ts = grow(self.mailbox_ts, (num_nodes,))
        win = torch.zeros((2, num_nodes), dtype=torch.int64, device=self.device)
        win[:, :self.num_nodes] = self._win
        self._win = win
        self.num_nodes = num_nodes

    def backup(self) -> Dict:
        """Backup the current memory and mailbox."""
        return {
            'node_memory': self.node_memory.clone(),
            'node_memory_ts': self.node_memory_ts.clone(),
            'mailbox': self.mailbox.clone(),
            'mailbox_ts': self.mailbox_ts.clone(),
        }

    def restore(self, backup: Dict):
        """Restore the memory and mailbox from the backup."""
        self.node_memory.copy_(backup['node_memory'])
        self.node_memory_ts.copy_(backup['node_memory_ts'])
        self.mailbox.copy_(backup['mailbox'])
        self.mailbox_ts.copy_(backup['mailbox_ts'])

    def prepare_input(self, b):
        """
        Prepare the input for the memory module (memory.py:156-190): fills
        b.srcdata['mem', 'mem_ts', 'mail_ts', 'mem_input'] for b.srcdata['ID'].
        """
        ids = b.srcdata['ID']
        if ids.device != self.device or ids.dtype != torch.int64 or not ids.is_contiguous():
            ids = ids.to(self.device, torch.int64).contiguous()
        n = int(ids.shape[0])
        f32 = dict(dtype=torch.float32, device=self.device)
        mem = torch.empty((n, self.dim_memory), **f32)
        mem_ts = torch.empty((n,), **f32)
        mail_ts = torch.empty((n,), **f32)
        mem_input = torch.empty((n, self.dim_raw_message), **f32)
        if n:
            with torch.cuda.device(self.device):
                _capi.check(self._lib.gf_memory_prepare_input(
                    self.node_memory.data_ptr(), self.node_memory_ts.data_ptr(),
                    self.mailbox.data_ptr(), self.mailbox_ts.data_ptr(), self.num_nodes,
                    self.dim_memory, self.dim_raw_message, ids.data_ptr(), n, mem.data_ptr(),
                    mem_ts.data_ptr(), mail_ts.data_ptr(), mem_input.data_ptr(),
                    self.device.index, self._stream()))
        b.srcdata['mem'] = mem
        b.srcdata['mem_ts'] = mem_ts
        b.srcdata['mail_ts'] = mail_ts
        b.srcdata['mem_input'] = mem_input

    def update_mem_mail(self, last_updated_nid: torch.Tensor,
                        last_updated_memory: torch.Tensor,
                        last_updated_ts: torch.Tensor,
                        edge_feats: Optional[torch.Tensor] = None,
                        neg_sample_ratio: int = 1):
        """
        Update the mem and mailbox of last updated nodes (memory.py:192-269).
        """
        def dev(t, dtype):
            if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
                t = t.to(self.device, dtype).contiguous()
            return t
        nid = dev(last_updated_nid, torch.int64)
        mem = dev(last_updated_memory, torch.float32)
        ts = dev(last_updated_ts, torch.float32)
        ef = None if edge_feats is None else dev(edge_feats, torch.float32)
        n = int(nid.shape[0])
        if n == 0:
            return
        self._epoch += 1
        with torch.cuda.device(self.device):
            _capi.check(self._lib.gf_memory_update(
                self.node_memory.data_ptr(), self.node_memory_ts.data_ptr(),
                self.mailbox.data_ptr(), self.mailbox_ts.data_ptr(), self.num_nodes,
                self.dim_memory, self.dim_edge, nid.data_ptr(), mem.data_ptr(), ts.data_ptr(),
                None if ef is None else ef.data_ptr(), n, int(neg_sample_ratio),
                self._win[0].data_ptr(), self._win[1].data_ptr(), self._epoch,
                self.device.index, self._stream()))
