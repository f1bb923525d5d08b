"""Hash-partitioned sampling across the GPUs of one node.

The reference partitions the graph by source vertex across *machines* and samples remote
roots through torch RPC (gnnflow/distributed/dist_sampler.py:159-242, merge at :244-314).
Here the partitions are the GPUs of one node and the exchange is two all-to-all-v per layer
over RCCL/xGMI (`torch.distributed`, backend "nccl" on ROCm; "gloo" in the CPU tests):

    1. bucket this rank's roots by owner(root) = splitmix64(root) mod P
    2. all-to-all-v of packed (root id, root ts) to the OTHER ranks   16 B / root, asynchronous
    3. meanwhile the rank samples its OWN share of the roots on its shard (local HIP sampler):
       the remote pull of a layer overlaps that layer's local sample
    4. every rank samples the roots it received; all-to-all-v back: a fixed `fanout` slots per
       root of (dst, eid, ts|dt), invalid slots marked -1, so the reply sizes follow from the
       request sizes (no count exchange)
    5. the requester splices its own share back in, restores the original root order and
       keeps the valid slots

Step 5 restores the original root order, so for most-recent sampling the MFG is
bit-identical to what a single GPU holding the whole graph returns (the reference's merge
is partition-major instead, dist_sampler.py:294-299).  Uniform sampling stays
distribution-matched (each owner draws from its own Philox stream).

Everything is expressed on torch tensors of whatever device the process group works on, so
the same code runs on HBM tensors over RCCL and on CPU tensors over gloo; the local sampler
is injected (`local_sample_layer`), in production `gnnflow_amd.TemporalSampler.sample_layer`.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .temporal_sampler import PendingSample as _PendingSample

_MASK = (1 << 64) - 1


def splitmix64_np(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser (Steele, Lea, Flood 2014) on uint64 numpy arrays."""
    z = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    """logical shift right on int64 tensors (torch's >> is arithmetic)."""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _wrap(c: int) -> int:
    """64-bit constant as a signed python int (torch int64 arithmetic wraps)."""
    return c - (1 << 64) if c >= (1 << 63) else c


def owner_of(nodes: torch.Tensor, world_size: int) -> torch.Tensor:
    """owner(v) = splitmix64(v) mod P, deterministic on every rank and device (the
    reference's HashPartitioner uses Python's salted hash(str(v)) % P,
    gnnflow/distributed/partition.py:324, which is not reproducible across processes)."""
    z = nodes.to(torch.int64) + _wrap(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _wrap(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _wrap(0x94D049BB133111EB)
    z = z ^ _lsr(z, 31)
    # unsigned modulo of the 64-bit pattern: fold the top bit in
    lo = z & ((1 << 63) - 1)
    top = _lsr(z, 63)
    return ((lo % world_size) + top * ((1 << 63) % world_size)) % world_size


def owner_of_np(nodes: np.ndarray, world_size: int) -> np.ndarray:
    return (splitmix64_np(np.asarray(nodes, dtype=np.int64).astype(np.uint64))
            % np.uint64(world_size)).astype(np.int64)


class PartitionedGraph:
    """Wraps a per-rank DynamicGraph: `add_edges` keeps only the edges whose source this
    rank owns (edges are partitioned by source vertex, gnnflow/distributed/partition.py:25-26).
    Every rank is given the full batch, as the reference's dispatcher would route it."""

    def __init__(self, local_graph, rank: int, world_size: int):
        self.local = local_graph
        self.rank, self.world_size = rank, world_size
        self._global_edges = 0

    def add_edges(self, source_vertices, target_vertices, timestamps, eids=None,
                  add_reverse=False):
        src = np.asarray(source_vertices, dtype=np.int64)
        dst = np.asarray(target_vertices, dtype=np.int64)
        ts = np.asarray(timestamps, dtype=np.float32)
        if eids is None:   # gnnflow/dynamic_graph.py:111-113, on the global edge count
            eids = np.arange(self._global_edges, self._global_edges + len(src))
        eids = np.asarray(eids, dtype=np.int64)
        self._global_edges += len(src)
        if add_reverse:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
            ts, eids = np.concatenate([ts, ts]), np.concatenate([eids, eids])
        keep = owner_of_np(src, self.world_size) == self.rank
        if keep.any():
            self.local.add_edges(src[keep], dst[keep], ts[keep], eids[keep])


LayerFn = Callable[[torch.Tensor, torch.Tensor, int, int], "object"]


def _block_arrays(block, device):
    """(counts per root, dst, eid, ts_out, dt) of a locally sampled block (MFG-like:
    srcdata['ID','ts'], edata['dt','ID'], edges() -> (col,row), num_dst_nodes())."""
    def t(x, dtype):
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x))
        return x.to(device=device, dtype=dtype)
    R = block.num_dst_nodes()
    row = t(block.edges()[1], torch.int64)
    counts = torch.bincount(row, minlength=R) if R else torch.zeros(0, dtype=torch.int64, device=device)
    return (counts, t(block.srcdata["ID"], torch.int64)[R:], t(block.edata["ID"], torch.int64),
            t(block.srcdata["ts"], torch.float32)[R:], t(block.edata["dt"], torch.float32))


class LayerResult:
    """One (layer, snapshot) MFG in the reference's SamplingResult layout."""

    def __init__(self, all_nodes, all_ts, dt, eids, row, col, num_dst):
        self.srcdata = {"ID": all_nodes, "ts": all_ts}
        self.edata = {"dt": dt, "ID": eids}
        self._row, self._col, self._R = row, col, int(num_dst)

    def num_dst_nodes(self): return self._R
    def num_src_nodes(self): return int(self.srcdata["ID"].shape[0])
    def num_edges(self): return int(self._row.shape[0])
    def edges(self): return self._col, self._row


def _all_to_all_v(send: torch.Tensor, send_counts: List[int], recv_counts: List[int], group,
                  async_op: bool = False):
    """all-to-all-v of rows; async_op=True returns (recv, work): the collective runs on the
    backend's own stream / thread and `work.wait()` orders the caller behind it, so whatever
    the caller enqueues in between overlaps the exchange."""
    recv = send.new_empty((sum(recv_counts),) + tuple(send.shape[1:]))
    work = dist.all_to_all_single(recv, send, output_split_sizes=recv_counts,
                                  input_split_sizes=send_counts, group=group,
                                  async_op=async_op)
    return (recv, work) if async_op else recv


def _pack_f32_pair(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """two float32 vectors -> one int64 vector (bit pattern), so ids and timestamps travel
    in a single all-to-all"""
    p = torch.stack([a, b], dim=1)
    return p.new_empty(p.shape).copy_(p).view(torch.int64).reshape(-1)


def _unpack_f32_pair(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    # a fresh dense copy: .contiguous() keeps a stride-2 view when it has <= 1 element,
    # and a dtype-changing view needs stride 1
    f = x.new_empty(x.shape).copy_(x).view(torch.float32).reshape(-1, 2)
    return f[:, 0].contiguous(), f[:, 1].contiguous()


class PartitionedSampler:
    """TemporalSampler over a hash-partitioned graph: same `sample()` result layout as
    the single-GPU sampler ([layer][snapshot] reversed, gnnflow/temporal_sampler.py:149-165).

    Three collectives per layer: per-destination root counts; one
    all-to-all-v of packed (root id, root ts) requests; one all-to-all-v of replies with a
    FIXED `fanout` slots per root (invalid slots marked), so the reply sizes follow from the
    request sizes and no second count exchange is needed."""

    def __init__(self, local_sample_layer: LayerFn, fanouts: Sequence[int],
                 num_snapshots: int = 1, group=None, device: Optional[torch.device] = None):
        self._local = local_sample_layer
        self._fanouts = [int(f) for f in fanouts]
        self._L, self._S = len(self._fanouts), int(num_snapshots)
        self._group = group
        self._P = dist.get_world_size(group)
        self._rank = dist.get_rank(group)
        self._device = device or torch.device("cpu")

    def _sample_padded(self, nodes: torch.Tensor, ts: torch.Tensor, layer: int,
                       snapshot: int) -> torch.Tensor:
        """Samples `nodes` on the local shard and lays the result out as a FIXED `fanout`
        slots per root: [n, F, 3] int64 = (dst, eid, packed(ts, dt)), unused slots -1."""
        dev, F = self._device, self._fanouts[layer]
        n = int(nodes.shape[0])
        pad = torch.full((n, F, 3), -1, dtype=torch.int64, device=dev)
        if n == 0:
            return pad
        blk = self._local(nodes, ts, layer, snapshot)
        counts, dst, eid, ts_out, dt = _block_arrays(blk, dev)
        if dst.shape[0]:
            row_l = torch.repeat_interleave(torch.arange(n, device=dev), counts)
            base_l = torch.cumsum(counts, 0) - counts
            pos = torch.arange(dst.shape[0], device=dev) - base_l[row_l]
            pad[row_l, pos, 0] = dst
            pad[row_l, pos, 1] = eid
            pad[row_l, pos, 2] = _pack_f32_pair(ts_out, dt)
        return pad

    def sample_layer(self, nodes: torch.Tensor, ts: torch.Tensor, layer: int,
                     snapshot: int) -> LayerResult:
        dev, P, F, me = self._device, self._P, self._fanouts[layer], self._rank
        nodes = nodes.to(dev, torch.int64)
        ts = ts.to(dev, torch.float32)
        R = int(nodes.shape[0])
        # 1. bucket by owner (stable: keeps the original order inside a bucket)
        dest = owner_of(nodes, P)
        order = torch.argsort(dest, stable=True)
        send_counts = torch.bincount(dest, minlength=P)
        if P > 1:
            recv_counts = torch.empty_like(send_counts)
            dist.all_to_all_single(recv_counts, send_counts, group=self._group)
            sc, rc = send_counts.tolist(), recv_counts.tolist()
        else:
            sc = rc = [R]
        nodes_o, ts_o = nodes[order], ts[order]
        lo = sum(sc[:me])
        hi = lo + sc[me]            # [lo, hi) of the owner-sorted roots are this rank's own
        # 2. requests to the OTHER ranks only: [n, 2] int64 = (root id, root ts bits).  The
        #    exchange is started asynchronously ...
        sc_net, rc_net = list(sc), list(rc)
        sc_net[me] = rc_net[me] = 0
        pending = None
        if P > 1:
            away_nodes = torch.cat([nodes_o[:lo], nodes_o[hi:]])
            away_ts = torch.cat([ts_o[:lo], ts_o[hi:]])
            req = torch.stack([away_nodes, _pack_f32_pair(away_ts, torch.zeros_like(away_ts))],
                              dim=1)
            pending = _all_to_all_v(req, sc_net, rc_net, self._group, async_op=True)
        # 3. ... and overlaps with sampling this rank's own share on its shard
        own_pad = self._sample_padded(nodes_o[lo:hi], ts_o[lo:hi], layer, snapshot)
        if pending is not None:
            got, work = pending
            work.wait()
            got_ts, _ = _unpack_f32_pair(got[:, 1])
            # 4. sample what the other ranks asked for; replies: F*3 words per requested
            #    root, same splits as the requests (reversed), so no second count exchange
            remote_pad = self._sample_padded(got[:, 0].contiguous(), got_ts, layer, snapshot)
            rep_net = _all_to_all_v(remote_pad.reshape(-1, F * 3), rc_net, sc_net,
                                    self._group).reshape(R - (hi - lo), F, 3)
            rep = torch.cat([rep_net[:lo], own_pad, rep_net[lo:]])
        else:
            rep = own_pad
        # 5. back to the ORIGINAL root order, then keep the valid slots (root-major order)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(R, device=dev)
        rep = rep[inv]
        valid = rep[:, :, 0] >= 0
        rj = torch.nonzero(valid)                 # sorted by (root, slot)
        row = rj[:, 0].contiguous()
        sel = rep[valid]                          # [S, 3] in the same order
        S = int(sel.shape[0])
        b_ts, b_dt = _unpack_f32_pair(sel[:, 2]) if S else (ts.new_empty(0), ts.new_empty(0))
        all_nodes = torch.cat([nodes, sel[:, 0]])
        all_ts = torch.cat([ts, b_ts])
        col = torch.arange(R, R + S, device=dev)
        return LayerResult(all_nodes, all_ts, b_dt, sel[:, 1].contiguous(), row, col, R)

    def sample(self, nodes, ts) -> List[List[LayerResult]]:
        if not isinstance(nodes, torch.Tensor):
            nodes = torch.from_numpy(np.ascontiguousarray(nodes, dtype=np.int64))
        if not isinstance(ts, torch.Tensor):
            ts = torch.from_numpy(np.ascontiguousarray(ts, dtype=np.float32))
        results: List[List[LayerResult]] = []
        for layer in range(self._L):
            cur = []
            for s in range(self._S):
                if layer == 0:
                    n, t = nodes, ts
                else:
                    prev = results[-1][s]
                    n, t = prev.srcdata["ID"], prev.srcdata["ts"]
                cur.append(self.sample_layer(n, t, layer, s))
            results.append(cur)
        results.reverse()
        return results


# ---- the same exchange on the native entry points (device-resident, HIP kernels) ---------
def _backend_is_host_only(group) -> bool:
    return dist.get_backend(group) == "gloo"


def _exchange(out: torch.Tensor, send: torch.Tensor, out_splits, in_splits, group, async_op=False):
    """all_to_all_single(out, send) for HBM tensors.  Over RCCL ("nccl") the device buffers go
    straight in and async_op lets the caller overlap kernels with the transfer; a host-only
    backend (gloo: CPU tests, several ranks sharing one GPU) is staged through host memory."""
    if send.is_cuda and _backend_is_host_only(group):
        recv_cpu = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(recv_cpu, send.cpu(), output_split_sizes=out_splits,
                               input_split_sizes=in_splits, group=group)
        out.copy_(recv_cpu)
        return None
    return dist.all_to_all_single(out, send, output_split_sizes=out_splits,
                                  input_split_sizes=in_splits, group=group, async_op=async_op)


def _device_index(device):
    """Index of a cuda device for the C ABI; a bare "cuda" means torch's current device (NOT 0:
    rank k of a node works on device k)."""
    device = torch.device(device)
    return torch.cuda.current_device() if device.index is None else device.index


class NativeComm:
    """The library's own communicator over the ranks of a torch process group
    (include/gnnflow_hip.h gf_comm_* / gf_ipc_comm_*); torch.distributed only carries the
    bootstrap (control plane), never the data path.
      transport "rccl": RCCL — rank 0 draws the unique id, its 128 bytes are broadcast, every
                        rank joins; an all-to-all then costs a native call;
      transport "ipc":  hipIpc mailboxes + a process barrier in POSIX shared memory, for ranks
                        of one node that RCCL cannot serve — several ranks SHARING one GPU (the
                        one-GPU test box): host-synchronising, a test / single-box transport;
      transport "loopback": `NativeComm.loopback(P, device)` — P virtual ranks inside THIS
                        process, each driven by its own thread (world 8 on a one-GPU box)."""

    _counter = 0

    @classmethod
    def loopback(cls, world_size, device):
        """P communicators of one in-process group (include/gnnflow_hip.h
        gf_loopback_comm_create): rank r's calls must come from rank r's own thread."""
        import ctypes as C
        from . import _capi
        lib = _capi.load()
        device = torch.device(device)
        arr = (C.c_void_p * world_size)()
        _capi.check(lib.gf_loopback_comm_create(arr, world_size, _device_index(device)))
        out = []
        for r in range(world_size):
            c = cls.__new__(cls)
            c._lib, c.P, c.rank, c.transport = lib, world_size, r, "loopback"
            c.h = C.c_void_p(arr[r])
            out.append(c)
        return out

    def __init__(self, device, group=None, transport="rccl", mailbox_bytes=64 << 20):
        import ctypes as C
        import os
        from . import _capi
        self._lib = _capi.load()
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.transport = transport
        device = torch.device(device)
        self.h = C.c_void_p()
        if transport == "ipc":
            NativeComm._counter += 1
            # pid + counter + random token: an object left behind by a crashed run with a reused
            # pid is never picked up again
            name = ["/gnnflow_ipc_{}_{}_{}".format(os.getpid(), NativeComm._counter,
                                                   os.urandom(6).hex())]
            if self.P > 1:
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast_object_list(name, src=src, group=group)
            _capi.check(self._lib.gf_ipc_comm_create(C.byref(self.h), self.P, self.rank,
                                                     _device_index(device), int(mailbox_bytes),
                                                     name[0].encode()))
            mine = (C.c_uint8 * 64)()
            _capi.check(self._lib.gf_ipc_comm_handle(self.h, mine))
            handles = [None] * self.P
            if self.P > 1:
                dist.all_gather_object(handles, bytes(mine), group=group)
            else:
                handles = [bytes(mine)]
            blob = (C.c_uint8 * (64 * self.P))(*b"".join(handles))
            _capi.check(self._lib.gf_ipc_comm_open(self.h, blob))
            return
        idb = (C.c_uint8 * 128)()
        if self.rank == 0:
            _capi.check(self._lib.gf_comm_unique_id(idb))
        if self.P > 1:
            t = torch.tensor(list(idb), dtype=torch.uint8)
            on_device = not _backend_is_host_only(group)
            if on_device:
                t = t.to(device)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast(t, src=src, group=group)
            idb = (C.c_uint8 * 128)(*t.cpu().tolist())
        _capi.check(self._lib.gf_comm_create(C.byref(self.h), idb, self.P, self.rank,
                                             _device_index(device)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self._lib.gf_comm_destroy(self.h)
            self.h = None

    def info(self) -> dict:
        """What the transport itself reports (include/gnnflow_hip.h gf_comm_info): for RCCL
        ncclCommCount / ncclCommUserRank / ncclCommCuDevice."""
        import ctypes as C
        from . import _capi
        out = (C.c_int32 * 4)()
        _capi.check(self._lib.gf_comm_info(self.h, out))
        return {"nranks": int(out[0]), "rank": int(out[1]), "device": int(out[2]),
                "transport": {0: "rccl", 1: "ipc", 2: "loopback"}.get(int(out[3]), "?")}

    def abort(self):
        """Gives the communicator up without waiting for its peers (a hung collective)."""
        if getattr(self, "h", None) is not None and self.h.value:
            self._lib.gf_comm_abort(self.h)

    def time_all_to_all(self, bytes_per_peer: int, iters: int = 50, stream=None):
        """(device us, host us) per equal-split all-to-all of `bytes_per_peer` — collective."""
        import ctypes as C
        from . import _capi
        dev_us, host_us = C.c_double(0), C.c_double(0)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else None
        _capi.check(self._lib.gf_comm_time_all_to_all(self.h, int(bytes_per_peer), int(iters), st,
                                                      C.byref(dev_us), C.byref(host_us)))
        return dev_us.value, host_us.value

    def __del__(self):
        self.close()

    @staticmethod
    def choose(group=None):
        """Which transport carries the exchanges: "rccl" when the process group runs over RCCL
        (one GPU per rank), else None = staged through torch.distributed (gloo: CPU tests,
        several ranks sharing one card) unless GNNFLOW_PART_TRANSPORT says "ipc" (the library's
        hipIpc transport: the native chains with several ranks on one GPU) or "torch"."""
        import os
        mode = os.environ.get("GNNFLOW_PART_TRANSPORT", "auto").lower()
        if mode == "torch":
            return None
        if mode in ("ipc", "rccl"):
            return mode
        if mode == "native":
            return "rccl"
        if not dist.is_initialized() or not _backend_is_host_only(group):
            return "rccl"
        return None

    @staticmethod
    def usable(group=None) -> bool:
        return NativeComm.choose(group) is not None

    @staticmethod
    def create_agreed(device, group, kind, mailbox_bytes=64 << 20):
        """NativeComm(...) or None — the SAME answer on every rank: a rank whose creation
        failed (no RCCL to load, ncclCommInitRank error) tells the others through one
        all-reduce(min) over the bootstrap group, and then every rank drops its communicator
        and the exchanges go through torch.distributed everywhere.  (A group split between the
        two transports would hang in its first all-to-all.)"""
        import sys
        comm, err = None, None
        try:
            comm = NativeComm(device, group, kind, mailbox_bytes=mailbox_bytes)
        except Exception as e:     # noqa: BLE001
            err = e
        ok = comm is not None
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            host = _backend_is_host_only(group)
            t = torch.tensor([1 if ok else 0], dtype=torch.int32,
                             device="cpu" if host else torch.device(device))
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            ok = bool(int(t))
        if not ok:
            if comm is not None:
                comm.close()
            print("gnnflow_amd: no native communicator on every rank ({}); the exchanges go "
                  "through torch.distributed".format(
                      "{}: {}".format(type(err).__name__, err) if err else "another rank failed"),
                  file=sys.stderr)
            return None
        return comm


class _PartitionedPending:
    """PendingSample of a slotted partitioned sample: `wait()` also reads the sample's overflow
    flag and, if a slot overflowed on ANY rank (the flag travels in the slot headers, so every
    rank reads the same value for the same sample), samples the batch again through the
    variable-size exchange — on every rank, at the same point of its call sequence.
    A sample may be HELD (not issued yet) while it waits for the next batches to share its chain
    with (DevicePartitionedSampler chain_samples); `wait()` on a held sample issues the chain
    with the samples held so far."""

    def __init__(self, owner, lane, smp, pending, nodes, ts, stream, worker_enqueue=False):
        self._owner, self._lane, self._smp = owner, lane, smp
        self._nodes, self._ts, self._stream = nodes, ts, stream
        self._worker_enqueue = worker_enqueue
        self._result = None
        self._overflowed = False
        self._pending = None
        if pending is not None:
            self._attach(pending)

    def _attach(self, pending):
        self._pending = pending
        self._epoch = self._owner._slack_epoch     # the slot capacity this chain went out with
        # samples end in the order they were begun, possibly inside a younger sample's wait():
        # the flag is read when THIS sample's native sample_end has just returned
        pending._after_end = self._read_flag

    def _read_flag(self):
        own = self._owner
        flag = own._C.c_int(0)
        own._capi.check(own._lib.gf_sampler_part_overflowed(self._smp._h, own._C.byref(flag)))
        self._overflowed = bool(flag.value)
        self._pending._after_end = None      # no reference cycle left behind

    def wait(self):
        if self._result is None:
            own = self._owner
            if self._pending is None:
                own._issue_held()
            mfgs = self._pending.wait()
            own._waited += 1
            if self._overflowed:
                own._note_overflow(self._epoch)
                # the samples begun on this LANE after this one end first (their launches may
                # still be with the enqueue thread, and they use the lane's communicator; their
                # results are kept), then the batch is sampled again.  Every rank does this at the same point of its call sequence
                # and — the pipeline being the same on every rank — with the same samples in
                # flight on the lane, so the lane's communicator sees the same order of
                # collectives everywhere.
                for smp in self._lane.samplers:
                    q = smp._inflight
                    while q:
                        q[0].wait()
                mfgs = own._redo(self._lane, self._smp, self._nodes, self._ts,
                                 self._stream).wait()
            self._result = mfgs
            self._nodes = self._ts = None
            self._pending = self._lane = self._smp = self._owner = None
        return self._result


class _Lane:
    """One sampling lane of a DevicePartitionedSampler: a sampler over the rank's shard with its
    own native workspace and publish ring, the stream its chains run on, its communicator, and
    a ring of exchange workspaces (one per sample that can be in flight on the lane)."""

    def __init__(self, sampler, comm=None, clones=0):
        self.sampler = sampler
        # shared chains: sample j of a chain runs through samplers[j] (clones of `sampler`)
        self.samplers = [sampler] + [sampler.clone() for _ in range(clones)]
        self.stream = None          # lanes >= 1: their own stream, created on first use
        self.comm = comm
        self.comm_tried = comm is not None
        self.ws_ring = [None, None, None, None]
        self.ws_views = [None, None, None, None]
        self.ws_next = 0


class DevicePartitionedSampler:
    """PartitionedSampler whose per-rank work is native and chained on the device
    (include/gnnflow_hip.h "Chained form" / "Slotted form"): per layer `gf_sampler_part_plan_own`
    buckets the roots — their count is the previous layer's R + S and never leaves HBM — and
    samples this rank's own share while the request exchange is in flight, the received
    requests are served from this rank's shard, `gf_sampler_part_merge` builds the block.
    Returns `gnnflow_amd.MFGBlock`s in HBM exactly like `TemporalSampler.sample()` — for
    most-recent sampling bit-identical to one GPU holding the whole graph — so
    `Cache.fetch_feature` takes them unchanged.

    Host synchronisation with P ranks: NONE inside a sample (slotted form, the default): every
    peer has a slot of fixed capacity `slack` x the even share of the layer's worst-case root
    count in the request and reply buffers, so both exchanges of a layer are equal-split
    all-to-alls whose sizes are known in advance; plan -> exchange -> own share + serve ->
    exchange -> merge is enqueued without reading anything back, and the only wait is the
    final size read-back in `PendingSample.wait()`.  A slot that would overflow raises a flag
    that reaches every rank in the same exchange; that sample is then redone through the
    variable-size exchange (`slack=0` always takes it: all-to-all-v, one host synchronisation
    per layer for the split sizes).  With one rank the whole chain is a single native call
    that `sample_async` can hand to the enqueue thread like the plain sampler's.  A rank whose
    batch is empty still takes part in every collective (its peers would otherwise block).

    Throughput, not latency: a sample is a chain of ~13 dependent stream operations (two
    all-to-alls per layer among them), so consecutive batches are spread round-robin over
    `lanes` sampling lanes — each with its own stream, native workspace and communicator — and
    batch i+1's plan / exchange / serve overlaps batch i's (reference: the asynchronous
    per-partition requests of gnnflow/distributed/dist_sampler.py:188-220).  Every rank sends
    batch i to lane i mod lanes, one thread issues all chains in batch order, so every
    communicator sees the same order of collectives on every rank.

    sampler: a gnnflow_amd.TemporalSampler over THIS rank's shard (edges whose source it
    owns, see PartitionedGraph)."""

    def __init__(self, sampler, group=None, always_exchange=False, slack=None, slot_roots=None,
                 comm=None, overlap=None, lanes=None, pair=None, chain_samples=None,
                 narrow_ids=None, adapt_slack=None, edge_fill=None, reuse_roots=None):
        """always_exchange: take the multi-rank path — request / reply exchange, served
        requests, merge — even with one rank (where every message is empty).  For tests: it
        is the only way to run the RCCL branch on a one-GPU box.
        slack: capacity factor of the slotted exchange: a peer's slot holds `slack` x the even
        share of the layer's worst-case root count.  Default GNNFLOW_PART_SLACK, else
        1.2 + 0.1 x world size (1.4 at 2 ranks, 1.6 at 4, 2.0 at 8): a node that makes up a share
        h of a batch's roots sends them all to ONE owner, whose bucket is then 1 + h (P - 1) even
        shares — the REDDIT-shaped replay needs 1.15 / 1.31 / 1.64 at 2 / 4 / 8 ranks
        (scripts/slack_needed.py) — and the bytes on the wire are proportional to it.  A slot
        that overflows anyway costs a redo, never a wrong block.  0 = the variable-size exchange.
        slot_roots: the batch size (roots per sample() call) the slot capacities are derived
        from — the SAME number on every rank, because the slots of an equal-split exchange
        must have the same size everywhere.  None: agreed on by an all-reduce (max) of the
        ranks' first batches — one host synchronisation, once.  Larger batches later are still
        sampled correctly (they overflow into the variable-size exchange).
        comm: a NativeComm (the library's communicator; a list of them: one per lane) — the
        slotted chain is then ONE native call that the enqueue thread can issue
        (`worker_enqueue=True`), and world size / rank are the communicator's (a loopback
        communicator's ranks live in one process without torch.distributed).  None: created on
        first use when the process group runs over RCCL; the exchange goes through
        torch.distributed (staged through host memory over gloo) otherwise.
        overlap: run the exchanges on the communicator's own stream, the request exchange while
        the rank's own share is sampled (north_star's side-stream overlap).  Default
        GNNFLOW_PART_OVERLAP or OFF: on this runtime the two event hand-overs per exchange
        cost far more than the ~6 us own-share kernel they hide (one rank over RCCL, every
        message empty: 268 us per step with them, 116 us with everything in the sampling
        stream; profiles/r03_part_bench_one_gpu.jsonl).
        adapt_slack: raise the slot capacity by a quarter (up to twice the initial one) when
        two samples overflow within 64 — every rank sees the same samples overflow, so every
        rank takes the step at the same sample and the slots keep the same size everywhere.
        Default: on when `slack` is the default, off when it was given.
        lanes: sampling lanes (default GNNFLOW_PART_LANES, else 2 — 3 from 4 ranks on — with
        more than one rank / always_exchange, else 1).  A lane is used only by `sample_async(..., stream=...)` calls
        that name a stream (the pipelined loop); `sample()` always runs on lane 0.
        chain_samples: let up to this many (1..4) consecutive `sample_async(..., stream=...)`
        calls share ONE chain — its launches and its exchanges (include/gnnflow_hip.h
        gf_sampler_sample_partitioned_comm_group: 11 stream operations per m samples instead of
        per sample; the chain is bound by the host thread that issues it).  The samples are held
        until the chain is full or one of them is waited for (then the chain goes out with
        those held so far).  Default GNNFLOW_PART_CHAIN or 4; needs the library's communicator,
        one snapshot and layers of <= 32 768 roots (else single chains).
        pair: False = chain_samples 1 (the earlier name of the switch; GNNFLOW_PART_PAIR=0).
        narrow_ids: shared chains carry 12-byte reply slots {destination, edge id, edge time}
        instead of 24-byte ones — half the bytes of the reply exchange — which needs every node
        and edge id of every rank's shard in [0, 2^32 - 2].  It is part of the wire format: the
        same on every rank.  None (default; GNNFLOW_PART_NARROW=0 turns it off): decided once,
        before the first shared chain, from `graph.ids_fit_u32()` of every rank (one
        all-reduce; ranks that do not share a torch process group — the loopback transport —
        stay wide unless told).  Should a shard stop fitting later, its rank flags every sample
        of its chains as overflowed and all ranks redo them through the wide form.
        edge_fill: COMPACT replies of the shared chains: a reply slot travels as its rows' edge
        offsets + the sampled edges packed behind them (the reference ships back exactly the
        sampled edges, gnnflow/distributed/common.py:4-19), with room for this share of the
        slot's rows x fanout fixed records; a fuller slot flags the sample as overflowed (redo),
        and with adapt_slack two of those within 64 samples raise the share by half.  Part of
        the wire format: the same on every rank.  The share applies to the LAST layer; layer l
        gets edge_fill^(l / (L - 1)) — the first layer, whose roots are the batch itself, all of
        its records.  Default GNNFLOW_PART_EDGE_FILL, else 0.1 (on the REDDIT-shaped replay the
        fullest layer-1 slot holds 0.045 of its records, a layer-0 slot half of them:
        scripts/slack_needed.py);
        0 = the fixed records travel (always so for single chains and the variable-size form).
        reuse_roots: layer l + 1's first roots ARE layer l's roots with the same timestamps; with
        most-recent sampling and equal fanouts the shared chains neither request nor sample them
        again — the merge takes their edges from layer l's block (the reference requests every
        root of every layer, dist_sampler.py:174-186).  Part of the protocol: the same on every
        rank.  Default GNNFLOW_PART_REUSE_ROOTS or on."""
        import ctypes as C
        import os
        from . import _capi
        self._always_exchange = bool(always_exchange)
        self._C, self._capi = C, _capi
        self._lib = _capi.load()
        self._sampler = sampler
        self._group = group
        comms = list(comm) if isinstance(comm, (list, tuple)) else ([comm] if comm is not None else [])
        if comms:
            self._P, self._rank = comms[0].P, comms[0].rank
        else:
            self._P = dist.get_world_size(group) if dist.is_initialized() else 1
            self._rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._device = sampler._device
        self._fanouts = list(sampler._fanouts)
        self._L, self._S = sampler._num_layers, sampler._num_snapshots
        if slack is None:
            env = os.environ.get("GNNFLOW_PART_SLACK")
            slack = float(env) if env not in (None, "") else 1.2 + 0.1 * min(self._P, 16)
            if adapt_slack is None:
                adapt_slack = env in (None, "")
        self._slack = max(float(slack), 0.0)
        self._adapt_slack = bool(adapt_slack) and self._slack > 0
        self._slack_cap = 2.0 * self._slack     # (the ipc test transport's mailbox is sized for it)
        self._slack_epoch = 0                   # bumps so far
        self._waited = 0                        # samples completed (the same count on every rank)
        self._recent_overflows = []             # `_waited` of the overflows of this epoch
        self._slot_roots = int(slot_roots) if slot_roots else 0
        if overlap is None:
            overlap = os.environ.get("GNNFLOW_PART_OVERLAP", "0") != "0"
        self._overlap = bool(overlap)
        if lanes is None:
            # 2 lanes hide the chain of one rank over RCCL (measured); with 4 and more ranks the
            # exchanges are real transfers (the layer-1 replies of a chain of four are ~38 MB
            # per rank at P = 8), the chain is several times longer, and more of them must be
            # in flight: a third lane (free on one GPU; a fourth stretched the fetch kernels)
            lanes = int(os.environ.get("GNNFLOW_PART_LANES", "3" if self._P >= 4 else "2"))
        if (self._P == 1 and not self._always_exchange) or self._slack <= 0:
            # one rank without exchange: lanes were measured and do not pay (50-53 us per step
            # with 1 lane, 54 with 2, 48 with 3: profiles/README.md round 4); the variable-size
            # exchange synchronises the host per layer
            lanes = 1
        if comms and comms[0].transport in ("ipc", "loopback"):
            lanes = min(lanes, len(comms))    # host-synchronising transports: as many as given
        lanes = max(1, min(int(lanes), 4))
        if pair is None:
            pair = os.environ.get("GNNFLOW_PART_PAIR", "1") != "0"
        if chain_samples is None:
            chain_samples = int(os.environ.get("GNNFLOW_PART_CHAIN", "4"))
        chain = max(1, min(int(chain_samples), _capi.GF_PART_GROUP_MAX)) if pair else 1
        if not (self._slack > 0 and self._S == 1):
            chain = 1
        # one rank and nothing to exchange: the shared chain without its all-to-alls (the
        # native call takes no communicator)
        self._solo = self._P == 1 and not self._always_exchange
        if self._solo:
            self._adapt_slack = False      # no slots in use: only a forced flag "overflows"
        self.chain_samples = chain
        if narrow_ids is None and os.environ.get("GNNFLOW_PART_NARROW", "1") == "0":
            narrow_ids = False
        self._narrow = None if narrow_ids is None else bool(narrow_ids)   # None: not decided yet
        self._lanes = [_Lane(sampler, comms[0] if comms else None, chain - 1)]
        for k in range(1, lanes):
            self._lanes.append(_Lane(sampler.clone(), comms[k] if k < len(comms) else None,
                                     chain - 1))
        self.lanes = lanes
        if reuse_roots is None:
            reuse_roots = os.environ.get("GNNFLOW_PART_REUSE_ROOTS", "1") != "0"
        self._reuse_roots = bool(reuse_roots)
        if edge_fill is None:
            edge_fill = float(os.environ.get("GNNFLOW_PART_EDGE_FILL", "0.1"))
        self._edge_fill = 0.0
        # a chain of ONE sample may use the shared chains' machinery too (compact replies: the
        # reference ships back a partition's sampled edges, not fanout records per root,
        # gnnflow/distributed/dist_sampler.py:244-314) — rung "hash-simple" of bench.py's ladder
        groupable = self._slack > 0 and self._S == 1
        self._chain_of_one = (chain == 1 and groupable and
                              os.environ.get("GNNFLOW_PART_SINGLE_COMPACT", "1") != "0")
        self._set_edge_fill(max(0.0, min(float(edge_fill), 1.0))
                            if (chain > 1 or self._chain_of_one) else 0.0)
        self._chain_of_one = self._chain_of_one and self._edge_fill > 0.0
        self._held = []            # samples (of one lane) waiting for their chain to fill
        self.pairs = 0             # chains that carried more than one sample
        self.chained = 0           # samples that travelled in such chains
        self._rr = 0               # round-robin cursor over the lanes
        self._layouts = {}     # (R0, slack) -> ([layouts per layer], [workspace offsets], total)
        self.overflows = 0     # slotted samples that had to be redone

    def _set_edge_fill(self, fill: float):
        """The compact reply slots' edge capacity (in 1/1000: it travels in the flags word of
        the native calls, by value with every chain)."""
        self._edge_fill = int(round(float(fill) * 1000)) / 1000.0
        self._layouts = {}

    def comms(self):
        """The lanes' native communicators (those created so far)."""
        return [lane.comm for lane in self._lanes if lane.comm is not None]

    def wire_bytes_per_sample(self, R0=None) -> dict:
        """Bytes ONE rank sends to EACH peer for one sample() of `R0` roots (default: the agreed
        slot_roots) through the slotted exchange: the request slots (16 B per row, header row
        included) and the reply slots (fanout records of 12 / 24 B per row) of every layer and
        snapshot — fixed by the slot capacity, whatever the slots really hold."""
        R0 = int(R0 or max(self._slot_roots, 1))
        rec = 12 if self._narrow else 24
        fill = self._edge_fill
        req = rep = fixed = 0
        per_layer = []
        C = self._C
        out = (C.c_uint64 * 4)()
        for layer, F in enumerate(self._fanouts if self._slack > 0 else []):
            # the native layout itself (sampler.hip group_layout), not a restatement of it
            self._capi.check(self._lib.gf_sampler_part_group_slot(
                self._sampler._h, R0, layer, self._P, self._slack, self._slot_roots,
                (1 if self._narrow else 0) |
                (2 if (self._reuse_roots and (self.chain_samples > 1 or self._chain_of_one)) else 0),
                fill, out))
            stride = int(out[0])
            rq = stride * self._S * 16
            rp = self._S * int(out[1])
            fixed += stride * self._S * F * rec
            req += rq
            rep += rp
            per_layer.append({"request_bytes_to_each_peer": rq, "reply_bytes_to_each_peer": rp})
        return {"request_bytes_to_each_peer": req, "reply_bytes_to_each_peer": rep,
                "reply_bytes_to_each_peer_fixed_slots": fixed,
                "bytes_on_links_per_rank": (self._P - 1) * (req + rep),
                "reply_record_bytes": rec, "reply_edge_fill": fill, "slot_roots": R0,
                "slack": self._slack, "per_layer": per_layer}

    # the plain sampler's attributes the pipeline / cache helpers look at
    @property
    def _inflight(self):
        return self._sampler._inflight

    @property
    def _comm(self):
        return self._lanes[0].comm

    def _plan(self, R0: int, slack: float = 0.0):
        hit = self._layouts.get((R0, slack))
        if hit is None:
            C = self._C
            lays, offs, total = [], [], 0
            for layer in range(self._L):
                lay = self._capi.GfPartLayout()
                if slack > 0:
                    self._capi.check(self._lib.gf_sampler_part_layout_slotted(
                        self._sampler._h, R0, layer, self._P, slack, self._slot_roots,
                        C.byref(lay)))
                else:
                    self._capi.check(self._lib.gf_sampler_part_layout(
                        self._sampler._h, R0, layer, self._P, C.byref(lay)))
                lays.append(lay)
                row = []
                for _ in range(self._S):
                    row.append(total)
                    total += lay.total
                offs.append(row)
            hit = self._layouts[(R0, slack)] = (lays, offs, total)
        return hit

    def _workspace(self, lane, nbytes: int, stream):
        """grow-only scratch, one per sample that can be in flight on the lane at a time:
        (tensor, ring index)"""
        i = lane.ws_next
        lane.ws_next = (i + 1) % len(lane.ws_ring)
        ws = lane.ws_ring[i]
        if ws is None or ws.numel() < nbytes:
            with torch.cuda.stream(stream):
                ws = lane.ws_ring[i] = torch.empty(max(nbytes, 1), dtype=torch.uint8,
                                                   device=self._device)
            lane.ws_views[i] = None
        return ws, i

    def _slot_views(self, lane, ws, i, R0, lays, offs):
        """(requests out, inbox, served, replies in) int64 views of every (layer, snapshot) —
        the four buffers of the two equal-split exchanges — cut once per workspace and R0."""
        hit = lane.ws_views[i]
        if hit is None or hit[0] != R0:
            P, views = self._P, {}
            for layer in range(self._L):
                lay, F = lays[layer], self._fanouts[layer]
                n = P * lay.slot_stride
                for s in range(self._S):
                    off = offs[layer][s]

                    def cut(at, nbytes):
                        return ws[off + at: off + at + nbytes].view(torch.int64)
                    views[(layer, s)] = (cut(lay.requests, 16 * n), cut(lay.inbox, 16 * n),
                                         cut(lay.served, 24 * F * n), cut(lay.replies, 24 * F * n))
            hit = lane.ws_views[i] = (R0, views)
        return hit[1]

    def _pick_lane(self, stream):
        """(lane, stream) of the next sample: lane 0 on the caller's stream unless the caller
        named a stream and there are several lanes — then round-robin, lanes >= 1 on their own
        streams (the caller consumes the blocks after wait(), which has seen them complete)."""
        if stream is None:
            return self._lanes[0], torch.cuda.current_stream(self._device)
        if self.lanes == 1:
            return self._lanes[0], stream
        k = self._rr % self.lanes
        self._rr += 1
        lane = self._lanes[k]
        if k == 0:
            return lane, stream
        if lane.stream is None:
            import os
            prio = int(os.environ.get("GNNFLOW_PART_LANE_PRIORITY", "0"))
            if prio:
                lane.stream = torch.cuda.Stream(device=self._device, priority=prio)
            else:       # from the process-wide set (pipeline.side_stream: hardware queues)
                from .pipeline import side_stream
                lane.stream = side_stream(self._device, k)
        return lane, lane.stream

    def sample_async(self, nodes, ts, stream=None, worker_enqueue=False):
        """TemporalSampler.sample_async for the partitioned graph: returns a pending sample;
        `.wait()` gives the MFGs.  With more than one rank the launches and the collectives
        are enqueued by this call without any host synchronisation (slotted form)."""
        if (self.chain_samples > 1 or self._chain_of_one) and stream is not None:
            return self._sample_chained(nodes, ts, stream, worker_enqueue)
        lane, stream = self._pick_lane(stream)
        smp = lane.sampler
        if len(smp._inflight) >= smp._max_inflight:
            smp._inflight[0].wait()
        nodes, ts = smp._to_device(nodes, ts, stream)
        if self._P == 1 and not self._always_exchange:
            return self._sample_one_rank(lane, nodes, ts, stream, worker_enqueue)
        if self._slack > 0:
            return _PartitionedPending(
                self, lane, smp, self._sample_slotted(lane, smp, nodes, ts, stream, worker_enqueue),
                nodes, ts, stream)
        return self._sample_variable(lane, smp, nodes, ts, stream)

    # ---- shared chains: up to chain_samples consecutive samples in one chain ------------------
    def _sample_chained(self, nodes, ts, stream, worker_enqueue):
        held = self._held
        if not held:
            lane, st = self._pick_lane(stream)
        else:
            lane, st = held[0]._lane, held[0]._stream
        smp = lane.samplers[len(held)]
        nodes, ts = smp._to_device(nodes, ts, st)
        p = _PartitionedPending(self, lane, smp, None, nodes, ts, st, worker_enqueue)
        held.append(p)
        if len(held) == self.chain_samples:
            self._issue_held()
        return p

    def _issue_held(self):
        """The chain of the samples held so far goes out (it is full, or one of them is waited
        for before the next batch arrived)."""
        held, self._held = self._held, []
        if not held:
            return
        lane, stream = held[0]._lane, held[0]._stream
        worker_enqueue = held[-1]._worker_enqueue
        for p in held:
            smp = p._smp
            if len(smp._inflight) >= smp._max_inflight:
                smp._inflight[0].wait()
        solo = self._solo
        # (a lone sample of a lane that chains travels as a chain of one when compact replies are
        # on: the same wire format as its neighbours' — every rank decides alike, the samples held
        # are the same on all of them)
        if len(held) == 1 and not (self._edge_fill > 0.0 and not solo):
            p = held[0]
            p._attach(self._sample_one_rank(lane, p._nodes, p._ts, stream, worker_enqueue, p._smp)
                      if solo else
                      self._sample_slotted(lane, p._smp, p._nodes, p._ts, stream, worker_enqueue))
            return
        C, lib, check = self._C, self._lib, self._capi.check
        m = len(held)
        Rs = [int(p._nodes.shape[0]) for p in held]
        if not self._slot_roots:
            self._slot_roots = self._agree_on_slot_roots(max(Rs + [1]))
        comm = None if solo else self._ensure_comm(lane)
        ws_bytes = 0
        if self._narrow is None:
            self._narrow = self._agree_on_narrow_ids(comm)
        # flags word of the native calls: bit 0 narrow records, bits 8.. the edge fill in 1/1000
        narrow = (1 if self._narrow else 0) | (2 if self._reuse_roots else 0) | \
            ((int(round(self._edge_fill * 1000)) << 8) if comm is not None else 0)

        def group_bytes(roots):
            key = ("chain", narrow) + tuple(roots)
            n = self._layouts.get(key)
            if n is None:
                arr = (C.c_size_t * len(roots))(*roots)
                out = C.c_size_t(0)
                check(lib.gf_sampler_part_group_ws_bytes(lane.sampler._h, arr, len(roots), self._P,
                                                         self._slack, self._slot_roots, narrow,
                                                         C.byref(out)))
                n = self._layouts[key] = out.value
            return n
        force = 0
        if comm is not None or solo:
            # Whether samples share a chain is part of the PROTOCOL (their slots travel in one
            # exchange): every rank must decide alike, so the decision follows from the batch
            # size all ranks agreed on (slot_roots), not from this rank's own batches.
            sr = max(self._slot_roots, 1)
            if group_bytes([sr]) > 0:
                # A batch too large for the shared chain's kernels (more roots than slot_roots,
                # so that a layer exceeds 32 768 roots) cannot change the protocol on its own:
                # an EMPTY stand-in travels in its place with the sample's overflow flag forced,
                # every rank sees the flag and the real batch is sampled in the redo.
                for j in range(m):
                    if group_bytes([max(Rs[j], 1)]) == 0:
                        force |= 1 << j
                        Rs[j] = 0
                ws_bytes = group_bytes([max(R, 1) for R in Rs])
                assert ws_bytes, "a chain of fitting samples has a workspace"
                if (narrow & 1) and not self._sampler._graph.ids_fit_u32():
                    # this rank's shard outgrew the 12-byte slots: the wire format cannot change
                    # on one rank's say-so, so every sample of the chain is flagged and redone
                    force = (1 << m) - 1
        if not ws_bytes:      # no communicator / not chainable: single chains
            for p in held:
                p._attach(self._sample_one_rank(lane, p._nodes, p._ts, stream, worker_enqueue,
                                                p._smp) if solo else
                          self._sample_slotted(lane, p._smp, p._nodes, p._ts, stream,
                                               worker_enqueue))
            return
        outs = [self._output(p._smp, R, stream) for p, R in zip(held, Rs)]
        ws, _ = self._workspace(lane, ws_bytes, stream)
        call = lib.gf_sampler_sample_partitioned_comm_group_async \
            if (worker_enqueue and (solo or comm.transport != "loopback")) \
            else lib.gf_sampler_sample_partitioned_comm_group
        desc = (self._capi.GfGroupSample * m)()
        for j, (p, R, (slab, out_ptr, nbytes)) in enumerate(zip(held, Rs, outs)):
            d = desc[j]
            d.sampler = p._smp._h.value
            d.d_roots = p._nodes.data_ptr() if R else None
            d.d_root_ts = p._ts.data_ptr() if R else None
            d.num_roots, d.d_out, d.out_bytes = R, out_ptr, nbytes
        check(call(None if solo else comm.h, desc, m, ws.data_ptr(), ws_bytes, self._slack,
                   self._slot_roots, force, narrow, outs[0][0][6]))
        for p, R, (slab, _, _) in zip(held, Rs, outs):
            p._attach(self._pend(p._smp, slab, (p._nodes, p._ts, ws), R))
        self.pairs += 1
        self.chained += m

    def _output(self, smp, R, stream):
        C = self._C
        nbytes = smp._bytes_cache.get(max(R, 1))
        if nbytes is None:
            n = C.c_size_t(0)
            self._capi.check(self._lib.gf_sampler_output_bytes(smp._h, max(R, 1), C.byref(n)))
            nbytes = smp._bytes_cache[max(R, 1)] = n.value
        slab, off = smp._output_buffer(nbytes, stream)
        return slab, slab[5] + off, nbytes

    def _pend(self, smp, slab, keep, R):
        # R = 0 still yields real (empty) blocks whose sizes come from the device
        pending = _PendingSample(smp, slab, keep, max(R, 1), None)
        smp._inflight.append(pending)
        return pending

    def _redo(self, lane, smp, nodes, ts, stream):
        """An overflowed sample again, through the form without slots."""
        if self._solo:
            return self._sample_one_rank(lane, nodes, ts, stream, False, smp)
        return self._sample_variable(lane, smp, nodes, ts, stream)

    def _sample_one_rank(self, lane, nodes, ts, stream, worker_enqueue, smp=None):
        lib, smp = self._lib, (smp or lane.sampler)
        R = int(nodes.shape[0])
        _, _, ws_bytes = self._plan(max(R, 1))
        slab, out_ptr, nbytes = self._output(smp, R, stream)
        ws, _ = self._workspace(lane, ws_bytes, stream)
        call = lib.gf_sampler_sample_partitioned_async if worker_enqueue \
            else lib.gf_sampler_sample_partitioned
        self._capi.check(call(smp._h, nodes.data_ptr() if R else None,
                              ts.data_ptr() if R else None, R, out_ptr, nbytes, ws.data_ptr(),
                              ws_bytes, slab[6]))
        return self._pend(smp, slab, (nodes, ts, ws), R)

    def _ensure_comm(self, lane):
        """The lane's communicator, created on first use (collective: every rank creates its
        lanes' communicators in the same order) — or None: exchanges through torch.distributed."""
        if not lane.comm_tried:
            # all lanes' communicators at once, before the first chain is issued: creating one
            # is a collective with allocations and synchronisations of its own, which must not
            # meet another lane's exchanges in flight
            for other in self._lanes:
                if other is not lane and not other.comm_tried:
                    self._create_comm(other)
            self._create_comm(lane)
        return lane.comm

    def _create_comm(self, lane):
        if not lane.comm_tried:
            lane.comm_tried = True
            kind = NativeComm.choose(self._group)
            if kind is not None:
                # ipc: the mailbox holds the largest message — a layer's reply slots for a batch
                # of slot_roots roots (same on every rank)
                P = self._P
                ref = self._plan(max(self._slot_roots, 1), self._slack)[0]
                # (and, for an overflowed sample's redo through the variable-size exchange,
                # never less than 16 MiB)
                m = self.chain_samples          # a shared exchange carries m slots per peer
                box = max([16 << 20] + [m * P * lay.slot_stride * (24 * f + 16) + 512 * P
                                        for lay, f in zip(ref, self._fanouts)])
                lane.comm = NativeComm.create_agreed(self._device, self._group, kind,
                                                     mailbox_bytes=2 * box)
        return lane.comm

    def _sample_slotted(self, lane, smp, nodes, ts, stream, worker_enqueue=False):
        """The whole sample enqueued on `stream` without reading anything back: per layer
        plan -> equal-split exchange of the request slots (asynchronous, overlapped with the
        own share) -> serve -> equal-split exchange of the reply slots -> merge."""
        lib, check = self._lib, self._capi.check
        P, me, group = self._P, self._rank, self._group
        R = int(nodes.shape[0])
        R0 = max(R, 1)
        if not self._slot_roots:
            self._slot_roots = self._agree_on_slot_roots(R0)
        lays, offs, ws_bytes = self._plan(R0, self._slack)
        comm = self._ensure_comm(lane)
        slab, out_ptr, nbytes = self._output(smp, R, stream)
        ws, wi = self._workspace(lane, ws_bytes, stream)
        if comm is not None:
            # the library's own communicator: the chain is one native call (a loopback
            # communicator's ranks are threads: they cannot share the one enqueue thread)
            call = lib.gf_sampler_sample_partitioned_comm_async \
                if (worker_enqueue and comm.transport != "loopback") \
                else lib.gf_sampler_sample_partitioned_comm
            check(call(smp._h, comm.h, nodes.data_ptr() if R else None,
                       ts.data_ptr() if R else None, R, out_ptr, nbytes, ws.data_ptr(), ws_bytes,
                       self._slack, self._slot_roots, 1 if self._overlap else 0, slab[6]))
            return self._pend(smp, slab, (nodes, ts, ws), R)
        views = self._slot_views(lane, ws, wi, R0, lays, offs)
        base = ws.data_ptr()
        with torch.cuda.stream(stream):
            check(lib.gf_sampler_part_begin_slotted(
                smp._h, nodes.data_ptr() if R else None, ts.data_ptr() if R else None, R,
                out_ptr, nbytes, P, me, self._slack, self._slot_roots, slab[6]))
            try:
                for layer in range(self._L):
                    total = lays[layer].total
                    for s in range(self._S):
                        req, inbox, served, rep = views[(layer, s)]
                        at = base + offs[layer][s]
                        check(lib.gf_sampler_part_plan_own(smp._h, layer, s, at, total, 1))
                        work = _exchange(inbox, req, None, None, group, async_op=True)
                        check(lib.gf_sampler_part_plan_own(smp._h, layer, s, at, total, 2))
                        if work is not None:
                            work.wait()          # orders the stream, not the host
                        check(lib.gf_sampler_part_serve(smp._h, layer, s, at, total))
                        _exchange(rep, served, None, None, group)
                        check(lib.gf_sampler_part_merge(smp._h, layer, s, at, total))
                check(lib.gf_sampler_part_commit(smp._h))
            except Exception:
                lib.gf_sampler_part_abort(smp._h)
                raise
        return self._pend(smp, slab, (nodes, ts, ws), R)

    def _note_overflow(self, epoch):
        """A slotted sample has to be redone.  With adapt_slack: two of them within 64 samples
        (chains that went out before the last step do not count) raise the capacity for the
        chains issued from now on — deterministically, so on every rank at the same sample."""
        self.overflows += 1
        if not self._adapt_slack or epoch != self._slack_epoch or \
                (self._slack >= self._slack_cap and not 0.0 < self._edge_fill < 1.0):
            return
        recent = [w for w in self._recent_overflows if self._waited - w < 64] + [self._waited]
        self._recent_overflows = recent
        if len(recent) < 2:
            return
        self._slack = min(self._slack * 1.25, self._slack_cap)
        if 0.0 < self._edge_fill < 1.0:     # a compact slot may have run out of edge room instead
            self._set_edge_fill(min(self._edge_fill * 1.5, 1.0))
        self._slack_epoch += 1
        self._recent_overflows = []
        self._layouts = {}
        for lane in self._lanes:
            lane.ws_views = [None] * len(lane.ws_views)

    def _agree_on_narrow_ids(self, comm) -> bool:
        """12-byte reply slots only if EVERY rank's shard fits them (one all-reduce, once)."""
        mine = bool(self._sampler._graph.ids_fit_u32())
        if self._P == 1:
            return mine
        if not dist.is_initialized() or (comm is not None and comm.transport == "loopback"):
            return False          # no way to ask the other ranks: stay wide unless told
        host = _backend_is_host_only(self._group)
        t = torch.tensor([1 if mine else 0], dtype=torch.int64,
                         device="cpu" if host else self._device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self._group)
        return bool(int(t.item()))

    def _agree_on_slot_roots(self, R0: int) -> int:
        if self._P == 1:
            return R0
        if not dist.is_initialized() or (self._comm is not None
                                         and self._comm.transport == "loopback"):
            raise ValueError("DevicePartitionedSampler: pass slot_roots (the batch size every rank "
                             "agrees on) when the ranks do not share a torch process group")
        host = _backend_is_host_only(self._group)
        t = torch.tensor([R0], dtype=torch.int64, device="cpu" if host else self._device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._group)
        return int(t.item())

    # ---- exchanges of the variable-size form: the lane's communicator when it has one, else
    # torch.distributed ---------------------------------------------------------------------
    def _xchg_equal(self, comm, out, send):
        """equal split: send[p] -> rank p's out[me] (rows of equal size)"""
        if comm is None:
            return _exchange(out, send, None, None, self._group)
        st = torch.cuda.current_stream(self._device).cuda_stream
        per_peer = send.numel() * send.element_size() // self._P
        self._capi.check(self._lib.gf_comm_all_to_all(comm.h, send.data_ptr(), out.data_ptr(),
                                                      per_peer, st))
        return None

    def _xchg_rows(self, comm, out, send, out_rows, in_rows, async_op=False):
        """variable split: in_rows[p] rows of `send` to rank p, out_rows[p] rows from it"""
        if comm is None:
            return _exchange(out, send, out_rows, in_rows, self._group, async_op=async_op)
        C, P = self._C, self._P
        row = send.element_size() * (send.shape[1] if send.dim() > 1 else 1)
        arr = C.c_size_t * P
        sb, so, rb, ro = [], [], [], []
        a = b = 0
        for p in range(P):
            sb.append(in_rows[p] * row); so.append(a); a += in_rows[p] * row
            rb.append(out_rows[p] * row); ro.append(b); b += out_rows[p] * row
        st = torch.cuda.current_stream(self._device).cuda_stream
        # every rank makes every call, whatever its own sizes are (a transport may synchronise
        # the ranks inside it); an empty side still needs a valid pointer
        self._capi.check(self._lib.gf_comm_all_to_all_v(
            comm.h, send.data_ptr() if a else out.data_ptr(), arr(*sb), arr(*so),
            out.data_ptr() if b else send.data_ptr(), arr(*rb), arr(*ro), st))
        return None

    def _sample_variable(self, lane, smp, nodes, ts, stream):
        """Variable-size exchange: all-to-all-v of exactly the rows there are; the per-owner
        counts are read back once per layer (they are the split sizes)."""
        lib, check = self._lib, self._capi.check
        R = int(nodes.shape[0])
        lays, offs, ws_bytes = self._plan(max(R, 1))
        slab, out_ptr, nbytes = self._output(smp, R, stream)
        ws, _ = self._workspace(lane, ws_bytes, stream)
        sptr = slab[6]
        with torch.cuda.stream(stream):
            check(lib.gf_sampler_part_begin(
                smp._h, nodes.data_ptr() if R else None, ts.data_ptr() if R else None, R,
                out_ptr, nbytes, self._P, self._rank, sptr))
            try:
                for layer in range(self._L):
                    for s in range(self._S):
                        self._exchange_layer(lane, smp, layer, s, lays[layer], ws, offs[layer][s],
                                             sptr)
                check(lib.gf_sampler_part_commit(smp._h))
            except Exception:
                lib.gf_sampler_part_abort(smp._h)
                raise
        return self._pend(smp, slab, (nodes, ts, ws), R)

    def _exchange_layer(self, lane, smp, layer, snapshot, lay, ws, off, sptr):
        """One (layer, snapshot) with P > 1 ranks, on the current stream."""
        lib, check = self._lib, self._capi.check
        dev, P, me, F = self._device, self._P, self._rank, self._fanouts[layer]
        comm = lane.comm
        base = ws.data_ptr() + off
        # 1. bucket by owner (the layer's root count is still on the device)
        check(lib.gf_sampler_part_plan_own(smp._h, layer, snapshot, base, lay.total, 1))
        counts = ws[off + lay.counts: off + lay.counts + 8 * P].view(torch.int64)
        recv_counts = torch.empty_like(counts)
        self._xchg_equal(comm, recv_counts, counts)
        sc, rc = counts.tolist(), recv_counts.tolist()      # the layer's one host sync
        R = sum(sc)
        n_net = R - sc[me]
        sc[me] = rc[me] = 0
        req = ws[off + lay.requests: off + lay.requests + 16 * R].view(torch.int64).view(R, 2)
        rep = ws[off + lay.replies: off + lay.replies + 24 * F * R].view(torch.int64).view(R, F * 3)
        # 2. requests to the other ranks ...
        n_got = sum(rc)
        got = torch.empty((max(n_got, 1), 2), dtype=torch.int64, device=dev)[:n_got]
        work = self._xchg_rows(comm, got, req[:n_net], rc, sc, async_op=True)
        # ... overlapped with this rank's own share on its shard
        check(lib.gf_sampler_part_plan_own(smp._h, layer, snapshot, base, lay.total, 2))
        if work is not None:
            work.wait()
        # 3. the received requests, served from this rank's shard; replies land in the prefix
        #    of `rep`
        served = torch.empty((max(n_got, 1), F * 3), dtype=torch.int64, device=dev)[:n_got]
        check(lib.gf_sampler_sample_layer_padded(
            smp._h, got.data_ptr() if n_got else None, n_got, layer, snapshot,
            served.data_ptr() if n_got else None, sptr))
        self._xchg_rows(comm, rep[:n_net], served, sc, rc)
        # 4. replies -> block in the original root order; sizes stay on the device
        check(lib.gf_sampler_part_merge(smp._h, layer, snapshot, base, lay.total))

    def sample(self, nodes, ts):
        return self.sample_async(nodes, ts).wait()

    def sample_layer(self, nodes, ts, layer: int, snapshot: int):
        """One (layer, snapshot) over the partitioned graph — the reference's per-layer call
        (gnnflow/distributed/dist_sampler.py:159-242 `sample_layer_global`: bucket by
        partition, sample remotely, merge).  A collective: every rank calls it for the same
        (layer, snapshot), with no roots if it has none.  Returns the block
        `TemporalSampler.sample_layer` would return on one GPU holding the whole graph (roots
        in the caller's order).  Synchronous: the per-owner counts and the block sizes are read
        back, as the reference's RPC round trip is."""
        C, lib, check = self._C, self._lib, self._capi.check
        smp, dev, P, me = self._sampler, self._device, self._P, self._rank
        comm = self._lanes[0].comm
        layer, snapshot = int(layer), int(snapshot)
        if not 0 <= layer < self._L or not 0 <= snapshot < self._S:
            raise ValueError("sample_layer: layer / snapshot out of range")
        F = self._fanouts[layer]
        nodes, ts = smp._to_device(nodes, ts)
        R = int(nodes.shape[0])
        if P == 1 and not self._always_exchange:
            return smp.sample_layer(nodes, ts, layer, snapshot)
        with torch.cuda.device(dev):
            st = smp._stream()
            n = C.c_size_t(0)
            check(lib.gf_partition_scratch_bytes(max(R, 1), P, C.byref(n)))
            scratch = torch.empty(max(n.value, 16), dtype=torch.uint8, device=dev)
            req = torch.empty((max(R, 1), 2), dtype=torch.int64, device=dev)
            pos = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
            counts = torch.zeros(P, dtype=torch.int64, device=dev)
            if R:
                check(lib.gf_partition_plan(nodes.data_ptr(), ts.data_ptr(), R, P, me,
                                            req.data_ptr(), pos.data_ptr(), counts.data_ptr(),
                                            scratch.data_ptr(), scratch.numel(), dev.index, st))
            recv_counts = torch.empty_like(counts)
            self._xchg_equal(comm, recv_counts, counts)
            sc, rc = counts.tolist(), recv_counts.tolist()
            n_net = R - sc[me]
            sc[me] = rc[me] = 0
            n_got = sum(rc)
            got = torch.empty((max(n_got, 1), 2), dtype=torch.int64, device=dev)[:n_got]
            work = self._xchg_rows(comm, got, req[:n_net], rc, sc, async_op=True)
            rep = torch.empty((max(R, 1), F * 3), dtype=torch.int64, device=dev)
            if R - n_net:       # own share: the suffix of the request rows
                check(lib.gf_sampler_sample_layer_padded(
                    smp._h, req[n_net:].data_ptr(), R - n_net, layer, snapshot,
                    rep[n_net:].data_ptr(), st))
            if work is not None:
                work.wait()
            served = torch.empty((max(n_got, 1), F * 3), dtype=torch.int64, device=dev)[:n_got]
            if n_got:
                check(lib.gf_sampler_sample_layer_padded(smp._h, got.data_ptr(), n_got, layer,
                                                         snapshot, served.data_ptr(), st))
            self._xchg_rows(comm, rep[:n_net], served, sc, rc)
            if R == 0:
                return smp._empty_block()
            nb = C.c_size_t(0)
            check(lib.gf_sampler_layer_output_bytes(smp._h, R, layer, C.byref(nb)))
            buf = torch.empty(nb.value, dtype=torch.uint8, device=dev)
            gb = self._capi.GfBlock()
            check(lib.gf_sampler_merge_padded(smp._h, nodes.data_ptr(), ts.data_ptr(), R, layer,
                                              rep.data_ptr(), pos.data_ptr(), buf.data_ptr(),
                                              nb.value, C.byref(gb), st))
            return smp._block(buf, gb)


# ---- feature tables sharded by owner (SURVEY.md 8(e) "Features") --------------------------
class FeatureShards:
    """One kind's feature table (node or edge features) sharded over the ranks: rank
    owner(key) holds the row, key = the node id for node features and the edge's SOURCE node
    for edge features — where the reference's KVStore keeps them
    (gnnflow/distributed/kvstore.py:70-126; `pull(mode='edge', nid=...)`, :300-308).

    `pull(ids, keys)` is the reference's `KVStoreClient.pull` as one all-to-all-v of ids and
    one of rows over the process group (RCCL on HBM tensors; gloo on CPU tensors or staged):
    a COLLECTIVE — every rank calls it the same number of times, with zero ids if it needs
    nothing.  Rows come back in the order of `ids`.
    """

    def __init__(self, num_ids: int, local_ids: torch.Tensor, local_rows: torch.Tensor,
                 group=None, rank=None, world_size=None):
        """rank / world_size: only for ranks that do not share a torch process group (the
        virtual ranks of a loopback communicator); else the group's."""
        self.num_ids = int(num_ids)
        self.device = local_rows.device
        self.dim = int(local_rows.shape[1])
        self.group = group
        if world_size is not None:
            self.P, self.rank = int(world_size), int(rank)
        else:
            self.P = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_ids = local_ids.to(self.device, torch.int64)
        self.rows = local_rows.to(torch.float32).contiguous()
        # global id -> local row (dense: 4 B per id of the GLOBAL id space on every rank)
        self.index = torch.full((self.num_ids,), -1, dtype=torch.int32, device=self.device)
        self.index[self.local_ids] = torch.arange(self.local_ids.shape[0], dtype=torch.int32,
                                                  device=self.device)
        self.bytes_sent = 0       # ids out + rows back, for the traffic figures in DESIGN.md
        self.rows_pulled = 0
        # tests: run the collective protocol even with one rank (every row then travels
        # through the all-to-all-v to this rank itself)
        self.always_exchange = False

    @classmethod
    def from_full(cls, feats, keys, rank: int, world_size: int, device, group=None):
        """Keeps the rows this rank owns out of a full table (tests, benches, loaders that
        read the whole file): feats [num_ids, dim], keys [num_ids] = the node id that decides
        each row's owner (nodes: arange(num_ids); edges: the edge's source node)."""
        feats = torch.as_tensor(feats)
        keys_np = np.asarray(keys, dtype=np.int64)
        own = np.nonzero(owner_of_np(keys_np, world_size) == rank)[0]
        ids = torch.from_numpy(own)
        explicit = not dist.is_initialized() or dist.get_world_size(group) != world_size
        return cls(feats.shape[0], ids.to(device), feats[ids].to(device, torch.float32), group,
                   rank=rank if explicit else None, world_size=world_size if explicit else None)

    def pull(self, ids: torch.Tensor, keys: torch.Tensor) -> torch.Tensor:
        dev, P = self.device, self.P
        ids = ids.to(dev, torch.int64)
        n = int(ids.shape[0])
        if P == 1 and not self.always_exchange:
            local = self.index[ids].long()
            if n and bool((local < 0).any()):
                raise KeyError("FeatureShards.pull: id without a row on its owner")
            return self.rows[local]
        owners = owner_of(keys.to(dev, torch.int64), P)
        order = torch.argsort(owners, stable=True)
        counts = torch.bincount(owners, minlength=P)
        recv_counts = torch.empty_like(counts)
        _exchange(recv_counts, counts, None, None, self.group)
        sc, rc = counts.tolist(), recv_counts.tolist()
        got = torch.empty(sum(rc), dtype=torch.int64, device=dev)
        _exchange(got, ids[order].contiguous(), rc, sc, self.group)
        local = self.index[got].long()
        if got.shape[0] and bool((local < 0).any()):
            raise KeyError("FeatureShards.pull: asked for an id this rank does not own")
        served = self.rows[local]
        back = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
        _exchange(back, served, sc, rc, self.group)
        out = torch.empty_like(back)
        out[order] = back
        me = self.rank
        self.bytes_sent += 8 * (n - sc[me]) + 4 * self.dim * (sum(rc) - rc[me])
        self.rows_pulled += n - sc[me]
        return out


class ShardedFeatures:
    """What `Cache(distributed=True, kvstore_client=...)` takes: the node and / or edge
    feature shards of this rank, and the transport of their pulls."""

    def __init__(self, node: Optional[FeatureShards] = None, edge: Optional[FeatureShards] = None,
                 group=None, comm: Optional["NativeComm"] = None):
        self.node, self.edge = node, edge
        self.group = group
        any_shard = node if node is not None else edge
        self.P = any_shard.P if any_shard is not None else 1
        self.rank = any_shard.rank if any_shard is not None else 0
        self.device = any_shard.device if any_shard is not None else torch.device("cpu")
        self._comm = comm
        self._comm_tried = comm is not None
        self.always_exchange = False     # tests: run the exchanges with one rank too
        self.rows_pulled = 0             # rows that came from other ranks
        self.bytes_sent = 0              # ids out + rows back
        self.host_syncs = 0              # count read-backs (one per fetch round)

    # ---- the exchanges of a native pull (gnnflow_amd/cache/cache.py _fetch_distributed) -----
    def comm(self):
        """The library's own RCCL communicator (None: go through torch.distributed — gloo in
        the CPU tests and when several ranks share one card)."""
        if not self._comm_tried:
            self._comm_tried = True
            kind = NativeComm.choose(self.group)
            if self.device.type == "cuda" and dist.is_initialized() and kind is not None:
                import os
                mb = int(os.environ.get("GNNFLOW_IPC_MAILBOX_MB", "256"))
                self._comm = NativeComm(self.device, self.group, kind, mailbox_bytes=mb << 20)
        return self._comm

    def exchange_counts(self, counts: torch.Tensor) -> torch.Tensor:
        """counts [nctx, P] (int32, device): rows this rank sends to each owner per context ->
        [nctx, P]: rows each rank asks THIS rank for.  Equal split, sizes known: no sync."""
        nctx, P = counts.shape
        if P == 1 and not self.always_exchange:
            return counts.clone()
        send = counts.t().contiguous()              # owner-major: what goes to rank p
        recv = torch.empty_like(send)
        comm = self.comm()
        if comm is not None:
            from . import _capi
            st = torch.cuda.current_stream(self.device).cuda_stream
            _capi.check(comm._lib.gf_comm_all_to_all(comm.h, send.data_ptr(), recv.data_ptr(),
                                                     4 * nctx, st))
        else:
            _exchange(recv, send, None, None, self.group)
        return recv.t().contiguous()

    def exchange_segments(self, send: List[torch.Tensor], sc: List[List[int]],
                          recv: List[torch.Tensor], rc: List[List[int]]):
        """One variable-size exchange per context, all in ONE RCCL group when the library's
        communicator carries them: context k sends sc[k][p] rows of send[k] (owner-major) to
        rank p and receives rc[k][p] rows into recv[k] (rank-major)."""
        if self.P == 1 and not self.always_exchange:
            for a, b in zip(send, recv):
                b.copy_(a)
            return
        comm = self.comm()
        if comm is not None:
            import ctypes as C
            from . import _capi
            P = self.P
            st = torch.cuda.current_stream(self.device).cuda_stream
            for k in range(len(send)):
                row = send[k].element_size() * (send[k].shape[1] if send[k].dim() > 1 else 1)
                arr = C.c_size_t * P
                sb, so, rb, ro = [], [], [], []
                a = b = 0
                for p in range(P):
                    sb.append(sc[k][p] * row); so.append(a); a += sc[k][p] * row
                    rb.append(rc[k][p] * row); ro.append(b); b += rc[k][p] * row
                # every rank makes every call (the ipc transport synchronises the ranks inside)
                _capi.check(comm._lib.gf_comm_all_to_all_v(
                    comm.h, send[k].data_ptr() if a else recv[k].data_ptr(), arr(*sb), arr(*so),
                    recv[k].data_ptr() if b else send[k].data_ptr(), arr(*rb), arr(*ro), st))
            return
        for k in range(len(send)):
            _exchange(recv[k], send[k], rc[k], sc[k], self.group)
