"""GPU: native hash-partitioned sampling (the chained gf_sampler_part_* calls and
gf_sampler_sample_layer_padded through gnnflow_amd.dist.DevicePartitionedSampler; the
one-layer gf_partition_plan directly).
  * one rank: the plan / padded / merge chain alone must reproduce TemporalSampler.sample()
    bit for bit (all roots are the rank's own);
  * the bucketing kernel against numpy for several world sizes and ranks (stable order,
    "[other owners ascending | own]" layout, counts);
  * two processes sharing the one GPU, exchanging through gloo (staged via host memory):
    every rank's MFGs equal the CPU oracle over the whole graph."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(seed=5, N=400, E=12000):
    from tests import synth
    return synth.powerlaw_graph(N, E, seed=seed, tie_levels=500)


@pytest.mark.parametrize("P,rank", [(1, 0), (2, 0), (2, 1), (3, 1), (8, 5), (64, 63)])
@pytest.mark.parametrize("R", [1, 255, 256, 257, 5000, 70000])
def test_partition_plan_matches_numpy(P, rank, R):
    import torch
    from gnnflow_amd import _capi
    from gnnflow_amd.dist import owner_of_np
    lib = _capi.load()
    rng = np.random.RandomState(R + P)
    nodes = rng.randint(0, 1 << 40, R).astype(np.int64)
    ts = rng.rand(R).astype(np.float32)
    dev = torch.device("cuda", 0)
    dn, dt = torch.from_numpy(nodes).to(dev), torch.from_numpy(ts).to(dev)
    need = C.c_size_t(0)
    _capi.check(lib.gf_partition_scratch_bytes(R, P, C.byref(need)))
    scratch = torch.empty(need.value, dtype=torch.uint8, device=dev)
    req = torch.empty((R, 2), dtype=torch.int64, device=dev)
    pos = torch.empty(R, dtype=torch.int32, device=dev)
    counts = torch.empty(P, dtype=torch.int64, device=dev)
    _capi.check(lib.gf_partition_plan(dn.data_ptr(), dt.data_ptr(), R, P, rank, req.data_ptr(),
                                      pos.data_ptr(), counts.data_ptr(), scratch.data_ptr(),
                                      need.value, 0, None))
    owner = owner_of_np(nodes, P)
    key = np.where(owner == rank, P, owner)          # own share goes last
    order = np.argsort(key, kind="stable")
    want_pos = np.empty(R, np.int64)
    want_pos[order] = np.arange(R)
    assert np.array_equal(pos.cpu().numpy().astype(np.int64), want_pos)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(owner, minlength=P))
    got = req.cpu().numpy()
    assert np.array_equal(got[:, 0], nodes[order])
    assert np.array_equal(got[:, 1].astype(np.uint64).astype(np.uint32).view(np.float32), ts[order])


@pytest.mark.parametrize("strategy,snapshots,window,prop_time",
                         [("recent", 1, 0.0, False), ("recent", 2, 60.0, True)])
def test_single_rank_chain_equals_plain_sampler(strategy, snapshots, window, prop_time):
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler
    from tests import synth
    src, dst, ts, eid = _graph()
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert")
    g.add_edges(src, dst, ts, eid, add_reverse=True)
    kw = dict(fanouts=[7, 5], sample_strategy=strategy, num_snapshots=snapshots,
              snapshot_time_window=window, prop_time=prop_time)
    plain = TemporalSampler(g, **kw)
    part = DevicePartitionedSampler(TemporalSampler(g, **kw))
    for it, R in enumerate([1, 97, 600, 3000]):
        nodes, t = synth.random_roots(400, R, 1000.0, seed=it, extra_ids=[403])
        got, want = part.sample(nodes, t), plain.sample(nodes, t)
        for gl, wl in zip(got, want):
            for gb, wb in zip(gl, wl):
                assert gb.num_dst_nodes() == wb.num_dst_nodes()
                for a, b in ((gb.srcdata["ID"], wb.srcdata["ID"]), (gb.srcdata["ts"], wb.srcdata["ts"]),
                             (gb.edata["ID"], wb.edata["ID"]), (gb.edata["dt"], wb.edata["dt"]),
                             (gb.edges()[0], wb.edges()[0]), (gb.edges()[1], wb.edges()[1])):
                    assert torch.equal(a, b)
    empty = part.sample(np.zeros(0, np.int64), np.zeros(0, np.float32))
    assert all(b.num_edges() == 0 and b.num_src_nodes() == 0 for mfg in empty for b in mfg)
    # pipelined use: several samples in flight, handed to the enqueue thread, FIFO results
    side = torch.cuda.Stream()
    reqs = [synth.random_roots(400, R, 1000.0, seed=50 + R) for R in (300, 2, 1500)]
    pend = [part.sample_async(torch.from_numpy(n).cuda(), torch.from_numpy(t).cuda(),
                              stream=side, worker_enqueue=True) for n, t in reqs]
    for (n, t), p in zip(reqs, pend):
        got, want = p.wait(), plain.sample(n, t)
        for gl, wl in zip(got, want):
            for gb, wb in zip(gl, wl):
                assert torch.equal(gb.edata["ID"], wb.edata["ID"])
                assert torch.equal(gb.srcdata["ID"], wb.srcdata["ID"])


def _same(gb, wb):
    ok = np.array_equal(gb.srcdata["ID"].cpu().numpy(), wb.srcdata["ID"])
    ok &= np.array_equal(gb.srcdata["ts"].cpu().numpy(), wb.srcdata["ts"])
    ok &= np.array_equal(gb.edata["ID"].cpu().numpy(), wb.edata["ID"])
    ok &= np.array_equal(gb.edata["dt"].cpu().numpy().view(np.uint8),
                         np.asarray(wb.edata["dt"]).view(np.uint8))
    ok &= np.array_equal(gb.edges()[0].cpu().numpy(), wb.edges()[0])
    ok &= np.array_equal(gb.edges()[1].cpu().numpy(), wb.edges()[1])
    return bool(ok)


def _worker(rank, world, port, ret, tiled, transport="torch"):
    sys.path.insert(0, ROOT)
    # torch: the stages issued from Python, exchanges staged through gloo + host memory;
    # ipc: the library's hipIpc transport — the NATIVE one-call chain with several ranks for real
    os.environ["GNNFLOW_PART_TRANSPORT"] = transport
    if tiled:   # read once per process: the count / scan / scatter form of the plan for every layer
        os.environ["GNNFLOW_PARTITION_SMALL_PLAN"] = "0"
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnflow_amd import DynamicGraph, TemporalSampler
        from gnnflow_amd.dist import DevicePartitionedSampler, PartitionedGraph
        from oracle import oracle as O
        from tests import synth
        src, dst, ts, eid = _graph()
        full = O.OracleGraph(minimum_block_size=8)
        shard = DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert")
        pg = PartitionedGraph(shard, rank, world)
        for lo in range(0, len(src), 2500):
            sl = slice(lo, lo + 2500)
            full.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=True)
            pg.add_edges(src[sl], dst[sl], ts[sl], eid[sl], add_reverse=True)
        ref = O.OracleSampler(full, [6, 4], "recent")
        ok = True
        # slack 2: the slotted exchange (no host synchronisation inside a sample); 0.02: slots so
        # small that they overflow -> the flag reaches every rank and the sample is redone through
        # the variable-size exchange; 0: the variable-size exchange itself
        for slack in (2.0, 0.02, 0.0):
            # 6000 roots: layer 1 is bounded by 42 000 > 32 768 roots -> the tiled plan and the
            # count / scan / emit merge
            sizes = [0, 1, 97, 600, 2000] + ([6000] if slack == 2.0 else [])
            # slot_roots: the ranks' batches differ in size, the slots must not.  With slack 0.02
            # it is left to the first sample's all-reduce (max of 0 and 1 roots): tiny slots
            part = DevicePartitionedSampler(TemporalSampler(shard, [6, 4], "recent"), slack=slack,
                                            slot_roots=max(sizes) if slack == 2.0 else None)
            for it in range(len(sizes)):
                # rotated per rank: in every round a different rank has an EMPTY batch and must
                # still take part in the layer's collectives (its peers would block otherwise)
                R = sizes[(it + rank) % len(sizes)]
                nodes, t = synth.random_roots(400, R, 1000.0, seed=1000 * rank + it, extra_ids=[403])
                got, want = part.sample(nodes, t), ref.sample(nodes, t)
                for gl, wl in zip(got, want):
                    for gb, wb in zip(gl, wl):
                        ok &= _same(gb, wb)
            if slack == 2.0:
                ok &= part.overflows == 0
            if slack > 0:
                ok &= (part._comm is not None) == (transport == "ipc")
            if slack == 0.02:
                # every rank counts the same overflowed samples (a flag raised on one rank is
                # seen by all of them)
                n_over = torch.tensor([part.overflows])
                both = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(both, n_over)
                ok &= part.overflows > 0 and all(int(x) == part.overflows for x in both)
                # three samples in flight on a side stream, overflowing ones among them: the
                # redo happens inside wait(), in the same order on every rank
                side = torch.cuda.Stream()
                reqs = [synth.random_roots(400, R, 1000.0, seed=77 * rank + R)
                        for R in (300, 3 + rank, 1500)]
                pend = [part.sample_async(torch.from_numpy(n).cuda(), torch.from_numpy(t).cuda(),
                                          stream=side, worker_enqueue=transport == "ipc")
                        for n, t in reqs]
                for (n, t), p in zip(reqs, pend):
                    for gl, wl in zip(p.wait(), ref.sample(n, t)):
                        for gb, wb in zip(gl, wl):
                            ok &= _same(gb, wb)
            # the per-layer call (reference: sample_layer_global), roots in the caller's order
            nodes, t = synth.random_roots(400, 50 * rank, 1000.0, seed=9 + rank)
            for layer, snap in ((0, 0), (1, 0)):
                gb = part.sample_layer(nodes, t, layer, snap)
                wb = ref.sample_layer(nodes, t, layer, snap)
                ok &= gb.num_dst_nodes() == wb.num_dst_nodes() and _same(gb, wb)
        torch.cuda.synchronize()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


# At most 4 ranks: the GPU box allows 6 processes on the card, and the pytest parent holds it
# too (world 8 lives on the CPU: tests/test_dist_gloo.py).
@pytest.mark.parametrize("world,tiled,transport", [
    (2, False, "torch"), (2, True, "torch"), (3, False, "torch"), (4, False, "torch"),
    (2, False, "ipc"), (2, True, "ipc"), (4, False, "ipc")])
def test_ranks_sharing_one_gpu_match_the_oracle(world, tiled, transport):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, tiled, transport), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _rccl_worker(rank, world, port, ret):
    """One rank, backend "nccl" (= RCCL): the multi-rank path of DevicePartitionedSampler —
    count exchange, asynchronous request all-to-all-v overlapped with the own share, served
    requests, reply all-to-all-v, merge — with every message empty, and FeatureShards' pull
    protocol; both straight on HBM tensors through RCCL, on the sampling stream."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from gnnflow_amd import DynamicGraph, TemporalSampler
        from gnnflow_amd.dist import DevicePartitionedSampler, _backend_is_host_only, _exchange
        from tests import synth
        assert not _backend_is_host_only(None)
        # the collective itself, on HBM tensors, also asynchronously
        a = torch.arange(12, dtype=torch.int64, device=dev).view(6, 2)
        b = torch.empty_like(a)
        w = _exchange(b, a, [6], [6], None, async_op=True)
        w.wait()
        ok = bool(torch.equal(a, b))
        src, dst, ts, eid = _graph()
        g = DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert")
        g.add_edges(src, dst, ts, eid, add_reverse=True)
        kw = dict(fanouts=[7, 5], sample_strategy="recent")
        plain = TemporalSampler(g, **kw)
        side = torch.cuda.Stream()
        # the slotted exchange — through the library's own RCCL communicator (one native call
        # per sample, also via the enqueue thread; with and without the communicator's side
        # stream) and through torch.distributed's equal-split all_to_all_single — and the
        # variable-size one (all-to-all-v); every message is empty
        for slack, transport, overlap in ((2.0, "native", True), (2.0, "native", False),
                                          (2.0, "torch", True), (0.0, "auto", True)):
            os.environ["GNNFLOW_PART_TRANSPORT"] = transport
            part = DevicePartitionedSampler(TemporalSampler(g, **kw), always_exchange=True,
                                            slack=slack, overlap=overlap)
            for it, R in enumerate([0, 1, 97, 600, 3000]):
                nodes, t = synth.random_roots(400, R, 1000.0, seed=it, extra_ids=[403])
                if it % 2:
                    got = part.sample_async(torch.from_numpy(nodes).to(dev),
                                            torch.from_numpy(t).to(dev), stream=side,
                                            worker_enqueue=transport == "native").wait()
                else:
                    got = part.sample(nodes, t)
                want = plain.sample(nodes, t)
                for gl, wl in zip(got, want):
                    for gb, wb in zip(gl, wl):
                        ok &= gb.num_dst_nodes() == wb.num_dst_nodes()
                        for x, y in ((gb.srcdata["ID"], wb.srcdata["ID"]), (gb.srcdata["ts"], wb.srcdata["ts"]),
                                     (gb.edata["ID"], wb.edata["ID"]), (gb.edata["dt"], wb.edata["dt"]),
                                     (gb.edges()[0], wb.edges()[0]), (gb.edges()[1], wb.edges()[1])):
                            ok &= bool(torch.equal(x, y))
            ok &= part.overflows == 0
            ok &= (part._comm is not None) == (slack > 0 and transport == "native")
            if transport == "native" and slack > 0:   # three samples in flight, enqueue thread
                reqs = [synth.random_roots(400, R, 1000.0, seed=50 + R) for R in (300, 2, 1500)]
                pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t_).to(dev),
                                          stream=side, worker_enqueue=True) for n, t_ in reqs]
                for (n, t_), p in zip(reqs, pend):
                    for gl, wl in zip(p.wait(), plain.sample(n, t_)):
                        for gb, wb in zip(gl, wl):
                            ok &= bool(torch.equal(gb.edata["ID"], wb.edata["ID"]) and
                                       torch.equal(gb.srcdata["ID"], wb.srcdata["ID"]))
            gb, wb = part.sample_layer(nodes, t, 1, 0), plain.sample_layer(nodes, t, 1, 0)
            ok &= bool(torch.equal(gb.edata["ID"], wb.edata["ID"]) and
                       torch.equal(gb.srcdata["ID"], wb.srcdata["ID"]))
        os.environ["GNNFLOW_PART_TRANSPORT"] = "auto"
        # the communicator's variable-size form: 3 + 5 bytes out of / into the middle of buffers
        from gnnflow_amd.dist import NativeComm
        from gnnflow_amd import _capi
        comm = NativeComm(dev)
        a8 = torch.arange(32, dtype=torch.uint8, device=dev)
        b8 = torch.zeros(32, dtype=torch.uint8, device=dev)
        one = C.c_size_t * 1
        _capi.check(_capi.load().gf_comm_all_to_all_v(
            comm.h, a8.data_ptr(), one(5), one(3), b8.data_ptr(), one(5), one(7), None))
        torch.cuda.synchronize()
        ok &= b8.tolist() == [0] * 7 + [3, 4, 5, 6, 7] + [0] * 20
        comm.close()
        # owner-sharded feature rows: ids out and rows back through RCCL (to this rank itself)
        from gnnflow_amd.dist import FeatureShards
        rng = np.random.RandomState(3)
        table = rng.rand(5000, 172).astype(np.float32)
        shards = FeatureShards.from_full(torch.from_numpy(table), np.arange(5000), 0, 1, dev)
        shards.always_exchange = True
        for n in (0, 1, 777, 4096):
            ids = rng.randint(0, 5000, n).astype(np.int64)
            rows = shards.pull(torch.from_numpy(ids).to(dev), torch.from_numpy(ids).to(dev))
            ok &= bool(np.array_equal(rows.cpu().numpy(), table[ids]))
        # Cache(distributed=True): the natively planned pull — count exchange, ids out, rows
        # back — through the library's communicator (every message to this rank itself)
        from tests.test_gpu_dist_features import _run as run_sharded_cache
        ok &= run_sharded_cache(0, 1, 0.1, always_exchange=True)
        torch.cuda.synchronize()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_multi_rank_path_over_rccl_with_one_rank():
    """No box with more than one GPU has been available, so RCCL never carried a message
    between two ranks; this at least runs the RCCL branch of the exchange — process-group
    initialisation with a device id, all_to_all_single on HBM tensors (synchronous and
    async_op + wait), stream ordering against the sampler's kernels on a side stream — with
    one rank, where the partitioned sampler normally short-circuits to the native chain."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, port, ret), nprocs=1, join=True)
    assert dict(ret) == {0: True}


def test_uniform_policy_through_the_padded_kernel_is_uniform():
    """Partitioned uniform sampling is distribution-matched (each owner draws from its own
    Philox stream): every in-window edge equally likely, all `fanout` slots filled (sampling
    with replacement), successive calls use fresh draws."""
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler
    n = 16
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 4, 64, "insert")
    g.add_edges(np.zeros(n, np.int64), np.arange(1, n + 1), np.arange(n, dtype=np.float32))
    part = DevicePartitionedSampler(TemporalSampler(g, [8], "uniform", seed=2024))
    R = 20000
    roots, ts = np.zeros(R, np.int64), np.full(R, 12.0, np.float32)
    b = part.sample(roots, ts)[0][0]
    picked = b.srcdata["ID"][R:].cpu().numpy()
    assert len(picked) == R * 8
    counts = np.bincount(picked, minlength=n + 1)[1:13]   # 12 candidates: dst 1..12
    assert counts.sum() == R * 8
    expected = R * 8 / 12.0
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert chi2 < 40.0, chi2   # 11 dof: P(chi2 > 40) ~ 4e-5
    again = part.sample(roots, ts)[0][0].srcdata["ID"][R:].cpu().numpy()
    assert not np.array_equal(picked, again)
    # a root without candidates gets no slot at all
    empty = part.sample(np.array([0, 5]), np.array([0.0, 12.0], np.float32))[0][0]
    assert empty.num_edges() == 0


def test_fused_merge_granules_survive_recycled_workspaces():
    """The fused merge's look-back granules live in the sampler's workspace.  Samplers that
    come and go reuse each other's freed memory, so a stale granule of an EARLIER sampler must
    never carry the tag of a later launch (tags are unique in the process, and a fresh
    workspace is cleared): create / sample / drop many samplers and compare every block."""
    import gc
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm
    from tests import synth
    src, dst, ts, eid = _graph()
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert")
    g.add_edges(src, dst, ts, eid, add_reverse=True)
    kw = dict(fanouts=[7, 5], sample_strategy="recent")
    plain = TemporalSampler(g, **kw)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    for round_ in range(8):
        comm = NativeComm.loopback(1, dev)[0]
        part = DevicePartitionedSampler(TemporalSampler(g, **kw), comm=comm, always_exchange=True,
                                        slot_roots=3000)
        sizes = [3000, 600, 97, 1500][round_ % 4:] + [2000, 1]
        reqs = [synth.random_roots(400, R, 1000.0, seed=31 * round_ + R) for R in sizes]
        pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev),
                                  stream=side) for n, t in reqs]
        for (n, t), p in zip(reqs, pend):
            for gl, wl in zip(p.wait(), plain.sample(n, t)):
                for gb, wb in zip(gl, wl):
                    for a, b in ((gb.srcdata["ID"], wb.srcdata["ID"]), (gb.edata["ID"], wb.edata["ID"]),
                                 (gb.edata["dt"], wb.edata["dt"]), (gb.edges()[1], wb.edges()[1])):
                        assert torch.equal(a, b), round_
        assert part.pairs >= 1
        del part, pend
        comm.close()
        gc.collect()
        torch.cuda.empty_cache()


def test_narrow_reply_slots_fall_back_when_an_id_outgrows_32_bits():
    """12-byte reply slots need every id below 2^32 - 1.  A graph that stops fitting (here: one
    edge id of 2^33) cannot change the wire format on its own rank's say-so: every sample of its
    chains is flagged as overflowed and redone through the wide form — same MFGs."""
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm
    from tests import synth
    src, dst, ts, eid = _graph()
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert")
    g.add_edges(src, dst, ts, eid, add_reverse=True)
    assert g.ids_fit_u32()
    kw = dict(fanouts=[7, 5], sample_strategy="recent")
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    comm = NativeComm.loopback(1, dev)[0]
    part = DevicePartitionedSampler(TemporalSampler(g, **kw), comm=comm, always_exchange=True,
                                    slot_roots=600, chain_samples=4, narrow_ids=True)
    plain = TemporalSampler(g, **kw)
    reqs = [synth.random_roots(400, R, 1000.0, seed=5 + R) for R in (600, 97, 300, 600)]

    def run():
        pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev),
                                  stream=side) for n, t in reqs]
        for (n, t), p in zip(reqs, pend):
            for gl, wl in zip(p.wait(), plain.sample(n, t)):
                for gb, wb in zip(gl, wl):
                    for a, b in ((gb.srcdata["ID"], wb.srcdata["ID"]), (gb.srcdata["ts"], wb.srcdata["ts"]),
                                 (gb.edata["ID"], wb.edata["ID"]), (gb.edata["dt"], wb.edata["dt"]),
                                 (gb.edges()[1], wb.edges()[1])):
                        assert torch.equal(a, b)
    run()
    assert part.overflows == 0 and part.chained == 4
    # one more edge whose id needs 34 bits, newer than everything: it is sampled
    g.add_edges(np.array([3]), np.array([5]), np.array([float(ts.max()) + 1.0], np.float32),
                np.array([1 << 33]))
    assert not g.ids_fit_u32()
    reqs[0] = (np.concatenate([reqs[0][0], [3]]),
               np.concatenate([reqs[0][1], [float(ts.max()) + 2.0]]).astype(np.float32))
    run()
    assert part.overflows == 4 and part.chained == 8
    comm.close()


@pytest.mark.parametrize("chain", [2, 3, 4])
def test_one_rank_shared_chains_without_exchange(chain):
    """World size 1, nothing to exchange: consecutive sample_async() calls still share a chain
    (gf_sampler_sample_partitioned_comm_group with a NULL communicator: plan, own shares, merge,
    publish — no all-to-all).  Ragged sizes, an empty batch, a batch beyond the agreed size whose
    layers exceed the shared chain's limit (an empty stand-in with the overflow flag forced, then
    the redo through the chain without slots), a chain that goes out before it is full."""
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler
    from tests import synth
    src, dst, ts, eid = _graph()
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert")
    g.add_edges(src, dst, ts, eid, add_reverse=True)
    kw = dict(fanouts=[7, 5], sample_strategy="recent")
    plain = TemporalSampler(g, **kw)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    part = DevicePartitionedSampler(TemporalSampler(g, **kw), slot_roots=600, chain_samples=chain,
                                    narrow_ids=(chain != 3))      # 12-byte reply slots / 24
    assert part.chain_samples == chain and part.lanes == 1
    sizes = [600, 97, 0, 1500, 1, 600, 3, 2000, 600, 5000, 5]    # 5000 x 8 > 32 768 roots
    reqs = [synth.random_roots(400, R, 1000.0, seed=77 + 3 * i + R) for i, R in enumerate(sizes)]
    for wave in (reqs[:7], reqs[7:]):
        pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev),
                                  stream=side, worker_enqueue=True) for n, t in wave]
        for (n, t), p in zip(wave, pend):
            for gl, wl in zip(p.wait(), plain.sample(n, t)):
                for gb, wb in zip(gl, wl):
                    for a, b in ((gb.srcdata["ID"], wb.srcdata["ID"]), (gb.srcdata["ts"], wb.srcdata["ts"]),
                                 (gb.edata["ID"], wb.edata["ID"]), (gb.edata["dt"], wb.edata["dt"]),
                                 (gb.edges()[0], wb.edges()[0]), (gb.edges()[1], wb.edges()[1])):
                        assert torch.equal(a, b), (chain, len(n))
    assert part.pairs >= 2 and part.chained >= 2 * 2
    assert part.overflows == 1          # the 5000-root batch, redone without slots


def _reused_roots():
    import ctypes
    from gnnflow_amd import _capi
    v = ctypes.c_uint64(0)
    _capi.check(_capi.load().gf_debug_part_reused_roots(ctypes.byref(v)))
    return int(v.value)


@pytest.mark.parametrize("fanouts,strategy,reuse,expect", [
    ([7, 7], "recent", True, True),       # layer 1's first R roots ARE layer 0's roots
    ([7, 5], "recent", True, False),      # another fanout: layer 0's block does not answer them
    ([7, 7], "uniform", True, False),     # fresh draws per layer
    ([7, 7], "recent", False, False),     # switched off (GNNFLOW_PART_REUSE_ROOTS=0)
])
def test_layer_roots_are_not_requested_twice(fanouts, strategy, reuse, expect):
    """gf_debug_part_reused_roots counts the roots a plan skipped because the previous layer's
    block already holds their edges (flags bit 1).  It moves exactly when the rule of DESIGN
    6.2 applies — equal fanouts, most-recent sampling, one snapshot — and the MFGs are the
    single-process sampler's either way."""
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.dist import DevicePartitionedSampler, NativeComm
    from tests import synth
    src, dst, ts, eid = _graph()
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 8, 64, "insert")
    g.add_edges(src, dst, ts, eid, add_reverse=True)
    kw = dict(fanouts=fanouts, sample_strategy=strategy, seed=11)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    comm = NativeComm.loopback(1, dev)[0]
    part = DevicePartitionedSampler(TemporalSampler(g, **kw), comm=comm, always_exchange=True,
                                    slot_roots=600, chain_samples=2, edge_fill=1.0,
                                    reuse_roots=reuse)
    plain = TemporalSampler(g, **kw)
    reqs = [synth.random_roots(400, R, 1000.0, seed=31 + R) for R in (600, 211)]
    before = _reused_roots()
    pend = [part.sample_async(torch.from_numpy(n).to(dev), torch.from_numpy(t).to(dev), stream=side)
            for n, t in reqs]
    got = [p.wait() for p in pend]
    moved = _reused_roots() - before
    assert part.chained == 2 and part.overflows == 0
    if expect:
        assert moved == sum(len(n) for n, _ in reqs), moved     # every layer-0 root, once
    else:
        assert moved == 0, moved
    if strategy == "recent":
        for (n, t), mf in zip(reqs, got):
            for gl, wl in zip(mf, plain.sample(n, t)):
                for gb, wb in zip(gl, wl):
                    for a, b in ((gb.srcdata["ID"], wb.srcdata["ID"]), (gb.srcdata["ts"], wb.srcdata["ts"]),
                                 (gb.edata["ID"], wb.edata["ID"]), (gb.edata["dt"], wb.edata["dt"]),
                                 (gb.edges()[0], wb.edges()[0]), (gb.edges()[1], wb.edges()[1])):
                        assert torch.equal(a, b)
    comm.close()
