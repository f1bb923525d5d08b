"""The software-pipelined hot-path loop: sample() of batch i+1 overlapped with
fetch_feature() of batch i.

The reference's training loop prefetches the next batch's `sampler.sample` on a Python
thread while the current batch is fetched and trained
(scripts/offline_edge_prediction.py:343-346,397-405).  Here one Python thread does the
same: batch i+1's sampling kernels are enqueued on a side HIP stream
(`TemporalSampler.sample_async`, launches issued by the library's enqueue thread) before
batch i's `Cache.fetch_feature` is issued on the caller's stream, so the two overlap on
the GPU.  Every batch goes through the same calls as the plain loop; nothing is skipped.

bench.py times this loop and tests/test_gpu_pipeline_parity.py checks THIS loop — the same
function — against the CPU oracle, so the path that is measured is the path that is tested.
"""
import os
from collections import deque
from typing import Callable, List, Optional, Sequence, Tuple

import torch


_SIDE_STREAMS = {}


def _shares_queue(device, a: torch.cuda.Stream, b: torch.cuda.Stream) -> bool:
    import ctypes as C
    from . import _capi
    out = C.c_int(0)
    _capi.check(_capi.load().gf_streams_share_queue(
        device.index, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), 150, C.byref(out)))
    return bool(out.value)


def side_stream(device, k: int = 0, priority: int = 0) -> torch.cuda.Stream:
    """The k-th side stream of `device`, created once per process.  HIP spreads its streams over
    four hardware queues, and two busy streams that land on one queue serialise: a pipeline that
    made a fresh stream per sampling lane shifted the mapping of every stream created after it
    (the hash-partitioned loop timed in the same process went from 31.6 to 46.7 us per step), and
    the staging ring's pull stream, created as the process's fifth stream, shared the FETCH
    stream's queue (50 us per pinned step against 39).  So a side stream is not taken as it comes:
    a candidate that shares its queue with the current stream or with a side stream picked before
    (include/gnnflow_hip.h gf_streams_share_queue: does a small kernel on the one finish while a
    spinning kernel holds the other?) is set aside and the next one is tried — the four busy
    streams of a pipelined step (fetch, two sampling lanes, the pull) end up on the four queues
    whatever else the process created before.  The pipeline's sampling lanes and the partitioned
    sampler's lanes take their streams from this one small set."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device.index, int(k))
    st = _SIDE_STREAMS.get(key)
    if st is None:
        probe = os.environ.get("GNNFLOW_STREAM_PROBE", "1") != "0" and priority == 0
        taken = [torch.cuda.current_stream(device)] + \
            [v for (d, _k), v in _SIDE_STREAMS.items() if d == device.index]
        rejected = _SIDE_STREAMS.setdefault(("rejected", device.index), [])
        for _ in range(8 if probe else 1):
            st = torch.cuda.Stream(device=device, priority=priority)
            if not probe or len(taken) >= 4 or not any(_shares_queue(device, t, st) for t in taken):
                break
            rejected.append(st)     # (kept alive: a stream given back may be handed out again)
        _SIDE_STREAMS[key] = st
    return st


class ReplayPipeline:
    """Replays device-resident batches `(roots, timestamps, eids)` through
    `sampler.sample` + `cache.fetch_feature`.

    pipelined=True : sample_async(stream=side, worker_enqueue=True) `depth` batches ahead
                     (default 3 with two sampling lanes, 2 with one: the sampling of batch
                     i+depth is issued before batch i+1's is waited for, which takes the
                     sampler's issue + execution latency off the critical path),
                     fetch_feature(async_enqueue=True) on the current stream.
    pipelined=False: the plain loop `mfgs = sampler.sample(r, t); cache.fetch_feature(mfgs, e)`.
    cache=None     : sampling only; pipelined=True keeps `depth` (at least 2 per lane) samples in
                     flight over the lanes and waits for them in batch order — what
                     benchmarks/benchmark_sampler.py:71-87 times, without its host wait per batch.
    sample_lanes   : consecutive batches are sampled round-robin by this many samplers (clones
                     of `sampler` over the same graph: own native workspace, own side stream),
                     so the four dependent launches of one sample() overlap the next one's.
                     Uniform draws are keyed by the sampler's call counter: every sample is
                     begun with the call number it would have had on the one sampler
                     (sample_async(call_base=...)), so the lanes reproduce its stream of draws
                     bit for bit.  Default: GNNFLOW_SAMPLE_LANES or 2.
    """

    def __init__(self, sampler, cache, batches: Sequence[Tuple[torch.Tensor, torch.Tensor,
                                                                torch.Tensor]],
                 device: torch.device, pipelined: bool = True, depth: Optional[int] = None,
                 sample_lanes: Optional[int] = None):
        self.sampler, self.cache, self.batches = sampler, cache, batches
        self.device = torch.device(device)
        self.pipelined = bool(pipelined) and hasattr(sampler, "sample_async")
        self.side = side_stream(self.device, 0) if self.pipelined else None
        if sample_lanes is None:
            sample_lanes = int(os.environ.get("GNNFLOW_SAMPLE_LANES", "2"))
        self.lanes = [(sampler, self.side)]
        if self.pipelined and sample_lanes > 1 and hasattr(sampler, "clone") and \
                getattr(sampler, "_strategy", None) in ("recent", "uniform") and \
                hasattr(sampler, "set_call_counter") and \
                not hasattr(sampler, "chain_samples"):
            # sampling only: the loop runs at the pace of the thread that issues the samples'
            # launches (19 us per sample) unless each lane has an issuer of its own; with a cache
            # the fetch's issuer is a third busy thread and a fourth did not pay (profiles/README)
            two_issuers = cache is None and \
                os.environ.get("GNNFLOW_SAMPLE_ENQUEUE_THREADS", "2") != "1"
            for k in range(1, min(int(sample_lanes), 4)):
                clone = sampler.clone()
                if two_issuers and k % 2 == 1 and hasattr(clone, "set_enqueue_lane"):
                    clone.set_enqueue_lane(2)
                self.lanes.append((clone, side_stream(self.device, k)))
        # a sampler (of a lane of the partitioned sampler: one per sample of a shared chain)
        # holds 4 begun samples at most
        if depth is None:
            # two sampling lanes: a sample's chain (4 launches through the enqueue thread, ~45 us
            # from begin to done) must be three steps ahead of its fetch, or the loop waits for it
            # (26.2-27.2 against 28.8-29.0 us per step on a slow host, same box)
            depth = 3 if len(self.lanes) > 1 else 2
        self.depth = max(1, min(int(depth), 3 * max(1, getattr(self.sampler, "lanes", 1)) *
                                max(1, getattr(self.sampler, "chain_samples", 1))))
        # GNNFLOW_PIPELINE_FETCH_FIRST=1: submit batch i's fetch before the sample of batch
        # i + depth (the chains of a partitioned sampler over a communicator share the fetches'
        # issuing thread; measured: no consistent difference, profiles/README.md round 4)
        ff = os.environ.get("GNNFLOW_PIPELINE_FETCH_FIRST")
        self.fetch_first = ff is not None and ff != "0"

    def step(self, i: int):
        r, t, e = self.batches[i % len(self.batches)]
        mfgs = self.sampler.sample(r, t)
        if self.cache is not None:
            self.cache.fetch_feature(mfgs, e)
        return mfgs

    def run(self, first: int, count: int,
            on_step: Optional[Callable[[int, List[List]], None]] = None) -> None:
        """Runs `count` steps over batches first, first+1, ... (modulo the number of
        batches); calls on_step(batch_index, mfgs) after each step's work was issued."""
        if count <= 0:
            return
        nb = len(self.batches)
        if not self.pipelined:
            for i in range(first, first + count):
                mfgs = self.step(i)
                if on_step:
                    on_step(i % nb, mfgs)
            return
        cache, lanes, nl = self.cache, self.lanes, len(self.lanes)
        batches = self.batches
        main = torch.cuda.current_stream(self.device)
        last = first + count
        pending = deque()
        nxt = first

        # several lanes + uniform sampling: sample j draws from the call number it would have on
        # the one sampler; afterwards the primary sampler's counter stands where it would
        keyed = nl > 1 and getattr(self.sampler, "_strategy", None) == "uniform"
        if keyed:
            per = self.sampler.calls_per_sample
            call0 = self.sampler.call_counter() - first * per

        def begin(j):
            r, t, _ = batches[j % nb]
            sampler, side = lanes[j % nl]
            if keyed:
                return sampler.sample_async(r, t, stream=side, worker_enqueue=True,
                                            call_base=call0 + j * per)
            return sampler.sample_async(r, t, stream=side, worker_enqueue=True)

        # Host-resident feature tables (feature_placement="pinned" with a staging ring): batch
        # i+1's sample is waited for one step early and its coming misses are pulled over the
        # host link on a side stream while batch i is fetched (Cache.prefetch_feature; the
        # reference moves a miss host -> pinned -> device inside fetch_feature,
        # gnnflow/cache/cache.py:288-313,381-388) — one more sample in flight pays for the
        # earlier wait.
        if cache is None:
            depth = max(self.depth, 2 * nl)
            while nxt < last and len(pending) < depth:
                pending.append(begin(nxt))
                nxt += 1
            for i in range(first, last):
                mfgs = pending.popleft().wait()
                if nxt < last:
                    pending.append(begin(nxt))
                    nxt += 1
                if on_step:
                    on_step(i % nb, mfgs)
            if keyed:
                self.sampler.set_call_counter(call0 + last * per)
            return
        staged = bool(getattr(cache, "staging", False))
        depth = self.depth
        while nxt < last and len(pending) < depth:
            pending.append(begin(nxt))
            nxt += 1
        fetch_first = self.fetch_first
        if staged:
            # Batch i+2's announcement rides in batch i's fetch submission: the pull has two steps
            # to land, and the fetch of batch i depends on the announcements up to batch i's own
            # only (set_staging_lag), so that it finds the event it needs complete when it is
            # issued — a stream that really has to wait for another stream's event loses
            # 12-20 us per hand-over (profiles/README.md, round 6).
            lead = max(1, min(int(os.environ.get("GNNFLOW_STAGE_LEAD", "2")), 3))
            ready = deque()           # MFGs waited for and announced, in batch order
            cache.set_staging_lag(0)
            j = first
            while j < last and len(ready) < lead:
                m = pending.popleft().wait()
                if nxt < last:
                    pending.append(begin(nxt))
                    nxt += 1
                cache.prefetch_feature(m, batches[j % nb][2], async_enqueue=True)
                ready.append(m)
                j += 1
            cache.set_staging_lag(lead - 1)
            for i in range(first, last):
                mfgs = ready.popleft()
                for mfg in mfgs:
                    for b in mfg:
                        b.record_stream(main)
                ann = None
                if j < last:
                    ahead = pending.popleft().wait()
                    if nxt < last:
                        pending.append(begin(nxt))
                        nxt += 1
                    ann = (ahead, batches[j % nb][2])
                    ready.append(ahead)
                    j += 1
                elif last - 1 - i < lead - 1:
                    # the tail: no announcement rides along any more — the newest generation
                    # moves no further, so the fetches' lag shrinks with the batches left
                    cache.set_staging_lag(last - 1 - i)
                cache.fetch_feature(mfgs, batches[i % nb][2], async_enqueue=True, announce=ann)
                if on_step:
                    on_step(i % nb, mfgs)
            cache.wait_enqueued()
            cache.set_staging_lag(0)
            if keyed:
                self.sampler.set_call_counter(call0 + last * per)
            return
        for i in range(first, last):
            mfgs = pending.popleft().wait()
            if nxt < last and not fetch_first:
                pending.append(begin(nxt))
                nxt += 1
            for mfg in mfgs:
                for b in mfg:
                    b.record_stream(main)
            cache.fetch_feature(mfgs, batches[i % nb][2], async_enqueue=True)
            if nxt < last and fetch_first:
                pending.append(begin(nxt))
                nxt += 1
            if on_step:
                on_step(i % nb, mfgs)
        cache.wait_enqueued()
        if keyed:
            self.sampler.set_call_counter(call0 + last * per)
