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
from typing import Callable, List, Optional, Sequence, Tuple

import torch


class ReplayPipeline:
    """Replays device-resident batches `(roots, timestamps, eids)` through
    `sampler.sample` + `cache.fetch_feature`.

    pipelined=True : sample_async(stream=side, worker_enqueue=True) one batch ahead,
                     fetch_feature(async_enqueue=True) on the current stream.
    pipelined=False: the plain loop `mfgs = sampler.sample(r, t); cache.fetch_feature(mfgs, e)`.
    cache=None     : sampling only.
    """

    def __init__(self, sampler, cache, batches: Sequence[Tuple[torch.Tensor, torch.Tensor,
                                                                torch.Tensor]],
                 device: torch.device, pipelined: bool = True):
        self.sampler, self.cache, self.batches = sampler, cache, batches
        self.device = torch.device(device)
        self.pipelined = bool(pipelined) and cache is not None and \
            hasattr(sampler, "sample_async")
        self.side = torch.cuda.Stream(device=self.device) if self.pipelined else None

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
        sampler, cache, side = self.sampler, self.cache, self.side
        main = torch.cuda.current_stream(self.device)
        r, t, _ = self.batches[first % nb]
        pending = sampler.sample_async(r, t, stream=side, worker_enqueue=True)
        for i in range(first, first + count):
            mfgs = pending.wait()
            if i + 1 < first + count:
                r, t, _ = self.batches[(i + 1) % nb]
                pending = sampler.sample_async(r, t, stream=side, worker_enqueue=True)
            for mfg in mfgs:
                for b in mfg:
                    b.record_stream(main)
            cache.fetch_feature(mfgs, self.batches[i % nb][2], async_enqueue=True)
            if on_step:
                on_step(i % nb, mfgs)
        cache.wait_enqueued()
