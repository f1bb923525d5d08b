"""Host cost and device time of one small asynchronous H2D copy of pinned memory (the staging
ring's target range: 600 rows x 688 B = 413 KB) on a side stream."""
import time
import torch
dev = torch.device("cuda", 0)
src = torch.empty((1 << 24,), dtype=torch.float32).pin_memory()
dst = torch.empty((600 * 172,), dtype=torch.float32, device=dev)
st = torch.cuda.Stream()
for nbytes in (4096, 65536, 600 * 688, 1 << 22):
    n = nbytes // 4
    with torch.cuda.stream(st):
        d = torch.empty((n,), dtype=torch.float32, device=dev)
        for _ in range(10):
            d.copy_(src[:n], non_blocking=True)
        st.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            d.copy_(src[i * 1000:i * 1000 + n], non_blocking=True)
        t1 = time.perf_counter()
        st.synchronize()
        t2 = time.perf_counter()
    print("{:8d} B: host {:.1f} us per call, {:.1f} us per copy end to end ({:.1f} GB/s)".format(
        nbytes, 1e6 * (t1 - t0) / 200, 1e6 * (t2 - t0) / 200, nbytes / ((t2 - t0) / 200) / 1e9))
