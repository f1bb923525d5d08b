"""What the host link gives on this box: (a) hipMemcpy of a pinned buffer (the DMA engines),
(b) the gather kernel reading scattered 688-byte rows of a pinned table zero-copy (what a missed
row of feature_placement="pinned" costs), at several row counts.  Prints one JSON line each."""
import ctypes as C
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from gnnflow_amd import _capi  # noqa: E402

lib = _capi.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
d = 172
N = 2_000_000                       # 1.4 GB of pinned rows
table = torch.empty((N, d), dtype=torch.float32).pin_memory()
table.uniform_()
dst = torch.empty((1 << 26,), dtype=torch.float32, device=dev)   # 256 MB
src = table.view(-1)[:1 << 26]
for _ in range(2):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(json.dumps({"probe": "hipMemcpy pinned->device 256 MB", "GBps": dst.numel() * 4 / dt / 1e9}))
g = torch.Generator(device=dev).manual_seed(1)
st = _capi.current_stream(dev)
for rows in (512, 2048, 8192, 32768, 131072):
    ids = torch.randint(0, N, (rows,), generator=g, device=dev)
    out = torch.empty((rows, d), dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        if rep == 2:
            e0.record()
        _capi.check(lib.gf_gather_rows(table.data_ptr(), N, d, ids.data_ptr(), rows,
                                       out.data_ptr(), 0, st))
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1)
    assert torch.equal(out.cpu(), table[ids.cpu()])
    print(json.dumps({"probe": "zero-copy gather of pinned rows", "rows": rows,
                      "bytes": rows * d * 4, "us": us, "GBps": rows * d * 4 / us / 1e3}))
