"""Time of ops.order_batch alone (config-4 batch: 2^18 indices of a 256^3 volume): python tools/order_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import _lib, ops
_lib.load()
dev = torch.device("cuda", 0)
for shape, n in (((256, 256, 256), 1 << 18), ((352, 352, 6, 15), 1 << 18), ((256, 256, 256), 1 << 20)):
    total = 1
    for e in shape:
        total *= e
    idx0 = torch.randperm(total, device=dev)[:n].contiguous()
    ws = ops.order_batch_workspace(n, len(shape), dev)
    idx = idx0.clone()
    for _ in range(5):
        idx.copy_(idx0); ops.order_batch(idx, shape, ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    a.record()
    for _ in range(reps):
        ops.order_batch(idx, shape, ws)   # (re-ordering an ordered batch: the same work)
    b.record()
    torch.cuda.synchronize()
    print(shape, n, "order_batch %.1f us" % (a.elapsed_time(b) / reps * 1e3), flush=True)
