#!/usr/bin/env python3
"""What the counting stage of the table gradient costs a config-4 step: the same FIXED batch every step
(so the counts of the first steps stay valid), with the stage queued as usual and with it skipped.
The difference bounds what folding the count into another kernel (the lookup computes the same hashes)
could gain.  Timing tool, not a test."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from mri_interpolation_amd import _lib, ops, trainer

_lib.load()
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["cfg4"]
model = bench.build_model(w).to(dev)
step = trainer.FusedStep(model, model.configure_optimizers())
n = w["batch"]
torch.manual_seed(0)
coords = torch.rand(n, 3, device=dev)
target = torch.rand(n, 1, device=dev)


def run(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.train_step(coords, target, lambda: coords)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


run(50)
real = ops.hashgrid_backward_prepare
for rep in range(3):
    a = run(300)
    ops.hashgrid_backward_prepare = lambda *args, **kw: None  # both workspaces hold this batch's counts
    b = run(300)
    ops.hashgrid_backward_prepare = real
    print(f"with the counting stage {a:.4f} ms/step, without {b:.4f} ms/step", flush=True)
