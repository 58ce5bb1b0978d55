#!/usr/bin/env python3
"""Where the HOST spends its time queueing a config-4 step (cProfile over 400 steps)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from mri_interpolation_amd import _lib, datamodules, trainer

_lib.load()
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["cfg4"]
vol, _ = bench.load_volume(w, dev)
ds = datamodules.MriImage(volume=vol, device=dev)
loader = datamodules.DeviceLoader(ds, w["batch"], shuffle=True, drop_last=True, seed=1337)
model = bench.build_model(w).to(dev)
step = trainer.FusedStep(model, model.configure_optimizers())
pipe = datamodules.BatchPipeline(loader)


def one():
    c, t = pipe.current()
    step.train_step(c, t, pipe.produce_next, late_work=pipe.produce_late)
    pipe.advance()


for _ in range(100):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    one()
host = (time.perf_counter() - t0) / 400 * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 400 * 1e3
print(f"host queue {host:.4f} ms/step, with the GPU {total:.4f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(400):
    one()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
