#!/usr/bin/env python3
"""Host nanoseconds per call inside mri_fused_step (tools-only build with -DMRI_STEP_TRACE):
    python tools/build_variant.py --name=libmri_trace.so -DMRI_STEP_TRACE      # here
    MRI_LIB=tools/libmri_trace.so python tools/step_trace.py                    # on the GPU box
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from mri_interpolation_amd import _lib, datamodules, trainer

lib = _lib.load()
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["cfg4"]
vol, _ = bench.load_volume(w, dev)
ds = datamodules.MriImage(volume=vol, device=dev)
loader = datamodules.DeviceLoader(ds, w["batch"], shuffle=True, drop_last=True, seed=1337)
model = bench.build_model(w).to(dev)
step = trainer.FusedStep(model, model.configure_optimizers())
pipe = datamodules.BatchPipeline(loader)
loop = trainer.SteadyLoop(step, pipe, mode="native").capture()
for _ in range(100):
    loop.step_once()
torch.cuda.synchronize()
buf0 = (C.c_longlong * 25)()
lib.mri_debug_step_trace(buf0)
t0 = time.perf_counter()
for _ in range(400):
    loop.step_once()
host = (time.perf_counter() - t0) / 400 * 1e6
loop.finish()
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 400 * 1e6
buf = (C.c_longlong * 25)()
lib.mri_debug_step_trace(buf)
names = ["join+fork (3 calls)", "sample", "gather", "memset absmax", "count next (2 memsets + 3 kernels)", "lookup",
         "decoder (+ slab reduce)", "table gradient (3 kernels)", "adam", "record join"]
calls = buf[24] - buf0[24]
print(f"host {host:.1f} us/step queueing, {total:.1f} us/step with the GPU; inside mri_fused_step:")
inside = 0.0
for i, nm in enumerate(names):
    us = (buf[i] - buf0[i]) / calls / 1e3
    inside += us
    print(f"  {nm:40s} {us:8.1f} us")
print(f"  {'sum':40s} {inside:8.1f} us   (Python around the call: {host - inside:.1f} us)")
