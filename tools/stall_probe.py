#!/usr/bin/env python3
"""Hunt the rare ~50 ms host stall inside a queued step (round 4: one step of a 20-step bench leg took 50-58 ms
on the host on about every third run).  Tools-only build with -DMRI_STEP_TRACE:
    python tools/build_variant.py --name=libmri_trace.so -DMRI_STEP_TRACE      # here
    MRI_LIB=tools/libmri_trace.so python tools/stall_probe.py [bursts] [steps]  # on the GPU box
Runs `bursts` bursts of `steps` natively queued steps, each burst behind a synchronize (the shape of a bench
leg), and prints every step whose host time exceeds 3 ms with the per-call nanoseconds inside mri_fused_step."""
import ctypes as C
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from mri_interpolation_amd import _lib, datamodules, trainer

bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
sample_every = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # > 0: every n-th step carries the five phase events
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
names = ["join+fork", "sample", "gather", "memset absmax", "count next", "lookup", "decoder", "table gradient", "adam",
         "record join"]
has_trace = hasattr(lib, "mri_debug_step_trace")
a, b = (C.c_longlong * 25)(), (C.c_longlong * 25)()
if os.environ.get("MRI_NO_GC"):
    gc.disable()
found = 0
t_all = time.perf_counter()
for burst in range(bursts):
    loop.finish()
    torch.cuda.synchronize()
    for i in range(steps):
        if has_trace:
            lib.mri_debug_step_trace(a)
        sample = sample_every > 0 and i % sample_every == 3
        step.phase_events = {} if sample else None
        t0 = time.perf_counter()
        loop.step_once(sample=sample)
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 3.0:
            found += 1
            msg = f"burst {burst} step {i}{' (sampled)' if sample else ''}: {dt:.1f} ms on the host"
            if has_trace:
                lib.mri_debug_step_trace(b)
                parts = [(names[j], (b[j] - a[j]) / 1e6) for j in range(len(names))]
                msg += "; inside the call: " + ", ".join(f"{n} {v:.2f}" for n, v in parts if v > 0.05) + \
                       f" (sum {sum(v for _, v in parts):.1f} ms)"
            print(msg, flush=True)
loop.finish()
torch.cuda.synchronize()
print(f"{bursts} bursts x {steps} steps in {time.perf_counter() - t_all:.1f} s: {found} stalled steps")
