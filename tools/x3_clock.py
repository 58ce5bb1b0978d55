#!/usr/bin/env python3
"""The shader clock the bf16x3 decoder's tile loop runs at (MI355X_MICROARCH.md, 'DVFS give-back' item 6): shader
cycles (s_memtime) per 100 MHz tick (s_memrealtime), one pair of stamps around the whole loop of a tools-only build
(-DMRI_X3_PROFILE -DMRI_X3_CLOCK_ONLY, the whole library: the stamps sit in the decoder only), median over
workgroups -- (a) in 2 s of back-to-back decoder launches on random data, (b) inside config 4's real training step.

    python tools/x3_clock.py --build-only     # here (cross-compile), the .so travels
    python tools/x3_clock.py                  # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libmri_x3clock.so")
if "--build-only" in sys.argv:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "--name=libmri_x3clock.so",
                           "-DMRI_X3_PROFILE", "-DMRI_X3_CLOCK_ONLY"])
    sys.exit(0)
os.environ["MRI_LIB"] = LIB
import torch

import bench
from mri_interpolation_amd import _lib, datamodules, ops, trainer

lib = _lib.load()
dev = torch.device("cuda", 0)
prof = torch.zeros(256 * 8 * 32, dtype=torch.int64, device=dev)
assert lib.mri_debug_set_x3_profile(C.c_void_p(prof.data_ptr())) == 0


def clock(label, ms):
    p = prof.cpu().reshape(256, 8, 32).double()
    cyc, ticks = p[:, :, 24], p[:, :, 25]
    ghz = (cyc / ticks.clamp_min(1) * 0.1).flatten()
    ghz = ghz[ticks.flatten() > 0]
    tiles = (1 << 18) / 32 / 256
    print(f"{label}: in-kernel clock {float(ghz.median()):.3f} GHz (min {float(ghz.min()):.3f}, max {float(ghz.max()):.3f}); "
          f"{float(cyc.mean()) / tiles:.0f} cycles per tile"
          + (f"; decoder {ms:.4f} ms -> at 2.4 GHz the same cycles would take {ms * float(ghz.median()) / 2.4:.4f} ms"
             if ms is not None else ""), flush=True)


# (a) the decoder alone, back to back, random data
n, k_in, H = 1 << 18, 32, 128
torch.manual_seed(0)
params = [(torch.randn(H, k_in, device=dev) * 0.2, torch.randn(H, device=dev) * 0.1),
          (torch.randn(H, H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1),
          (torch.randn(1, H, device=dev) * 0.1, torch.randn(1, device=dev) * 0.1)]
x = torch.rand(k_in, n, device=dev) * 2 - 1
t = torch.rand(n, 1, device=dev)
grads = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in params]
dx, y, loss = torch.empty_like(x), torch.empty(n, 1, device=dev), torch.zeros(1, device=dev)
run = lambda: ops.tiny_mlp_train(x, t, params, grads, loss, d_x=dx, y=y, overwrite=True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    run()
b.record()
torch.cuda.synchronize()
clock("decoder alone, back to back", a.elapsed_time(b) / 50)

# (b) inside the real step of config 4
w = bench.WORKLOADS["cfg4"]
vol, _ = bench.load_volume(w, dev)
ds = datamodules.MriImage(volume=vol, device=dev)
loader = datamodules.DeviceLoader(ds, w["batch"], shuffle=True, drop_last=True, seed=1337)
model = bench.build_model(w).to(dev)
step = trainer.FusedStep(model, model.configure_optimizers())
pipe = datamodules.BatchPipeline(loader)
loop = trainer.SteadyLoop(step, pipe, mode="native").capture()
for _ in range(4000):
    loop.step_once()
loop.finish()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    loop.step_once()
loop.finish()
torch.cuda.synchronize()
ms_step = (time.perf_counter() - t0) / 400 * 1e3
print(f"config-4 step {ms_step:.4f} ms")
clock("decoder inside the config-4 step", None)
