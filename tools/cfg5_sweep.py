#!/usr/bin/env python3
"""BASELINE config 5 on the sample volume: held-out-frame PSNR for a few encoder geometries and
training lengths (GPU box).  Even frames are trained on, odd frames are held out; the number to
beat is linear interpolation in t (interp.py's baseline)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from mri_interpolation_amd import _lib, datamodules, models, trainer

_lib.load()
dev = torch.device("cuda", 0)
w = dict(bench.WORKLOADS["cfg5"])
vol, _ = bench.load_volume(w, dev)
even = datamodules.MriImage(volume=vol, device=dev, frames=slice(0, None, 2))
odd = datamodules.MriImage(volume=vol, device=dev, frames=slice(1, None, 2))
ev = even.pixels.view(even.shape)
n_odd = odd.shape[-1]
linear = 0.5 * (ev[..., :n_odd] + ev[..., 1:n_odd + 1])
print(f"linear interpolation in t: {trainer.psnr(linear.reshape(-1, 1), odd.pixels):.3f} dB", flush=True)
fin = 16 * 1.4 ** 15
CASES = {
    "cfg5 (16,16,5,7)->(2489,2489,5,7) L16": dict(base=(16, 16, 5, 7), finest=(fin, fin, 5, 7), levels=16),
    "xy->352": dict(base=(16, 16, 5, 7), finest=(352, 352, 5, 7), levels=16),
    "xy->176": dict(base=(16, 16, 5, 7), finest=(176, 176, 5, 7), levels=16),
    "xy->352, t 3->7": dict(base=(16, 16, 5, 3), finest=(352, 352, 5, 7), levels=16),
    "xy->352, t 2->7, L8": dict(base=(16, 16, 5, 2), finest=(352, 352, 5, 7), levels=8),
    "xy->352, L8": dict(base=(16, 16, 5, 7), finest=(352, 352, 5, 7), levels=8),
    "xy->704": dict(base=(16, 16, 5, 7), finest=(704, 704, 5, 7), levels=16),
}
for name, c in CASES.items():
    torch.manual_seed(1337)
    net = models.HashMLP(dim_in=4, n_levels=c["levels"], n_features_per_level=2, log2_hashmap_size=19,
                         base_resolution=c["base"], finest_resolution=c["finest"], dim_hidden=128,
                         dim_out=1, n_layers=3, activation=torch.nn.ReLU, batch_norm=False,
                         final_activation=False, lr=5e-3).to(dev)
    step = trainer.FusedStep(net, net.configure_optimizers())
    loader = datamodules.DeviceLoader(even, 1 << 18, shuffle=True, drop_last=True, seed=1337)
    pipe = datamodules.BatchPipeline(loader)
    done, out = 0, {}
    for stop in (250, 500, 1000, 2000, 4000):
        while done < stop:
            x, y = pipe.current()
            step.train_step(x, y, pipe.produce_next)
            pipe.advance()
            done += 1
        with torch.no_grad():
            held = torch.cat([step.forward(x)[0].clone()
                              for x, _ in datamodules.DeviceLoader(odd, 1 << 20, shuffle=False)])
            fit = torch.cat([step.forward(x)[0].clone()
                             for x, _ in datamodules.DeviceLoader(even, 1 << 20, shuffle=False)])
        out[stop] = (round(trainer.psnr(held, odd.pixels), 3), round(trainer.psnr(fit, even.pixels), 2))
    print(f"{name:42s} held-out / trained dB at steps: {json.dumps(out)}", flush=True)
