#!/usr/bin/env python3
"""Decoder gradients of one config-4 step at full batch against float64: the kernel (MRI_LIB selects the build)
and the f32 oracle.  Probe for MRI_DW_TERMS (tools/build_variant.py -DMRI_DW_TERMS=3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import rel_err
from mri_interpolation_amd import _lib, models, trainer
from oracle import detrand, hashgrid as ohash, mlp as omlp, train as otrain

_lib.load()
n, lr, finest = 1 << 18, 5e-3, 16 * 1.4 ** 15
model = otrain.HashMlpModel(3, 16, 2, 19, 16, finest, [128, 128], seed=11, table_scale=1e-4)
net = models.HashMLP(3, 16, 2, 19, 16, finest, dim_hidden=128, n_layers=3, activation=torch.nn.ReLU,
                     batch_norm=False, final_activation=False, lr=lr)
with torch.no_grad():
    net.encoder.table.copy_(torch.cat(model.tables))
    for blk, (w, b) in zip(net.decoder, model.mlp):
        blk[0].weight.copy_(w)
        blk[0].bias.copy_(b)
net = net.cuda()
x = torch.from_numpy(detrand.uniform(n * 3, 41, 0.0, 1.0).reshape(n, 3))
y = torch.from_numpy(detrand.uniform(n, 42, 0.0, 1.0).reshape(n, 1))
torch.set_num_threads(16)
_, _, grads = otrain.loss_and_grads(model, x, y)
step = trainer.FusedStep(net, net.configure_optimizers())
_, ws = step.forward(x.cuda(), train=True)
step.backward(x.cuda(), y.cuda(), ws)
z = ohash.encode(x, model.tables, model.resolutions).double()
p64 = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in model.mlp]
omlp.mse_loss(omlp.relu_mlp_forward(z, p64, False), y.double()).backward()
print("lib:", os.environ.get("MRI_LIB", "in-tree"))
for i, (blk, (w64, b64)) in enumerate(zip(net.decoder, p64)):
    for name, got, ref32, want in (("gw", blk[0].weight.grad, grads[16 + 2 * i], w64.grad),
                                   ("gb", blk[0].bias.grad, grads[17 + 2 * i], b64.grad)):
        k = rel_err(got.cpu().numpy(), want.numpy())
        o = rel_err(ref32.numpy(), want.numpy())
        print(f"  {name}{i}: kernel {k[0]:.2e} (max) {k[1]:.2e} (L2)   f32 oracle {o[0]:.2e} {o[1]:.2e}")
