import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import models, trainer
torch.manual_seed(2)
net = models.HashMLP(3, 16, 2, 19, 16, 512, dim_hidden=64, n_layers=3, activation=torch.nn.ReLU, batch_norm=False, final_activation=False).cuda()
with torch.no_grad():
    net.encoder.table.uniform_(-0.5, 0.5)
step = trainer.FusedStep(net, net.configure_optimizers())
x = torch.rand(30000, 3, device="cuda"); y = torch.rand(30000, 1, device="cuda")
_, ws = step.forward(x, train=True)
step.backward(x, y, ws); want = step.flat.grad.clone()
step.backward(x, y, ws); print("repeat equal:", torch.equal(step.flat.grad, want))
step.world = 2; step.grad_buckets = 1
step.backward(x, y, ws); g = step.flat.grad * 2
print("world2 single launch equal:", torch.equal(g, want), float((g - want).abs().max()))
step.grad_buckets = 4
step.backward(x, y, ws); g = step.flat.grad * 2
d = (g - want).abs()
print("bucketed equal:", torch.equal(g, want), float(d.max()), int((d > 0).sum()), "of", d.numel())
nz = torch.nonzero(d > 0).flatten()
print(nz[:10].tolist(), nz[-10:].tolist())
enc = net.encoder
print([enc._row_span(l) for l in range(16)])
