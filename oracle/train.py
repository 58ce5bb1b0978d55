"""Oracle: one full training step on the CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Composition of the reference's hot loop:
  models.py:61-66   training_step: y_pred = forward(x); loss = F.mse_loss(y, y_pred)
  loss.backward()   autograd
  models.py:68-70   Adam(self.parameters(), lr).step()
for the two model families of BASELINE.json: hash-grid + ReLU tiny-MLP and SIREN.
Also the `cpu_baseline` ("port") leg of bench.py.
"""
from typing import List, Optional, Sequence

import torch

from . import hashgrid, mlp


class HashMlpModel:
    """encoder (encoding.py:131-191 / :273-336) + ReLU MLP (models.py:46-56)."""

    def __init__(self, dim, n_levels, n_features, log2_hashmap_size, base_resolution,
                 finest_resolution, hidden: Sequence[int], dim_out=1, seed=0,
                 final_activation=False, table_scale=1e-4):
        self.resolutions, self.sizes = hashgrid.resolutions_for(
            dim, n_levels, log2_hashmap_size, base_resolution, finest_resolution)
        self.tables = hashgrid.init_tables(self.sizes, n_features, seed, table_scale)
        dims = [n_levels * n_features] + list(hidden) + [dim_out]
        self.mlp = mlp.linear_init(dims, seed + 1)
        self.final_activation = final_activation

    def parameters(self) -> List[torch.Tensor]:
        out = list(self.tables)
        for w, b in self.mlp:
            out += [w, b]
        return out

    def forward(self, x):
        z = hashgrid.encode(x, self.tables, self.resolutions)
        return mlp.relu_mlp_forward(z, self.mlp, self.final_activation)


class SirenModel:
    """SirenNet (models.py:160-233)."""

    def __init__(self, dim_in, dim_hidden, dim_out, n_layers, w0=30.0, w0_initial=30.0,
                 seed=0):
        self.params = mlp.siren_init(dim_in, dim_hidden, dim_out, n_layers, seed, w0=w0)
        self.w0, self.w0_initial = w0, w0_initial

    def parameters(self) -> List[torch.Tensor]:
        out = []
        for w, b in self.params:
            out += [w, b]
        return out

    def forward(self, x):
        return mlp.siren_forward(x, self.params, self.w0, self.w0_initial)


def loss_and_grads(model, x: torch.Tensor, y: torch.Tensor):
    """loss, y_pred and d loss / d parameter for every parameter."""
    ps = model.parameters()
    for p in ps:
        p.requires_grad_(True)
        p.grad = None
    pred = model.forward(x)
    loss = mlp.mse_loss(pred, y)
    loss.backward()
    grads = [p.grad for p in ps]
    for p in ps:
        p.requires_grad_(False)
    return loss.detach(), pred.detach(), grads


def train_steps(model, batches, lr: float, opt: Optional[mlp.Adam] = None):
    """Run one Adam step per (x, y) batch; returns (losses, optimiser)."""
    if opt is None:
        opt = mlp.Adam(model.parameters(), lr=lr)
    losses = []
    for x, y in batches:
        loss, _, grads = loss_and_grads(model, x, y)
        opt.step(grads)
        for p in model.parameters():
            p.grad = None
        losses.append(float(loss))
    return losses, opt


def as_double(model):
    """A float64 copy of a HashMlpModel / SirenModel: the YARDSTICK of the parity tests (the same op
    sequence evaluated in double on the same float32 inputs), against which two float32 evaluations --
    the HIP path's and the oracle's / the reference's -- are compared with each other."""
    import copy
    m = copy.copy(model)
    if isinstance(model, HashMlpModel):
        m.tables = [t.detach().double().clone() for t in model.tables]
        m.mlp = [(w.detach().double().clone(), b.detach().double().clone()) for w, b in model.mlp]
    else:
        m.params = [(w.detach().double().clone(), None if b is None else b.detach().double().clone())
                    for w, b in model.params]
    return m


def loss_and_grads_chunked(model, x: torch.Tensor, y: torch.Tensor, chunk: int):
    """loss_and_grads over row chunks (the mean over all rows is the weighted sum of the chunks' means):
    bounds the memory of a float64 evaluation at 2^20 rows."""
    n = x.shape[0]
    total, grads = 0.0, None
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        loss, _, g = loss_and_grads(model, x[lo:hi], y[lo:hi])
        wgt = (hi - lo) / n
        total += float(loss) * wgt
        g = [None if t is None else t * wgt for t in g]
        grads = g if grads is None else [a if b is None else (b if a is None else a + b)
                                         for a, b in zip(grads, g)]
    return total, grads
