"""The launcher's GPU-less path: BASELINE config 1 ("SIREN on one 2-D slice via launcher.py, PyTorch CPU").

The reference trains on the CPU when no GPU is visible (reference launcher.py:157:
`accelerator='gpu' if torch.cuda.is_available() else 'cpu'`).  The MI355X hot path has no CPU form -- every op
of `ops.py` raises on CPU tensors and a missing `libmri_inr.so` is an error -- so this module is NOT a fallback of
that path: it is the launcher's plumbing mode for machines without a GPU, reachable only through
`launcher.py --accelerator cpu` (or `auto` with no GPU visible), for `SirenNet` alone, in plain PyTorch:

  * the model is `models.SirenNet` itself (same constructor, initialisation, parameters and state-dict keys, so
    checkpoints move between the two paths); only its evaluation is restated here with `F.linear` + `torch.sin`
    (reference models.py:153-156, 230-233);
  * data: coordinates = meshgrid(linspace(-1, 1, s)) flattened C-order, intensities min-max normalised to
    [-1, 1] (reference datamodules.py:140-166 with norm_siren), shuffled batches from a seeded `torch.randperm`
    per epoch (DataLoader(shuffle=True));
  * training_step = F.mse_loss, optimiser = torch.optim.Adam(lr) (reference models.py:61-70).

Nothing here is timed by bench.py or compared by the `-m gpu` parity tests; tests/test_cpu_path.py runs it on the
352 x 352 slice of the sample volume.
"""
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import models


def grid_coords(shape, norm_siren: bool = True) -> torch.Tensor:
    """(N, D) coordinates of a volume of `shape`, C-order (last axis fastest), reference datamodules.py:140-166."""
    lo = -1.0 if norm_siren else 0.0
    axes = [torch.linspace(lo, 1.0, int(s)) for s in shape]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(shape))


def normalised_pixels(volume: np.ndarray, norm_siren: bool = True) -> torch.Tensor:
    pix = torch.as_tensor(np.ascontiguousarray(volume), dtype=torch.float32).flatten()
    pix = (pix - pix.min()) / (pix.max() - pix.min())
    return (pix * 2 - 1 if norm_siren else pix).unsqueeze(-1)


def siren_forward(net: "models.SirenNet", x: torch.Tensor) -> torch.Tensor:
    """SirenNet.forward on CPU tensors: sin(w0 (x W^T + b)) per layer, then the linear last layer."""
    for i, layer in enumerate(net.layers):
        w0 = net.w0_initial if i == 0 else net.w0
        x = torch.sin(w0 * F.linear(x, layer.weight, layer.bias))
    return F.linear(x, net.last_layer.weight, net.last_layer.bias)


def fit(net: "models.SirenNet", coords: torch.Tensor, pixels: torch.Tensor, batch_size: int, epochs: int,
        seed: int = 1337, max_steps: int = -1, log_every: int = 0) -> Tuple[List[float], int]:
    """Adam on F.mse_loss over shuffled batches; returns (loss per step, steps)."""
    opt = torch.optim.Adam(net.parameters(), lr=net.lr)
    gen = torch.Generator().manual_seed(seed)
    n, losses, step = coords.shape[0], [], 0
    for _ in range(epochs):
        perm = torch.randperm(n, generator=gen)
        for lo in range(0, n, batch_size):
            idx = perm[lo:lo + batch_size]
            loss = F.mse_loss(siren_forward(net, coords[idx]), pixels[idx])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            step += 1
            if log_every and step % log_every == 0:
                print(f"step {step}: train_loss {losses[-1]:.6f}")
            if 0 < max_steps <= step:
                return losses, step
    return losses, step


@torch.no_grad()
def predict(net: "models.SirenNet", coords: torch.Tensor, batch_size: int) -> torch.Tensor:
    return torch.cat([siren_forward(net, coords[lo:lo + batch_size]) for lo in range(0, coords.shape[0], batch_size)])


def psnr(pred: torch.Tensor, truth: torch.Tensor) -> float:
    mse = float(((pred.double() - truth.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * float(np.log10(1.0 / mse))
