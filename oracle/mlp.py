"""Oracle: SIREN / ReLU tiny-MLP, MSE loss, Adam (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference `models.py`:
  BaseMLP layer stack + training contract   models.py:46-74
  Sine / SirenLayer (init + forward)        models.py:108-156
  SirenNet.forward                          models.py:199-233
  Modulator / ModulatedSirenNet.forward     models.py:236-260, 311-322
  HashMLP decoder blocks (as intended, Q1)  models.py:712-744
Adam restates torch.optim.Adam (single-tensor, default betas/eps, no weight
decay, no amsgrad) which `configure_optimizers` builds at models.py:68-70.

Parameters are plain lists of (weight (out,in), bias (out,)) float32 tensors so
that the same arrays can be handed to the HIP path in tests.
"""
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = List[Tuple[torch.Tensor, Optional[torch.Tensor]]]


# --------------------------------------------------------------------------- init
def siren_init(dim_in: int, dim_hidden: int, dim_out: int, n_layers: int, seed: int,
               w0: float = 30.0, sigma: float = 6.0, use_bias: bool = True) -> Params:
    """Weight ranges of models.py:144-151: U(+-1/dim_in) for the first layer,
    U(+-sqrt(sigma/dim_in)/w0) for the others (last_layer included, built with
    is_first=False at models.py:221-228).  Values from oracle.detrand."""
    from . import detrand
    out: Params = []
    for i in range(n_layers + 1):
        fan_in = dim_in if i == 0 else dim_hidden
        fan_out = dim_out if i == n_layers else dim_hidden
        bound = (1.0 / fan_in) if i == 0 else (math.sqrt(sigma / fan_in) / w0)
        w = detrand.uniform(fan_out * fan_in, seed * 100 + 2 * i, -bound, bound)
        b = detrand.uniform(fan_out, seed * 100 + 2 * i + 1, -bound, bound)
        out.append((torch.from_numpy(w.reshape(fan_out, fan_in).copy()),
                    torch.from_numpy(b.copy()) if use_bias else None))
    return out


def linear_init(dims: Sequence[int], seed: int) -> Params:
    """nn.Linear-style ranges U(+-1/sqrt(fan_in)) (what models.py:49-53 gets by
    default); values from oracle.detrand."""
    from . import detrand
    out: Params = []
    for i in range(len(dims) - 1):
        bound = 1.0 / math.sqrt(dims[i])
        w = detrand.uniform(dims[i + 1] * dims[i], seed * 100 + 2 * i, -bound, bound)
        b = detrand.uniform(dims[i + 1], seed * 100 + 2 * i + 1, -bound, bound)
        out.append((torch.from_numpy(w.reshape(dims[i + 1], dims[i]).copy()),
                    torch.from_numpy(b.copy())))
    return out


# --------------------------------------------------------------------------- forward
def siren_forward(x: torch.Tensor, params: Params, w0: float = 30.0,
                  w0_initial: float = 30.0) -> torch.Tensor:
    """models.py:230-233: n hidden SirenLayers sin(w0 * (x W^T + b)), then a
    linear last layer (Identity activation, models.py:218-228)."""
    n_hidden = len(params) - 1
    for i, (w, b) in enumerate(params[:n_hidden]):
        x = torch.sin((w0_initial if i == 0 else w0) * F.linear(x, w, b))
    w, b = params[-1]
    return F.linear(x, w, b)


def modulator_forward(z: torch.Tensor, params: Params) -> List[torch.Tensor]:
    """models.py:251-260: hidden_i = relu(Linear_i(x)); the next layer sees
    cat(hidden_i, z) (hidden first, latent second)."""
    x, hiddens = z, []
    for w, b in params:
        x = torch.relu(F.linear(x, w, b))
        hiddens.append(x)
        x = torch.cat((x, z), dim=1)
    return hiddens


def modulated_siren_forward(x: torch.Tensor, siren: Params, modulator: Params, w0: float = 30.0,
                            w0_initial: float = 30.0) -> torch.Tensor:
    """models.py:311-322: every hidden SirenLayer output is multiplied elementwise by the
    modulator's hidden state of the same depth (computed from the same coordinates), then
    the linear last layer."""
    mods = modulator_forward(x, modulator)
    n_hidden = len(siren) - 1
    for i, (w, b) in enumerate(siren[:n_hidden]):
        x = torch.sin((w0_initial if i == 0 else w0) * F.linear(x, w, b)) * mods[i]
    w, b = siren[-1]
    return F.linear(x, w, b)


def modulator_init(dim_in: int, dim_hidden: int, n_layers: int, seed: int) -> Params:
    """nn.Linear default ranges for models.py:245-249 (input widths dim_in, then
    dim_hidden + dim_in); values from oracle.detrand."""
    from . import detrand
    out: Params = []
    for i in range(n_layers):
        fan_in = dim_in if i == 0 else dim_hidden + dim_in
        bound = 1.0 / math.sqrt(fan_in)
        w = detrand.uniform(dim_hidden * fan_in, seed * 100 + 2 * i, -bound, bound)
        b = detrand.uniform(dim_hidden, seed * 100 + 2 * i + 1, -bound, bound)
        out.append((torch.from_numpy(w.reshape(dim_hidden, fan_in).copy()),
                    torch.from_numpy(b.copy())))
    return out


def relu_mlp_forward(x: torch.Tensor, params: Params, final_activation: bool) -> torch.Tensor:
    """models.py:46-56: Linear -> ReLU per layer.  `final_activation=True` keeps the
    ReLU after the last Linear exactly as BaseMLP.layers does; False is the
    tiny-MLP of BASELINE configs 2/4 (linear output, hash_config.json:22-27)."""
    last = len(params) - 1
    for i, (w, b) in enumerate(params):
        x = F.linear(x, w, b)
        if i < last or final_activation:
            x = torch.relu(x)
    return x


def gelu_mlp_forward(x: torch.Tensor, params: Params, final_activation: bool) -> torch.Tensor:
    """Notebook cell-37 decoder: Linear -> GELU(erf) blocks."""
    last = len(params) - 1
    for i, (w, b) in enumerate(params):
        x = F.linear(x, w, b)
        if i < last or final_activation:
            x = F.gelu(x)
    return x


def hashmlp_decoder_forward(z: torch.Tensor, params: Params, bn: list, training: bool,
                            eps: float = 1e-5, momentum: float = 0.1) -> torch.Tensor:
    """HashMLP decoder as intended: for each block
    Linear -> BatchNorm1d -> GELU -> Dropout(0)  (models.py:718-738), applied in
    sequence.  `bn[i]` = dict(weight, bias, running_mean, running_var)."""
    for (w, b), s in zip(params, bn):
        z = F.linear(z, w, b)
        z = F.batch_norm(z, s["running_mean"], s["running_var"], s["weight"], s["bias"],
                         training, momentum, eps)
        z = F.gelu(z)
    return z


# --------------------------------------------------------------------------- loss / optimiser
def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """models.py:64: F.mse_loss(y, y_pred), mean over every element."""
    return F.mse_loss(target, pred)


def psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """10 log10(1/MSE) for data in [0, 1] (what skimage's PSNR gives for
    non-negative float images; legacy_code/hash_experimentation.py:445-450)."""
    return float(10.0 * torch.log10(1.0 / F.mse_loss(pred, target)))


class Adam:
    """torch.optim.Adam defaults (models.py:68-70), written out per tensor.

    Same arithmetic order as torch's single-tensor path:
      m <- lerp(m, g, 1-b1);  v <- v*b2 + (1-b2)*g*g
      denom = sqrt(v)/sqrt(1-b2^t) + eps;  p <- p - (lr/(1-b1^t)) * m/denom
    """

    def __init__(self, params: Sequence[torch.Tensor], lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = list(params)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads: Sequence[Optional[torch.Tensor]]):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.t
        bc2 = 1.0 - b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if g is None:
                continue
            m.lerp_(g, 1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(m, denom, value=-step_size)
