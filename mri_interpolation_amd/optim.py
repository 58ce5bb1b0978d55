"""Adam over ONE flat parameter buffer, stepped by the fused gfx950 kernel.

Same update rule, defaults and call surface as `torch.optim.Adam(params, lr)` which the
reference builds in `configure_optimizers` (reference models.py:68-70): betas (0.9, 0.999),
eps 1e-8, no weight decay, no amsgrad.

MI355X-first: every parameter (hash tables + MLP weights) is re-pointed into one contiguous
buffer, gradients into a second one, so that a step is ONE kernel launch streaming
28 B/parameter, and the data-parallel gradient reduction is ONE RCCL collective.
"""
from typing import Iterable, List, Optional

import torch

from . import ops

_ALIGN = 4  # floats (16 B): keeps every parameter's base aligned for float4 access
_PAD = 256  # floats: allocation granule, see FlatBuffers.capacity


class FlatBuffers:
    """Parameters, gradients and Adam moments of a model as four flat float32 tensors."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("mri_interpolation_amd.optim: parameters must live on the GPU "
                               "(move the model with .cuda() first; there is no CPU fallback)")
        self.offsets, total = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("all parameters must be float32 on one device")
            self.offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        # the allocation is padded (zeros: Adam leaves them zero) so that it divides evenly,
        # in float4 units, over any power-of-two number of ranks up to 64 (reduce-scatter shards)
        self.capacity = (total + _PAD - 1) // _PAD * _PAD
        self._param_all = torch.zeros(self.capacity, device=dev)
        self._grad_all = torch.zeros(self.capacity, device=dev)
        self._exp_avg_all = torch.zeros(self.capacity, device=dev)
        self._exp_avg_sq_all = torch.zeros(self.capacity, device=dev)
        self.param, self.grad = self._param_all[:total], self._grad_all[:total]
        self.exp_avg, self.exp_avg_sq = self._exp_avg_all[:total], self._exp_avg_sq_all[:total]
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                n = p.numel()
                self.param[off:off + n].copy_(p.detach().reshape(-1))
                old_grad = p.grad
                p.data = self.param[off:off + n].view(p.shape)
                if old_grad is not None:
                    self.grad[off:off + n].copy_(old_grad.reshape(-1))
                p.grad = self.grad[off:off + n].view(p.shape)

    def grad_view(self, p: torch.nn.Parameter) -> torch.Tensor:
        i = next(k for k, q in enumerate(self.params) if q is p)
        off = self.offsets[i]
        return self.grad[off:off + p.numel()].view(p.shape)


class Adam:
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self._params = [p for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.flat: Optional[FlatBuffers] = None
        self.grad_scale = 1.0  # 1/world_size after an all-reduce(sum)
        # True once step_shard() has run with world > 1: the local moment buffers then hold this rank's
        # shard only (checkpoint.save refuses until parallel.gather_optimizer_state has made them whole)
        self.sharded = False
        self.param_groups = [dict(params=self._params, lr=lr, betas=betas, eps=eps)]

    def flatten(self) -> FlatBuffers:
        if self.flat is None:
            self.flat = FlatBuffers(self._params)
        return self.flat

    def zero_grad(self, set_to_none: bool = False):
        if self.flat is not None:
            self.flat.grad.zero_()
        else:
            for p in self._params:
                p.grad = None

    @torch.no_grad()
    def begin_step(self):
        """Open step number step_count + 1; every element is then updated exactly once, by
        step() in one launch or by step_range() piecewise (elementwise rule: same result)."""
        flat = self.flatten()
        for p, off in zip(flat.params, flat.offsets):
            # autograd may have replaced the view (first backward before flatten()):
            g = p.grad
            if g is not None and g.data_ptr() != flat.grad.data_ptr() + 4 * off:
                flat.grad[off:off + p.numel()].copy_(g.reshape(-1))
                p.grad = flat.grad[off:off + p.numel()].view(p.shape)
        self.step_count += 1

    @torch.no_grad()
    def step_range(self, lo: int, hi: int):
        """Update flat elements [lo, hi) for the step opened by begin_step()."""
        flat = self.flatten()
        lr = self.param_groups[0]["lr"]
        ops.adam_step(flat.param[lo:hi], flat.grad[lo:hi], flat.exp_avg[lo:hi],
                      flat.exp_avg_sq[lo:hi], lr, self.betas[0], self.betas[1], self.eps,
                      self.step_count, self.grad_scale)

    def step(self):
        self.begin_step()
        self.step_range(0, self.flatten().numel)

    @torch.no_grad()
    def step_shard(self, lo: int, hi: int):
        """step_range over the PADDED buffers (reduce-scatter shards are cut there)."""
        flat = self.flatten()
        lr = self.param_groups[0]["lr"]
        ops.adam_step(flat._param_all[lo:hi], flat._grad_all[lo:hi], flat._exp_avg_all[lo:hi],
                      flat._exp_avg_sq_all[lo:hi], lr, self.betas[0], self.betas[1], self.eps,
                      self.step_count, self.grad_scale)

    def state_dict(self):
        f = self.flatten()
        return dict(step=self.step_count, lr=self.lr, betas=self.betas, eps=self.eps,
                    exp_avg=f.exp_avg.clone(), exp_avg_sq=f.exp_avg_sq.clone())

    def load_state_dict(self, sd):
        f = self.flatten()
        self.step_count = int(sd["step"])
        f.exp_avg.copy_(sd["exp_avg"])
        f.exp_avg_sq.copy_(sd["exp_avg_sq"])
