"""Checkpoints in the layout PyTorch Lightning writes, so that the reference can load them.

The reference resumes through `config.model_cls.load_from_checkpoint(path, **ctor_kwargs)` (reference
launcher.py:97-117, strict) and trains under `pl.Trainer` whose default callback writes
`lightning_logs/version_N/checkpoints/epoch=E-step=S.ckpt`.  What Lightning reads from such a file:
`"pytorch-lightning_version"` (its migration step indexes it unconditionally), `"state_dict"` (strict:
every key of the module, including the dead `layers.*` stack `BaseMLP.__init__` builds inside the
reference's `HashMLP`), and -- for `Trainer.fit(ckpt_path=...)` -- `"optimizer_states"`
(one `torch.optim.Adam.state_dict()` per optimiser, parameters numbered in `model.parameters()` order of
the REFERENCE module), `"lr_schedulers"`, `"epoch"`, `"global_step"`.

This module writes exactly that from the MI355X-side classes (whose hash tables are ONE flat parameter and
whose Adam moments live in one flat buffer) and reads it back.  `load()` restores the PARAMETERS only by
default -- what the reference's resume does: `load_from_checkpoint(path, ..., lr=config.lr)` followed by
`trainer.fit(model, loader)` WITHOUT `ckpt_path` (reference launcher.py:97-165), i.e. a fresh Adam at the
command line's learning rate.  `resume_optimizer=True` (launcher `--resume_optimizer`) is the equivalent
of Lightning's `fit(ckpt_path=...)`: moments and step count come back too; the learning rate of the file
replaces the configured one only with `restore_lr=True`.
Host logic only: no kernel is involved.
"""
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

LIGHTNING_VERSION = "2.0.0"  # the era of the reference's notebook metadata (PL 1.7 - 2.0)


def _dead_base_stack(model) -> "OrderedDict[str, torch.Tensor]":
    """`layers.*` of the reference's HashMLP: `BaseMLP.__init__(*args, **kwargs)` runs with its own
    defaults dim_in 2, dim_hidden 128, dim_out 1 and the launcher's n_layers (reference models.py:25-56,
    677), registering Linear / ReLU pairs that forward() never uses.  Values: nn.Linear's default init
    (what Lightning would have saved: never trained)."""
    out = OrderedDict()
    n_layers = int(getattr(model, "n_layers", 0))
    # a forked, seeded generator: saving neither advances the caller's RNG nor depends on it
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(1337)
        for i in range(n_layers):
            lin = nn.Linear(2 if i == 0 else 128, 1 if i == n_layers - 1 else 128)
            out[f"layers.{2 * i}.weight"] = lin.weight.detach().clone()
            out[f"layers.{2 * i}.bias"] = lin.bias.detach().clone()
    return out


def reference_state_dict(model) -> "OrderedDict[str, torch.Tensor]":
    """model.state_dict() on the CPU with the keys, in the order, the reference module of the same class
    holds (HashMLP: the dead `layers.*` entries first, as `BaseMLP.__init__` registers them first)."""
    sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items())
    if type(model).__name__ == "HashMLP":
        sd = OrderedDict(list(_dead_base_stack(model).items()) + list(sd.items()))
    return sd


def _reference_parameter_names(model):
    """Parameter names in the REFERENCE module's `parameters()` order (= its state-dict order minus
    buffers); the flat table of this package counts as one parameter per level."""
    buffers = {n for n, _ in model.named_buffers()}
    return [k for k in reference_state_dict(model) if k not in buffers
            and not k.endswith("num_batches_tracked")]


def _moments_by_name(model, optimizer) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """{state-dict key: (exp_avg, exp_avg_sq)} from the flat Adam buffers, the table's per level."""
    if getattr(optimizer, "sharded", False):
        # dp_mode "reduce_scatter": a rank steps -- and holds the moments of -- its 1/world shard only
        # (optim.Adam.step_shard); the other shards of the local buffers are stale
        raise RuntimeError("checkpoint.save: the optimiser state is sharded over the ranks (dp_mode "
                           "'reduce_scatter'); gather it first (parallel.gather_optimizer_state) "
                           "or save with dp_mode 'all_reduce'")
    flat = optimizer.flatten()
    out = {}
    own = {id(p): n for n, p in model.named_parameters()}
    for p, off in zip(flat.params, flat.offsets):
        name, n = own[id(p)], p.numel()
        m = flat.exp_avg[off:off + n].view(p.shape).detach().cpu().clone()
        v = flat.exp_avg_sq[off:off + n].view(p.shape).detach().cpu().clone()
        if name.endswith("encoder.table"):
            enc = model.encoder
            for l in range(enc.n_levels):
                lo, hi = enc._row_span(l)
                out[name[:-len("table")] + f"levels.{l}.embedding.weight"] = (m[lo:hi].clone(), v[lo:hi].clone())
        else:
            out[name] = (m, v)
    return out


def adam_state_dict(model, step: int, lr: float, betas, eps: float,
                    moments: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]]) -> dict:
    """`torch.optim.Adam.state_dict()` as the reference's optimiser (models.py:68-70) would hold it:
    parameters numbered in the reference module's order; entries only for parameters that ever had a
    gradient (the dead `layers.*` stack has none, torch keeps no state for it)."""
    names = _reference_parameter_names(model)
    state = {}
    if moments is not None and step > 0:
        for i, name in enumerate(names):
            if name in moments:
                m, v = moments[name]
                state[i] = dict(step=torch.tensor(float(step)), exp_avg=m, exp_avg_sq=v)
    group = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None,
                 params=list(range(len(names))))
    return dict(state=state, param_groups=[group])


def lightning_checkpoint(model, optimizer=None, epoch: int = 0, global_step: int = 0,
                         moments=None, step: Optional[int] = None) -> dict:
    """The dictionary `pl.Trainer`'s ModelCheckpoint would have written for the reference module.
    `optimizer`: this package's optim.Adam (moments are read from its flat buffers); tests without a GPU
    pass `moments` / `step` directly."""
    lr, betas, eps = getattr(model, "lr", 1e-4), (0.9, 0.999), 1e-8
    if optimizer is not None:
        lr, betas, eps = optimizer.param_groups[0]["lr"], optimizer.betas, optimizer.eps
        if optimizer.flat is not None:
            moments = _moments_by_name(model, optimizer)
        step = optimizer.step_count
    return {
        "epoch": int(epoch), "global_step": int(global_step),
        "pytorch-lightning_version": LIGHTNING_VERSION,
        "state_dict": reference_state_dict(model),
        "loops": None, "callbacks": {},
        "optimizer_states": [adam_state_dict(model, int(step or 0), lr, betas, eps, moments)],
        "lr_schedulers": [],
    }


def save(path: str, model, optimizer=None, epoch: int = 0, global_step: int = 0):
    ckpt = lightning_checkpoint(model, optimizer, epoch, global_step)
    if ckpt["loops"] is None:
        del ckpt["loops"]  # Lightning restores loops only when the key is there
    torch.save(ckpt, path)
    return ckpt


def load(path: str, model, optimizer=None, map_location="cpu", resume_optimizer: bool = False,
         restore_lr: bool = False, allow_pickle: bool = False) -> dict:
    """Load the parameters from a checkpoint written by save() or by Lightning for the reference module;
    returns the checkpoint.  `layers.*` of a reference HashMLP checkpoint is dropped by the model's own
    loader (models.py).
    resume_optimizer: also restore Adam's moments and step count into `optimizer` (this package's
        optim.Adam) -- Lightning's `fit(ckpt_path=)`; the default leaves the optimiser as constructed,
        which is what the reference's launcher does (launcher.py:97-165).
    restore_lr: with resume_optimizer, take the learning rate of the file instead of the configured one.
    allow_pickle: what save() writes needs nothing beyond tensors and containers and is read with
        `weights_only=True`; a Lightning checkpoint holding callback / hyper-parameter objects needs the
        unrestricted unpickler, which executes code from the file: opt in explicitly for files you trust."""
    ckpt = torch.load(path, map_location=map_location, weights_only=not allow_pickle)
    model.load_state_dict(ckpt.get("state_dict", ckpt))
    states = ckpt.get("optimizer_states") or []
    if resume_optimizer and optimizer is None:
        raise ValueError("resume_optimizer=True needs the optimiser to restore into")
    if resume_optimizer and states and states[0].get("state"):
        names = _reference_parameter_names(model)
        by_name = {names[int(i)]: s for i, s in states[0]["state"].items()}
        flat = optimizer.flatten()
        own = {id(p): n for n, p in model.named_parameters()}
        steps = set()
        with torch.no_grad():
            for p, off in zip(flat.params, flat.offsets):
                name, n = own[id(p)], p.numel()
                m = flat.exp_avg[off:off + n].view(p.shape)
                v = flat.exp_avg_sq[off:off + n].view(p.shape)
                if name.endswith("encoder.table"):
                    enc = model.encoder
                    for l in range(enc.n_levels):
                        s = by_name.get(name[:-len("table")] + f"levels.{l}.embedding.weight")
                        if s is not None:
                            lo, hi = enc._row_span(l)
                            m[lo:hi].copy_(s["exp_avg"])
                            v[lo:hi].copy_(s["exp_avg_sq"])
                            steps.add(int(float(s["step"])))
                elif name in by_name:
                    s = by_name[name]
                    m.copy_(s["exp_avg"])
                    v.copy_(s["exp_avg_sq"])
                    steps.add(int(float(s["step"])))
        if len(steps) > 1:
            raise ValueError(f"checkpoint holds different Adam step counts per parameter ({sorted(steps)}): "
                             "the flat optimiser keeps one")
        if steps:
            optimizer.step_count = steps.pop()
        if restore_lr:
            group = states[0]["param_groups"][0]
            optimizer.param_groups[0]["lr"] = group.get("lr", optimizer.param_groups[0]["lr"])
    return ckpt
