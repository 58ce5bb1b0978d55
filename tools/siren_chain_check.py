#!/usr/bin/env python3
"""Fused SIREN chain kernels against the layer-wise path (GPU box): values and timings."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from mri_interpolation_amd import _lib, models, ops, trainer

_lib.load()
torch.manual_seed(0)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def pair(dim_in, n_layers, hidden=256):
    net = models.SirenNet(dim_in, hidden, 1, n_layers).cuda()
    nets = [net, copy.deepcopy(net)]
    steps = [trainer.FusedStep(m, m.configure_optimizers()) for m in nets]
    assert steps[0].use_chain and steps[1].use_chain
    steps[1].use_chain = False
    return nets, steps


worst = 0.0
for hidden, dim_in, n_layers in ((256, 3, 5), (256, 2, 2), (128, 3, 4), (128, 4, 1), (64, 2, 3), (64, 3, 6),
                                  (32, 1, 3), (32, 3, 2)):
    nets, steps = pair(dim_in, n_layers, hidden)
    for n in (1, 31, 64, 65, 192, 1000, 70001):
        x = torch.rand(n, dim_in, device="cuda") * 2 - 1
        y = torch.rand(n, 1, device="cuda") * 2 - 1
        outs = []
        for st in steps:
            pred, ws = st.forward(x, train=True)
            st.backward(x, y, ws)
            outs.append((pred.clone(), float(st.loss), ws))
        errs = dict(pred=rel(outs[0][0], outs[1][0]), loss=abs(outs[0][1] - outs[1][1]) / abs(outs[1][1]))
        with torch.no_grad():
            errs["infer"] = rel(steps[0].forward(x, train=False)[0], outs[1][0])
        for (name, p0), (_, p1) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
            errs[name] = rel(p0.grad, p1.grad)
        bad = {k: v for k, v in errs.items() if not v < 2e-5}
        worst = max(worst, max(errs.values()))
        print(f"hidden {hidden} dim_in {dim_in} layers {n_layers} n {n}: max err {max(errs.values()):.2e} "
              f"{'ok' if not bad else 'FAIL ' + str(bad)}", flush=True)
        assert not bad
    # whole steps stay together
    x = torch.rand(5000, dim_in, device="cuda") * 2 - 1
    y = torch.sin(3 * x[:, :1])
    for _ in range(5):
        la, lb = float(steps[0].train_step(x, y)), float(steps[1].train_step(x, y))
    print(f"   5 Adam steps: loss {la:.6e} vs {lb:.6e}, params {rel(steps[0].flat.param, steps[1].flat.param):.2e}")
print(f"worst error {worst:.2e}")

n = 1 << 20
nets, steps = pair(3, 5)
x = torch.rand(n, 3, device="cuda") * 2 - 1
y = torch.rand(n, 1, device="cuda")
flops = 2.0 * n * (3 * 256 + 4 * 256 * 256 + 256)
for name, st in (("chain", steps[0]), ("layer-wise", steps[1])):
    ws = st.forward(x, train=True)[1]
    t_f = timed(lambda: st.forward(x, train=True))
    t_b = timed(lambda: st.backward(x, y, ws))
    t_i = timed(lambda: st.forward(x, train=False))
    t_s = timed(lambda: st.train_step(x, y))
    print(f"{name:11s} fwd {t_f:6.3f} ms ({flops / t_f / 1e9:5.1f} TF)  bwd {t_b:6.3f} ms "
          f"({2 * flops / t_b / 1e9:5.1f} TF)  infer {t_i:6.3f} ms  step {t_s:6.3f} ms "
          f"({3 * flops / t_s / 1e9:5.1f} TF, {n / t_s / 1e3:5.1f} M coord/s)", flush=True)
