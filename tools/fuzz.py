#!/usr/bin/env python3
"""Seeded random sweeps of the HIP path against the CPU oracle (run on the MI355X box).

    python tools/fuzz.py [--encoders N] [--layers N] [--steps N] [--sirens N] [--first SEED]

  encoders  random hash grids (D 1-4, F 1-8, 1-6 levels, T 2^6-2^17, isotropic and per-axis
            resolutions, batch 1-20,000): forward and the three table-gradient methods
  layers    random Linear(+activation) shapes (batch 1-3000, widths 1-300, every activation,
            with / without bias): forward, backward-data, backward-weight
  steps     whole hash + ReLU tiny-MLP training steps on the fused kernel chain: loss, every
            gradient, parameters after Adam
  sirens    whole SIREN steps on the fused chain
Tolerance: 1e-5 per tensor (max-abs error / max-abs reference and relative L2), hash slots exact.

A whole step may legitimately miss that tolerance: when ONE hidden pre-activation of ONE
coordinate is ~1e-9 (seven orders below typical), the two f32 evaluations -- oracle and kernel,
different summation orders -- can land on different sides of the ReLU kink, and that coordinate's
contribution to every gradient differs by O(1).  `explain_relu_kink` proves that this is all
there is to such a miss: it re-evaluates the oracle's gradient with that single gate flipped and
requires the kernel's gradients to match THAT to 1e-5.  A failing step that is not explained
this way is reported as a defect (exit code 1).  Two such seeds are committed as regression
cases (tests/test_gpu_round2.py::test_relu_kink_seeds_are_fully_explained).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import hashgrid as ohash, mlp as omlp, train as otrain  # checker only

TOL = 1e-5


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not b.size:
        return 0.0
    return max(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30),
               np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ------------------------------------------------------------------------------ case generators
def encoder_case(seed):
    r = np.random.default_rng(seed)
    dim = int(r.integers(1, 5))
    feats = int(r.choice([1, 2, 4, 8]))
    levels = int(r.integers(1, 7))
    log2t = int(r.integers(6, 18))
    if r.random() < 0.5:
        base = int(r.integers(2, 17))
        finest = float(base * r.uniform(1.0, 16.0))
    else:
        base = tuple(int(v) for v in r.integers(2, 17, dim))
        finest = tuple(float(b * r.uniform(1.0, 16.0)) for b in base)
    n = int(r.choice([1, 63, 64, 65, 257, int(r.integers(1, 20001))]))
    return dict(dim=dim, feats=feats, levels=levels, log2t=log2t, base=base, finest=finest, n=n)


def step_case(seed):
    r = np.random.default_rng(seed)
    dim = int(r.integers(2, 5))
    feats = int(r.choice([1, 2, 4]))
    levels = int(r.integers(1, min(6, 32 // feats) + 1))
    hidden = int(r.choice([64, 128]))
    return dict(dim=dim, feats=feats, levels=levels, log2t=int(r.integers(6, 15)),
                base=int(r.integers(2, 17)), growth=float(r.uniform(1.0, 8.0)), hidden=hidden,
                n=int(r.choice([1000, 4097, int(r.integers(1, 4101))])), seed=seed)


# ------------------------------------------------------------------------------ encoder sweep
def run_encoder(seed, pkg):
    c = encoder_case(seed)
    res, sizes = ohash.resolutions_for(c["dim"], c["levels"], c["log2t"], c["base"], c["finest"])
    tables = ohash.init_tables(sizes, c["feats"], seed, 0.5)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(c["n"], c["dim"], generator=g)
    x[: min(4, c["n"])] = torch.tensor([0.0, 1.0, 0.999999, -0.3])[: min(4, c["n"]), None]
    d_out = torch.rand(c["n"], c["levels"] * c["feats"], generator=g) * 2 - 1
    for t in tables:
        t.requires_grad_(True)
    want = ohash.encode(x, tables, res)
    want.backward(d_out)
    cls = pkg.encoding.MultiResHashGrid if isinstance(c["base"], int) else pkg.encoding.MultiResHashGridV2
    enc = cls(c["dim"], n_levels=c["levels"], n_features_per_level=c["feats"],
              log2_hashmap_size=c["log2t"], base_resolution=c["base"], finest_resolution=c["finest"])
    assert enc.sizes == sizes, (enc.sizes, sizes)
    with torch.no_grad():
        enc.table.copy_(torch.cat([t.detach() for t in tables]))
    enc = enc.cuda()
    errs = {}
    with torch.no_grad():
        errs["forward"] = rel(enc(x.cuda()).cpu().numpy(), want.detach().numpy())
    ref_g = torch.cat([t.grad for t in tables]).numpy()
    for method in (0, 1, 2):
        got = torch.zeros_like(enc.table.data)
        pkg.ops.hashgrid_backward(enc.desc, x.cuda(), d_out.cuda(), got, method=method)
        got = got.cpu().numpy()
        # integer work: nothing may land outside the oracle's slots, no slot of any weight may be
        # lost (contributions below ~2^-40 max|g| may round to zero in the fixed-point sum)
        hit, ref_hit = np.abs(got).sum(1) != 0, np.abs(ref_g).sum(1)
        stray = (hit & (ref_hit == 0)).sum() + (~hit & (ref_hit > 1e-9 * ref_hit.max())).sum()
        errs[f"backward{method}"] = max(rel(got, ref_g), float(stray > 0))
    return c, errs


# ------------------------------------------------------------------------------ layer sweep
def run_layer(seed, pkg):
    r = np.random.default_rng(seed)
    m, n, k = int(r.integers(1, 3001)), int(r.integers(1, 301)), int(r.integers(1, 301))
    act = int(r.integers(0, 4))
    w0 = float(r.choice([1.0, 30.0])) if act == pkg.ops.ACT_SINE else 1.0
    bias = bool(r.random() < 0.8)
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(m, k, generator=g) * 2 - 1).requires_grad_(True)
    w = ((torch.rand(n, k, generator=g) * 2 - 1) / max(1.0, w0 * k ** 0.5)).requires_grad_(True)
    b = ((torch.rand(n, generator=g) * 2 - 1) / max(1.0, w0)).requires_grad_(True) if bias else None
    z = torch.nn.functional.linear(x, w, b)
    y = {0: lambda t: t, 1: torch.relu, 2: lambda t: torch.sin(w0 * t),
         3: torch.nn.functional.gelu}[act](z)
    dy = torch.rand(m, n, generator=g) * 2 - 1
    y.backward(dy)
    xc, wc = x.detach().cuda().requires_grad_(True), w.detach().cuda().requires_grad_(True)
    bc = b.detach().cuda().requires_grad_(True) if bias else None
    got = pkg.ops.linear_act(xc, wc, bc, act, w0)
    got.backward(dy.cuda())
    errs = dict(forward=rel(got.detach().cpu().numpy(), y.detach().numpy()),
                dx=rel(xc.grad.cpu().numpy(), x.grad.numpy()),
                dw=rel(wc.grad.cpu().numpy(), w.grad.numpy()))
    if bias:
        errs["db"] = rel(bc.grad.cpu().numpy(), b.grad.numpy())
    return dict(m=m, n=n, k=k, act=act, w0=w0, bias=bias), errs


# ------------------------------------------------------------------------------ whole steps
def build_step(c, pkg):
    finest = c["base"] * c["growth"]
    model = otrain.HashMlpModel(c["dim"], c["levels"], c["feats"], c["log2t"], c["base"], finest,
                                hidden=[c["hidden"]] * 2, seed=c["seed"], table_scale=0.5)
    net = pkg.models.HashMLP(c["dim"], c["levels"], c["feats"], c["log2t"], c["base"], finest,
                             dim_hidden=c["hidden"], n_layers=3, activation=torch.nn.ReLU,
                             batch_norm=False, final_activation=False, lr=5e-3)
    with torch.no_grad():
        net.encoder.table.copy_(torch.cat(model.tables))
        for blk, (w, b) in zip(net.decoder, model.mlp):
            blk[0].weight.copy_(w)
            blk[0].bias.copy_(b)
    g = torch.Generator().manual_seed(c["seed"] + 77)
    x = torch.rand(c["n"], c["dim"], generator=g)
    y = torch.rand(c["n"], 1, generator=g)
    return model, net.cuda(), x, y


def oracle_gradients(model, x, y, flip=None):
    """Gradients of the oracle's step, one array per tensor in the kernel's parameter order
    (table levels, then w, b per layer).  `flip` = (layer, row, unit) evaluates the SAME arithmetic with
    that one ReLU gate inverted."""
    ps = model.parameters()
    for p in ps:
        p.requires_grad_(True)
        p.grad = None
    z = ohash.encode(x, model.tables, model.resolutions)
    pre = []
    h = z
    for i, (w, b) in enumerate(model.mlp):
        a = torch.nn.functional.linear(h, w, b)
        if i == len(model.mlp) - 1:
            h = a
            break
        pre.append(a.detach())
        gate = (a > 0).to(a.dtype).detach()  # relu(a) = a * gate, same value and same gradient
        if flip is not None and flip[0] == i:
            gate[flip[1], flip[2]] = 1.0 - gate[flip[1], flip[2]]
        h = a * gate
    loss = omlp.mse_loss(h, y)
    loss.backward()
    grads = [p.grad.numpy().copy() for p in ps]  # one tensor per level, then w, b per layer
    for p in ps:
        p.requires_grad_(False)
        p.grad = None
    return float(loss), grads, pre


def rel_per_tensor(got, want):
    """Worst per-tensor error (the tolerance is per tensor: per level, per weight, per bias)."""
    return max(rel(g, w) for g, w in zip(got, want))


def kernel_gradients(net, x, y, pkg):
    step = pkg.trainer.FusedStep(net, net.configure_optimizers())
    assert step.use_tiny
    _, ws = step.forward(x.cuda(), train=True)
    step.backward(x.cuda(), y.cuda(), ws)
    enc = net.encoder
    grads = [enc.table.grad[slice(*enc._row_span(l))].cpu().numpy() for l in range(enc.n_levels)]
    for blk in net.decoder:
        grads += [blk[0].weight.grad.cpu().numpy(), blk[0].bias.grad.cpu().numpy()]
    return float(step.loss), grads


def explain_relu_kink(model, x, y, got, max_candidates=6):
    """(explained, detail).  Looks for hidden pre-activations at least six orders below the
    layer's typical magnitude; flips each such gate, one at a time, in the oracle and accepts iff
    the kernel's gradients match that evaluation to TOL."""
    _, _, pre = oracle_gradients(model, x, y)
    cands = []
    for layer, a in enumerate(pre):
        typical = float(a.abs().median())
        rows, units = torch.nonzero(a.abs() < 1e-6 * typical, as_tuple=True)
        cands += [(layer, int(r), int(u), float(a[r, u]), typical) for r, u in zip(rows, units)]
    if not cands or len(cands) > max_candidates:
        return False, f"{len(cands)} near-zero pre-activations"
    for layer, row, unit, val, typ in cands:
        _, flipped, _ = oracle_gradients(model, x, y, flip=(layer, row, unit))
        err = rel_per_tensor(got, flipped)
        if err <= TOL:
            return True, (f"coordinate {row}, layer {layer + 1} unit {unit}: pre-activation "
                          f"{val:.2e} (typical {typ:.2e}); with that gate flipped the kernel "
                          f"matches to {err:.1e}")
    return False, f"{len(cands)} near-zero pre-activations, no single flip explains the difference"


def run_step(seed, pkg):
    c = step_case(seed)
    model, net, x, y = build_step(c, pkg)
    want_loss, want, _ = oracle_gradients(model, x, y)
    got_loss, got = kernel_gradients(net, x, y, pkg)
    errs = dict(loss=abs(got_loss - want_loss) / max(abs(want_loss), 1e-30),
                grads=rel_per_tensor(got, want))
    note = None
    if errs["grads"] > TOL:
        ok, note = explain_relu_kink(model, x, y, got)
        errs["grads"] = 0.0 if ok else errs["grads"]
        note = ("ReLU kink: " if ok else "UNEXPLAINED: ") + note
    return c, errs, note


def run_siren(seed, pkg):
    r = np.random.default_rng(seed)
    dim, hidden = int(r.integers(1, 5)), int(r.choice([16, 64, 100, 256]))
    layers, n = int(r.integers(1, 5)), int(r.integers(1, 3001))
    model = otrain.SirenModel(dim, hidden, 1, layers, seed=seed)
    net = pkg.models.SirenNet(dim, hidden, 1, layers, lr=1e-4)
    with torch.no_grad():
        for layer, (w, b) in zip(list(net.layers) + [net.last_layer], model.params):
            layer.weight.copy_(w)
            layer.bias.copy_(b)
    net = net.cuda()
    g = torch.Generator().manual_seed(seed)
    x, y = torch.rand(n, dim, generator=g) * 2 - 1, torch.rand(n, 1, generator=g) * 2 - 1
    want_loss, _, grads = otrain.loss_and_grads(model, x, y)
    step = pkg.trainer.FusedStep(net, net.configure_optimizers())
    fused_loss = bool(seed & 1) and step.use_chain and step.chain_loss
    if fused_loss:  # the training step's form: loss inside the forward kernel, the backward chain continuing it
        step._chain_loss_pass(x.cuda(), y.cuda(), True, 1.0)
    else:
        _, ws = step.forward(x.cuda(), train=True)
        step.backward(x.cuda(), y.cuda(), ws)
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    want = torch.cat([g_.reshape(-1) for g_ in grads]).numpy()
    return dict(dim=dim, hidden=hidden, layers=layers, n=n, fused_loss=fused_loss), \
        dict(loss=abs(float(step.loss) - float(want_loss)) / max(abs(float(want_loss)), 1e-30),
             grads=rel(got, want))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--encoders", type=int, default=200)
    ap.add_argument("--layers", type=int, default=100)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--sirens", type=int, default=50)
    ap.add_argument("--first", type=int, default=1000, help="first seed of every sweep")
    args = ap.parse_args()
    import mri_interpolation_amd as pkg
    from mri_interpolation_amd import _lib, encoding, models, ops, trainer  # noqa: F401
    _lib.load()
    t0, failures, kinks = time.time(), 0, []
    for name, count, fn in (("encoder", args.encoders, run_encoder), ("layer", args.layers, run_layer),
                            ("step", args.steps, run_step), ("siren", args.sirens, run_siren)):
        for seed in range(args.first, args.first + count):
            out = fn(seed, pkg)
            case, errs = out[0], out[1]
            note = out[2] if len(out) > 2 else None
            bad = {k: v for k, v in errs.items() if not v <= TOL}
            if note:
                print(f"{name} seed {seed}: {note}", flush=True)
                if note.startswith("ReLU kink"):
                    kinks.append(seed)
            if bad:
                failures += 1
                print(f"FAIL {name} seed {seed}: {bad} case {case}", flush=True)
        print(f"{name}: {count} cases done, {time.time() - t0:.0f} s, {failures} failures so far",
              flush=True)
    print(f"fuzz finished: {failures} failures, {len(kinks)} steps explained by a ReLU kink "
          f"(seeds {kinks}) in {time.time() - t0:.0f} s")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
