#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE implementation.

Run in the build container only (the reference lives at /root/reference and never
travels):

    python tests/golden/make_golden.py

It imports the reference's own `encoding.py` and `models.py`, unmodified, with
`sys.modules` stand-ins for four imports that carry no arithmetic on the hot path
and are not installed here (pytorch_lightning -> nn.Module base class, utils,
commentjson, rff); see SURVEY.md section 8(c).  Every array written below is an
INPUT or an OUTPUT of those reference classes -- no reference source is stored.

Large inputs (hash tables, network weights) are not stored: both this script and
the tests rebuild them from (n, seed) with `oracle.detrand`.
"""
import gzip
import json
import os
import struct
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MRI_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import detrand, mlp as omlp, hashgrid as ohash  # noqa: E402


def import_reference():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        @property
        def device(self):
            p = next(self.parameters(), None)
            return p.device if p is not None else torch.device("cpu")

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule, pl.LightningDataModule = LightningModule, LightningDataModule
    plu = types.ModuleType("pytorch_lightning.utilities")
    plt = types.ModuleType("pytorch_lightning.utilities.types")
    plt.STEP_OUTPUT = object
    ut = types.ModuleType("utils")
    ut.create_mgrid = lambda shape: torch.stack(
        torch.meshgrid(*[torch.linspace(0, 1, s) for s in shape], indexing="ij"), dim=-1)
    rff = types.ModuleType("rff")
    rff.layers = types.ModuleType("rff.layers")
    sys.modules.update({
        "pytorch_lightning": pl, "pytorch_lightning.utilities": plu,
        "pytorch_lightning.utilities.types": plt, "utils": ut,
        "commentjson": json, "rff": rff, "rff.layers": rff.layers})
    sys.path.insert(0, REF)
    import encoding
    import models
    return encoding, models


def save(name, meta, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"{name:28s} {os.path.getsize(path) / 1024:9.1f} KiB")


def edge_rows(dim):
    """x rows that sit on the awkward spots of encoding.py:111-113."""
    rows = [np.zeros(dim), np.ones(dim), np.full(dim, 0.999999), np.full(dim, -0.3),
            np.full(dim, 1.2), np.full(dim, 0.5), np.full(dim, 1.0 / 16), np.full(dim, 15.0 / 16)]
    lin = torch.linspace(0, 1, 17).numpy()
    for k in (1, 5, 16):
        rows.append(np.full(dim, lin[k]))
    mixed = np.array([0.0, 1.0, 0.25, 0.75][:dim] + [0.5] * max(0, dim - 4))
    rows.append(mixed)
    return np.stack(rows).astype(np.float32)


def load_tables(enc, seed, scale):
    sizes = [lvl.embedding.weight.shape[0] for lvl in enc.levels]
    feats = enc.levels[0].embedding.weight.shape[1]
    tabs = ohash.init_tables(sizes, feats, seed, scale)
    with torch.no_grad():
        for lvl, t in zip(enc.levels, tabs):
            lvl.embedding.weight.copy_(t)
    return sizes


def sparse_grads(enc):
    idx, val = [], []
    for lvl in enc.levels:
        g = lvl.embedding.weight.grad
        nz = torch.nonzero(g.abs().sum(dim=1) != 0).flatten()
        idx.append(nz.numpy().astype(np.int32))
        val.append(g[nz].numpy())
    return idx, val


def encoder_fixture(name, enc, ctor, dim, n_rand, seed, scale):
    sizes = load_tables(enc, seed, scale)
    x = np.concatenate([detrand.uniform(n_rand * dim, seed + 7, 0.0, 1.0).reshape(n_rand, dim),
                        edge_rows(dim)])
    xt = torch.from_numpy(x)
    out = enc(xt)
    d_out = detrand.uniform(out.numel(), seed + 11, -1.0, 1.0).reshape(out.shape)
    out.backward(torch.from_numpy(d_out))
    idx, val = sparse_grads(enc)
    res = [[float(r) for r in np.atleast_1d(np.asarray(lvl.resolution, dtype=np.float64))]
           for lvl in enc.levels]
    arrays = {"x": x, "out": out.detach().numpy(), "d_out": d_out}
    for l, (i, v) in enumerate(zip(idx, val)):
        arrays[f"grad_idx_{l}"] = i
        arrays[f"grad_val_{l}"] = v
    save(name, dict(ctor=ctor, sizes=sizes, resolutions=res, table_seed=seed, table_scale=scale),
         **arrays)


def extra_models(encoding, models):
    """O9 / O10: SURVEY.md section 8(f) rank 4 -- ModulatedSirenNet (models.py:236-322) and
    the NeRF frequency encoding (encoding.py:43-66)."""
    for name, dim_in, hidden, n_layers, n, seed in [("modsiren_2d", 2, 64, 3, 192, 61),
                                                    ("modsiren_3d", 3, 128, 4, 160, 62)]:
        net = models.ModulatedSirenNet(dim_in=dim_in, dim_hidden=hidden, dim_out=1,
                                       n_layers=n_layers)
        sparams = omlp.siren_init(dim_in, hidden, 1, n_layers, seed)
        mparams = omlp.modulator_init(dim_in, hidden, n_layers, seed + 500)
        with torch.no_grad():
            for layer, (w, b) in zip(list(net.siren.layers) + [net.siren.last_layer], sparams):
                layer.weight.copy_(w)
                layer.bias.copy_(b)
            for seq, (w, b) in zip(net.modulator.layers, mparams):
                seq[0].weight.copy_(w)
                seq[0].bias.copy_(b)
        x = detrand.uniform(n * dim_in, seed + 1, -1.0, 1.0).reshape(n, dim_in)
        y = detrand.uniform(n, seed + 2, -1.0, 1.0).reshape(n, 1)
        xt = torch.from_numpy(x)
        pred = net(xt.clone())  # the reference multiplies in place
        loss = torch.nn.functional.mse_loss(torch.from_numpy(y), pred)
        loss.backward()
        arrays = dict(x=x, y=y, pred=pred.detach().numpy(), loss=np.float32(loss.item()))
        for i, layer in enumerate(list(net.siren.layers) + [net.siren.last_layer]):
            arrays[f"siren_gw_{i}"] = layer.weight.grad.numpy().copy()
            arrays[f"siren_gb_{i}"] = layer.bias.grad.numpy().copy()
        for i, seq in enumerate(net.modulator.layers):
            arrays[f"mod_gw_{i}"] = seq[0].weight.grad.numpy().copy()
            arrays[f"mod_gb_{i}"] = seq[0].bias.grad.numpy().copy()
        save(name, dict(dim_in=dim_in, dim_hidden=hidden, n_layers=n_layers, seed=seed,
                        w0=30.0, w0_initial=30.0,
                        state_dict_keys=sorted(net.state_dict().keys())), **arrays)

    arrays = {}
    for dim, n_levels in [(2, 10), (3, 6), (4, 4)]:
        enc = encoding.Frequency(dim, n_levels=n_levels)
        x = detrand.uniform(200 * dim, 70 + dim, -1.0, 1.0).reshape(200, dim)
        x[:4] = [[0.0] * dim, [1.0] * dim, [-1.0] * dim, [0.5] * dim]
        xt = torch.from_numpy(x).requires_grad_(True)
        out = enc(xt)
        g = detrand.uniform(out.numel(), 80 + dim, -1.0, 1.0).reshape(out.shape)
        out.backward(torch.from_numpy(g))
        arrays.update({f"x_{dim}": x, f"out_{dim}": out.detach().numpy(), f"g_{dim}": g,
                       f"dx_{dim}": xt.grad.numpy()})
    save("frequency", dict(cases=[[2, 10], [3, 6], [4, 4]]), **arrays)


def read_sample_volume():
    """int16 voxels + scl_slope of the reference's sample NIfTI (a data file, parsed by hand:
    nibabel is not installed)."""
    raw = gzip.open(os.path.join(REF, "sample_ankle_dyn_mri.nii.gz")).read()
    dims = struct.unpack("<8h", raw[40:56])
    vox_offset = int(struct.unpack("<f", raw[108:112])[0])
    slope, inter = struct.unpack("<ff", raw[112:120])
    shape = dims[1:1 + dims[0]]
    vol = np.frombuffer(raw, dtype="<i2", offset=vox_offset,
                        count=int(np.prod(shape))).reshape(shape, order="F")
    return vol, shape, slope, inter


def round2_fixtures(encoding, models):
    """Fixtures added in round 2 (the earlier ones stay byte-identical):
    O3b the per-axis 4-D encoder BASELINE config 5 trains with; O7b the notebook's HashMLP
    decoder (Linear -> GELU blocks, no BatchNorm: ReprésentationsImplicites.ipynb cell 37,
    i.e. models.py:712-739 with the BatchNorm1d / Dropout members skipped); O8b the whole
    sample volume (config 5's workload)."""
    fin4 = 16 * 1.4 ** 15
    kw = dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
              base_resolution=(16, 16, 5, 7), finest_resolution=(fin4, fin4, 5, 7))
    enc = encoding.MultiResHashGridV2(4, **kw)
    ctor = dict(cls="MultiResHashGridV2", dim=4,
                **{k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()})
    encoder_fixture("enc_v2_cfg5", enc, ctor, 4, 96, seed=len("enc_v2_cfg5"), scale=0.5)

    # ---- O7b: notebook decoder, forward + loss + every gradient + 2 Adam steps ---------------
    nb = dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=23,
              base_resolution=(64, 64, 5), finest_resolution=(512, 512, 15))
    hm = models.HashMLP(dim_in=3, dim_hidden=64, dim_out=1, n_layers=2, lr=5e-3, **nb)
    sizes = load_tables(hm.encoder, 91, 0.5)
    params = omlp.linear_init([16, 64, 1], 92)
    blocks = [nn.Sequential(blk[0], blk[2]) for blk in hm.decoder]  # Linear -> GELU
    assert all(isinstance(b[0], nn.Linear) and isinstance(b[1], nn.GELU) for b in blocks)
    with torch.no_grad():
        for b, (w, bias) in zip(blocks, params):
            b[0].weight.copy_(w)
            b[0].bias.copy_(bias)
    trainable = list(hm.encoder.parameters()) + [q for b in blocks for q in b.parameters()]
    opt = torch.optim.Adam(trainable, lr=5e-3)
    arrays = {}
    for step in range(2):
        x = detrand.uniform(384 * 3, 930 + step, 0.0, 1.0).reshape(384, 3)
        y = detrand.uniform(384, 940 + step, 0.0, 1.0).reshape(384, 1)
        opt.zero_grad()
        z = hm.encoder(torch.from_numpy(x))
        for b in blocks:
            z = b(z)
        loss = torch.nn.functional.mse_loss(z, torch.from_numpy(y))  # notebook: (y_pred, y)
        loss.backward()
        arrays[f"x_{step}"], arrays[f"y_{step}"] = x, y
        arrays[f"pred_{step}"] = z.detach().numpy().copy()
        arrays[f"loss_{step}"] = np.float32(loss.item())
        if step == 0:
            idx, val = sparse_grads(hm.encoder)
            for l, (i, v) in enumerate(zip(idx, val)):
                arrays[f"grad_idx_{l}"], arrays[f"grad_val_{l}"] = i, v.copy()
            for i, b in enumerate(blocks):
                arrays[f"gw_{i}"] = b[0].weight.grad.numpy().copy()
                arrays[f"gb_{i}"] = b[0].bias.grad.numpy().copy()
        opt.step()
        for i, b in enumerate(blocks):
            arrays[f"w_{step}_{i}"] = b[0].weight.detach().numpy().copy()
            arrays[f"b_{step}_{i}"] = b[0].bias.detach().numpy().copy()
        # tables after the step: the rows the step touched (Adam moves every touched row)
        for l, lvl in enumerate(hm.encoder.levels):
            rows = arrays[f"grad_idx_{l}"]
            arrays[f"table_{step}_{l}"] = lvl.embedding.weight.detach().numpy()[rows].copy()
    res = [[float(r) for r in np.atleast_1d(np.asarray(lvl.resolution, dtype=np.float64))]
           for lvl in hm.encoder.levels]
    save("hashmlp_gelu_notebook",
         dict(ctor=dict(dim=3, **{k: (list(v) if isinstance(v, tuple) else v)
                                  for k, v in nb.items()}),
              sizes=sizes, resolutions=res, table_seed=91, table_scale=0.5, mlp_seed=92,
              dims=[16, 64, 1], lr=5e-3, steps=2), **arrays)

    # ---- O6c: end-to-end SIREN at the width of BASELINE config 3 (3 -> 256 x 5 -> 1), 3 Adam steps:
    #      the shape the fused chain kernels (csrc/siren_chain.hip) serve ---------------------------
    net = models.SirenNet(dim_in=3, dim_hidden=256, dim_out=1, n_layers=5)
    params = omlp.siren_init(3, 256, 1, 5, 63)
    layers = list(net.layers) + [net.last_layer]
    with torch.no_grad():
        for layer, (w, b) in zip(layers, params):
            layer.weight.copy_(w)
            layer.bias.copy_(b)
    opt = net.configure_optimizers()
    arrays = {}
    for step in range(3):
        x = detrand.uniform(600 * 3, 760 + step, -1.0, 1.0).reshape(600, 3)
        y = detrand.uniform(600, 770 + step, -1.0, 1.0).reshape(600, 1)
        opt.zero_grad()
        loss = net.training_step((torch.from_numpy(x), torch.from_numpy(y)), step)
        loss.backward()
        if step == 0:
            arrays["pred_0"] = net(torch.from_numpy(x)).detach().numpy().copy()
            for i, layer in enumerate(layers):
                arrays[f"gb_{i}"] = layer.bias.grad.numpy().copy()
                arrays[f"gw_head_{i}"] = layer.weight.grad.numpy()[:8].copy()
        opt.step()
        arrays[f"x_{step}"], arrays[f"y_{step}"] = x, y
        arrays[f"loss_{step}"] = np.float32(loss.item())
        for i, layer in enumerate(layers):
            w = layer.weight.detach().numpy()
            arrays[f"w_{step}_{i}"] = (w if w.size <= 8192 else w[:16]).copy()  # head rows suffice
            arrays[f"wnorm_{step}_{i}"] = np.float64(np.linalg.norm(w.astype(np.float64)))
            arrays[f"b_{step}_{i}"] = layer.bias.detach().numpy().copy()
    save("e2e_siren256_adam", dict(dim_in=3, dim_hidden=256, n_layers=5, seed=63, lr=1e-4, steps=3),
         **arrays)

    # ---- O8b: the sample volume itself ----------------------------------------------------------
    vol, shape, slope, inter = read_sample_volume()
    save("sample_volume", dict(shape=list(shape), scl_slope=slope, scl_inter=inter,
                               source="sample_ankle_dyn_mri.nii.gz (int16 voxels, F-order file "
                                      "reshaped to the NIfTI axes x, y, z, t)"),
         raw_int16=np.ascontiguousarray(vol))


def round3_fixtures(encoding, models):
    """Fixture added in round 3 (the earlier ones stay byte-identical): O7c the reference's DEFAULT model
    -- `config.model_cls = HashMLP`, decoder blocks Linear -> BatchNorm1d -> GELU -> Dropout(0)
    (models.py:712-739), applied in sequence -- in train() mode: loss, EVERY gradient
    (tables, Linear and BatchNorm weights / biases), the running statistics, and two Adam steps over
    the parameters `configure_optimizers` hands to Adam (models.py:68-70), the dead `layers.*` stack
    included (its gradients are None, Adam skips it: Q3)."""
    kw = dict(n_levels=4, n_features_per_level=1, log2_hashmap_size=23,
              base_resolution=(64, 64, 5), finest_resolution=(352, 352, 15))
    hm = models.HashMLP(dim_in=3, dim_hidden=64, dim_out=1, n_layers=2, lr=5e-3, **kw)
    sizes = load_tables(hm.encoder, 95, 0.5)
    params = omlp.linear_init([4, 64, 1], 96)
    with torch.no_grad():
        for blk, (w, b) in zip(hm.decoder, params):
            blk[0].weight.copy_(w)
            blk[0].bias.copy_(b)
            # BatchNorm affine parameters away from their (1, 0) init, so that their gradients matter
            blk[1].weight.copy_(torch.from_numpy(detrand.uniform(blk[1].weight.numel(), 970, 0.5, 1.5)))
            blk[1].bias.copy_(torch.from_numpy(detrand.uniform(blk[1].bias.numel(), 971, -0.2, 0.2)))
    hm.train()
    opt = hm.configure_optimizers()  # Adam(self.parameters(), lr) -- reference models.py:68-70
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 5e-3
    arrays = {}
    for step in range(2):
        x = detrand.uniform(320 * 3, 980 + step, 0.0, 1.0).reshape(320, 3)
        y = detrand.uniform(320, 990 + step, 0.0, 1.0).reshape(320, 1)
        opt.zero_grad()
        z = hm.encoder(torch.from_numpy(x))
        for blk in hm.decoder:  # HashMLP.forward as intended (Q1)
            z = blk(z)
        loss = torch.nn.functional.mse_loss(torch.from_numpy(y), z)  # training_step, models.py:64
        loss.backward()
        arrays[f"x_{step}"], arrays[f"y_{step}"] = x, y
        arrays[f"pred_{step}"] = z.detach().numpy().copy()
        arrays[f"loss_{step}"] = np.float32(loss.item())
        if step == 0:
            idx, val = sparse_grads(hm.encoder)
            for l, (i, v) in enumerate(zip(idx, val)):
                arrays[f"grad_idx_{l}"], arrays[f"grad_val_{l}"] = i, v.copy()
            for i, blk in enumerate(hm.decoder):
                arrays[f"gw_{i}"] = blk[0].weight.grad.numpy().copy()
                arrays[f"gb_{i}"] = blk[0].bias.grad.numpy().copy()
                arrays[f"bn_gw_{i}"] = blk[1].weight.grad.numpy().copy()
                arrays[f"bn_gb_{i}"] = blk[1].bias.grad.numpy().copy()
        opt.step()
        for i, blk in enumerate(hm.decoder):
            arrays[f"w_{step}_{i}"] = blk[0].weight.detach().numpy().copy()
            arrays[f"b_{step}_{i}"] = blk[0].bias.detach().numpy().copy()
            arrays[f"bn_w_{step}_{i}"] = blk[1].weight.detach().numpy().copy()
            arrays[f"bn_b_{step}_{i}"] = blk[1].bias.detach().numpy().copy()
            arrays[f"bn_mean_{step}_{i}"] = blk[1].running_mean.numpy().copy()
            arrays[f"bn_var_{step}_{i}"] = blk[1].running_var.numpy().copy()
        for l, lvl in enumerate(hm.encoder.levels):
            rows = arrays[f"grad_idx_{l}"]
            arrays[f"table_{step}_{l}"] = lvl.embedding.weight.detach().numpy()[rows].copy()
    hm.eval()  # the running statistics in use: predict_step's forward
    z = hm.encoder(torch.from_numpy(arrays["x_0"]))
    for blk in hm.decoder:
        z = blk(z)
    arrays["pred_eval_after"] = z.detach().numpy().copy()
    res = [[float(r) for r in np.atleast_1d(np.asarray(lvl.resolution, dtype=np.float64))]
           for lvl in hm.encoder.levels]
    save("hashmlp_bn_adam",
         dict(ctor=dict(dim=3, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}),
              sizes=sizes, resolutions=res, table_seed=95, table_scale=0.5, mlp_seed=96, bn_seeds=[970, 971],
              dims=[4, 64, 1], lr=5e-3, steps=2, batch=320), **arrays)


def main():
    torch.manual_seed(1337)
    torch.set_num_threads(4)
    encoding, models = import_reference()
    if "--extra-only" in sys.argv:  # leave the existing fixtures untouched
        return extra_models(encoding, models)
    if "--round3-only" in sys.argv:
        return round3_fixtures(encoding, models)
    if "--round2-only" in sys.argv:
        return round2_fixtures(encoding, models)

    # ---- O1: hash ids (encoding.py:69-78) ------------------------------------------------
    arrays, cases = {}, []
    for dim in (2, 3, 4):
        for size in (4096, 8000, 15625, 262144, 274625, 524288):
            n = 384
            idx = detrand.integers(n * dim, 100 * dim + size % 97, -40, 3000).reshape(n, dim)
            idx[:8] = np.array([[-1] * dim, [0] * dim, [1] * dim, [2489] * dim, [2490] * dim,
                                [-2490] * dim, [65535] * dim, [2 ** 20 + 3] * dim])
            primes = torch.tensor(encoding.PRIMES, dtype=torch.int64)
            got = encoding.fast_hash(torch.from_numpy(idx.copy()), primes, size).numpy()
            key = f"d{dim}_t{size}"
            arrays["idx_" + key], arrays["ids_" + key] = idx, got
            cases.append([dim, size])
    save("hash_ids", dict(cases=cases, primes=list(encoding.PRIMES)), **arrays)

    # ---- O2: isotropic encoder forward + table gradient ----------------------------------
    fin4 = 16 * 1.4 ** 15
    for name, dim, kw, n_rand, scale in [
        ("enc_cfg2", 3, dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                             base_resolution=16, finest_resolution=512), 256, 0.5),
        ("enc_cfg4", 3, dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                             base_resolution=16, finest_resolution=fin4), 256, 0.5),
        ("enc_cfg5_4d", 4, dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                                base_resolution=16, finest_resolution=fin4), 96, 0.5),
        ("enc_defaults_2d", 2, dict(), 256, 1e-4),
        ("enc_f4_small", 3, dict(n_levels=5, n_features_per_level=4, log2_hashmap_size=12,
                                 base_resolution=4, finest_resolution=40), 256, 0.5),
    ]:
        enc = encoding.MultiResHashGrid(dim, **kw)
        encoder_fixture(name, enc, dict(cls="MultiResHashGrid", dim=dim, **kw), dim, n_rand,
                        seed=len(name), scale=scale)

    # ---- O3: anisotropic V2 encoder (encoding.py:273-336) --------------------------------
    for name, kw in [
        ("enc_v2_hashconfig", dict(n_levels=4, n_features_per_level=1, log2_hashmap_size=23,
                                   base_resolution=(64, 64, 5), finest_resolution=(352, 352, 15))),
        ("enc_v2_notebook", dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=23,
                                 base_resolution=(64, 64, 5), finest_resolution=(512, 512, 15))),
    ]:
        enc = encoding.MultiResHashGridV2(3, **kw)
        ctor = dict(cls="MultiResHashGridV2", dim=3, **{k: (list(v) if isinstance(v, tuple) else v)
                                                         for k, v in kw.items()})
        encoder_fixture(name, enc, ctor, 3, 256, seed=len(name), scale=0.5)

    # ---- O4: SIREN (models.py:108-233) ----------------------------------------------------
    for name, dim_in, hidden, n_layers, n in [("siren_3d_5x256", 3, 256, 5, 192),
                                              ("siren_2d_4x352", 2, 352, 4, 192),
                                              ("siren_2d_3x64", 2, 64, 3, 256)]:
        seed = 40 + n_layers
        net = models.SirenNet(dim_in=dim_in, dim_hidden=hidden, dim_out=1, n_layers=n_layers)
        params = omlp.siren_init(dim_in, hidden, 1, n_layers, seed)
        with torch.no_grad():
            for layer, (w, b) in zip(list(net.layers) + [net.last_layer], params):
                layer.weight.copy_(w)
                layer.bias.copy_(b)
        x = detrand.uniform(n * dim_in, seed + 1, -1.0, 1.0).reshape(n, dim_in)
        y = detrand.uniform(n, seed + 2, -1.0, 1.0).reshape(n, 1)
        xt = torch.from_numpy(x).requires_grad_(True)
        pred = net(xt)
        loss = net.training_step((xt, torch.from_numpy(y)), 0)
        loss.backward()
        arrays = dict(x=x, y=y, pred=pred.detach().numpy(), loss=np.float32(loss.item()),
                      dx=xt.grad.numpy())
        for i, layer in enumerate(list(net.layers) + [net.last_layer]):
            gw, gb = layer.weight.grad.numpy(), layer.bias.grad.numpy()
            arrays[f"gb_{i}"] = gb
            arrays[f"gw_norm_{i}"] = np.float64(np.linalg.norm(gw.astype(np.float64)))
            arrays[f"gw_head_{i}"] = gw if gw.size <= 8192 else gw[:8]
        save(name, dict(dim_in=dim_in, dim_hidden=hidden, n_layers=n_layers, seed=seed,
                        w0=30.0, w0_initial=30.0), **arrays)

    # ---- O5: ReLU MLP built exactly like BaseMLP.layers (models.py:46-56) -----------------
    for name, hidden in [("relu_mlp_64", 64), ("relu_mlp_128", 128)]:
        seed = hidden
        net = models.BaseMLP(dim_in=32, dim_out=1, dim_hidden=hidden, n_layers=3)
        params = omlp.linear_init([32, hidden, hidden, 1], seed)
        lin = [m for m in net.layers if isinstance(m, nn.Linear)]
        with torch.no_grad():
            for m, (w, b) in zip(lin, params):
                m.weight.copy_(w)
                m.bias.copy_(b)
        x = detrand.uniform(256 * 32, seed + 1, -1.0, 1.0).reshape(256, 32)
        y = detrand.uniform(256, seed + 2, 0.0, 1.0).reshape(256, 1)
        arrays = dict(x=x, y=y)
        for tag, stack in [("act", net.layers), ("lin", net.layers[:-1])]:
            net.zero_grad()
            xt = torch.from_numpy(x).requires_grad_(True)
            pred = stack(xt)
            loss = torch.nn.functional.mse_loss(torch.from_numpy(y), pred)
            loss.backward()
            arrays[f"pred_{tag}"] = pred.detach().numpy()
            arrays[f"loss_{tag}"] = np.float32(loss.item())
            arrays[f"dx_{tag}"] = xt.grad.numpy()
            for i, m in enumerate(lin):
                arrays[f"gw_{tag}_{i}"] = m.weight.grad.numpy().copy()
                arrays[f"gb_{tag}_{i}"] = m.bias.grad.numpy().copy()
        save(name, dict(dims=[32, hidden, hidden, 1], seed=seed), **arrays)

    # ---- O6: end-to-end hash + ReLU MLP, 3 Adam steps (models.py:61-70) -------------------
    kw = dict(n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=4,
              finest_resolution=32)
    enc = encoding.MultiResHashGrid(3, **kw)
    sizes = load_tables(enc, 77, 1e-4)
    dec = models.BaseMLP(dim_in=8, dim_out=1, dim_hidden=16, n_layers=3).layers[:-1]
    params = omlp.linear_init([8, 16, 16, 1], 78)
    lin = [m for m in dec if isinstance(m, nn.Linear)]
    with torch.no_grad():
        for m, (w, b) in zip(lin, params):
            m.weight.copy_(w)
            m.bias.copy_(b)
    opt = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()), lr=5e-3)
    arrays = {}
    for step in range(3):
        x = detrand.uniform(512 * 3, 900 + step, 0.0, 1.0).reshape(512, 3)
        y = detrand.uniform(512, 950 + step, 0.0, 1.0).reshape(512, 1)
        opt.zero_grad()
        pred = dec(enc(torch.from_numpy(x)))
        loss = torch.nn.functional.mse_loss(torch.from_numpy(y), pred)
        loss.backward()
        opt.step()
        arrays[f"x_{step}"], arrays[f"y_{step}"] = x, y
        arrays[f"loss_{step}"] = np.float32(loss.item())
        for l, lvl in enumerate(enc.levels):
            arrays[f"table_{step}_{l}"] = lvl.embedding.weight.detach().numpy().copy()
        for i, m in enumerate(lin):
            arrays[f"w_{step}_{i}"] = m.weight.detach().numpy().copy()
            arrays[f"b_{step}_{i}"] = m.bias.detach().numpy().copy()
    save("e2e_hash_adam", dict(ctor=dict(dim=3, **kw), sizes=sizes, dims=[8, 16, 16, 1],
                               table_seed=77, table_scale=1e-4, mlp_seed=78, lr=5e-3, steps=3),
         **arrays)

    # ---- O6b: end-to-end SIREN, 3 Adam steps ----------------------------------------------
    net = models.SirenNet(dim_in=2, dim_hidden=32, dim_out=1, n_layers=3)
    params = omlp.siren_init(2, 32, 1, 3, 61)
    layers = list(net.layers) + [net.last_layer]
    with torch.no_grad():
        for layer, (w, b) in zip(layers, params):
            layer.weight.copy_(w)
            layer.bias.copy_(b)
    opt = net.configure_optimizers()
    arrays = {}
    for step in range(3):
        x = detrand.uniform(512 * 2, 700 + step, -1.0, 1.0).reshape(512, 2)
        y = detrand.uniform(512, 750 + step, -1.0, 1.0).reshape(512, 1)
        opt.zero_grad()
        loss = net.training_step((torch.from_numpy(x), torch.from_numpy(y)), step)
        loss.backward()
        opt.step()
        arrays[f"x_{step}"], arrays[f"y_{step}"] = x, y
        arrays[f"loss_{step}"] = np.float32(loss.item())
        for i, layer in enumerate(layers):
            arrays[f"w_{step}_{i}"] = layer.weight.detach().numpy().copy()
            arrays[f"b_{step}_{i}"] = layer.bias.detach().numpy().copy()
    save("e2e_siren_adam", dict(dim_in=2, dim_hidden=32, n_layers=3, seed=61, lr=1e-4, steps=3),
         **arrays)

    # ---- O7: HashMLP as intended (encoder + sequential blocks) ----------
    hm = models.HashMLP(dim_in=3, n_levels=4, n_features_per_level=1, log2_hashmap_size=23,
                        base_resolution=(64, 64, 5), finest_resolution=(352, 352, 15),
                        dim_hidden=64, dim_out=1, n_layers=2)
    sizes = load_tables(hm.encoder, 88, 0.5)
    params = omlp.linear_init([4, 64, 1], 89)
    with torch.no_grad():
        for blk, (w, b) in zip(hm.decoder, params):
            blk[0].weight.copy_(w)
            blk[0].bias.copy_(b)
    x = detrand.uniform(256 * 3, 90, 0.0, 1.0).reshape(256, 3)
    arrays = dict(x=x)
    for mode in ("train", "eval"):
        hm.train(mode == "train")
        z = hm.encoder(torch.from_numpy(x))
        for blk in hm.decoder:
            z = blk(z)
        arrays[f"pred_{mode}"] = z.detach().numpy()
    for i, blk in enumerate(hm.decoder):
        arrays[f"bn_mean_{i}"] = blk[1].running_mean.numpy().copy()
        arrays[f"bn_var_{i}"] = blk[1].running_var.numpy().copy()
    keys = sorted(hm.state_dict().keys())
    save("hashmlp_intended", dict(sizes=sizes, table_seed=88, table_scale=0.5, mlp_seed=89,
                                  dims=[4, 64, 1], state_dict_keys=keys), **arrays)

    # ---- O8: one slice of the sample volume (data fixture for BASELINE config 1) ----------
    raw = gzip.open(os.path.join(REF, "sample_ankle_dyn_mri.nii.gz")).read()
    dims = struct.unpack("<8h", raw[40:56])
    vox_offset = int(struct.unpack("<f", raw[108:112])[0])
    slope, inter = struct.unpack("<ff", raw[112:120])
    shape = dims[1:1 + dims[0]]
    vol = np.frombuffer(raw, dtype="<i2", offset=vox_offset,
                        count=int(np.prod(shape))).reshape(shape, order="F")
    save("sample_slice_z3_t7", dict(shape=list(shape), scl_slope=slope, scl_inter=inter,
                                    slice="[:, :, 3, 7]"),
         raw_int16=np.ascontiguousarray(vol[:, :, 3, 7]))
    extra_models(encoding, models)
    round2_fixtures(encoding, models)
    round3_fixtures(encoding, models)


if __name__ == "__main__":
    main()
