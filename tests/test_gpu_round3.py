"""Round-3 parity cases on the MI355X.

* The table-gradient record formats (`mri_set_option("bwd_records", ...)`) judged PER SLOT against a
  float64 evaluation of the same sum, with rigorous bounds: an f32 product is within 2^-24 of its
  magnitude, so no f32 evaluation of a slot can be expected closer than 2^-24 (sum of magnitudes + |sum|).
  Reference: autograd of encoding.py:127-128 (f32 products, f32 accumulation).
* A 200-step training run with packed against f32 records.
* float64 yardsticks for what rounds 1-2 compared at widened tolerances.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REL_TOL, ROOT, assert_close, load_golden
from oracle import detrand
from oracle import hashgrid as ohash
from oracle import mlp as omlp
from oracle import train as otrain

pytestmark = pytest.mark.gpu

FIN4 = 16 * 1.4 ** 15
U = 2.0 ** -24  # unit roundoff of f32


@pytest.fixture(scope="module")
def amd():
    from mri_interpolation_amd import _lib, datamodules, encoding, models, ops, trainer
    assert torch.cuda.is_available(), "these tests need the GPU"
    _lib.load()
    yield type("NS", (), dict(lib=_lib, ops=ops, encoding=encoding, models=models,
                              trainer=trainer, datamodules=datamodules))
    _lib.set_option("bwd_records", 0)


def _report(name, payload):
    """Numbers a reader of DESIGN.md wants to see again: written beside the other GPU artefacts."""
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name), "w") as f:
            json.dump(payload, f, indent=1)


# --------------------------------------------------------------------------- record formats
def _cfg4_gradient_case(n, seed):
    """Coordinates and an incoming gradient d_enc whose magnitudes spread over ~12 binades per level
    (as the decoder's do: a few coordinates carry most of the loss)."""
    x = torch.from_numpy(detrand.uniform(n * 3, seed, 0.0, 1.0).reshape(n, 3))
    g = torch.from_numpy(detrand.uniform(n * 32, seed + 1, -1.0, 1.0).reshape(n, 32))
    spread = torch.from_numpy(detrand.uniform(n, seed + 2, -6.0, 2.0).reshape(n, 1))
    return x, (g * torch.exp2(spread) * 1e-5).float()


def _per_slot_stats(got, total, mag, floor):
    """max, rms and median of |got - exact| / |exact| over the slots whose exact value is above `floor`."""
    sel = total.abs() > floor
    rel = ((got[sel] - total[sel]).abs() / total[sel].abs())
    return float(rel.max()), float(rel.pow(2).mean().sqrt()), int(sel.sum()), float(rel.median())


def test_record_formats_per_slot_against_float64(amd):
    """BASELINE config 4 at its full batch (2^18 coordinates, 16 levels, T 2^19): every table-gradient
    path -- f32 records (default), packed 8-byte records, global f32 atomics -- against the float64 sum,
    slot by slot."""
    ops = amd.ops
    n = 1 << 18
    enc = amd.encoding.MultiResHashGrid(3, 16, 2, 19, 16, FIN4).cuda()
    res, sizes = ohash.resolutions_for(3, 16, 19, 16, FIN4)
    x, d = _cfg4_gradient_case(n, 71)
    yard = ohash.table_gradient_f64(x, d, sizes, res, 2)
    xg, dg = x.cuda(), d.t().contiguous().cuda()  # feature-major, as the fused step hands it over

    def run(method, records):
        amd.lib.set_option("bwd_records", records)
        g = torch.zeros_like(enc.table.data)
        ops.hashgrid_backward(enc.desc, xg, dg, g, feature_major=True, method=method)
        return g.double().cpu()

    paths = {"f32_records": run(2, 0), "packed_records": run(2, 1), "f32_atomics": run(1, 0)}
    amd.lib.set_option("bwd_records", 0)
    stats = {k: [] for k in ("f32_records", "packed_records", "f32_atomics")}
    for l in range(16):
        lo, hi = enc._row_span(l)
        total, mag, count = yard[l]
        gmax = float(d[:, 2 * l:2 * l + 2].abs().max())
        k = count.double().unsqueeze(1)
        # fixed point: one unit of 2^-40 max|g| per record at most (level_exponent keeps >= 40 bits)
        fixed = k * gmax * 2.0 ** -40
        bound_exact = U * (mag + total.abs()) * 1.001 + fixed
        # packed: every product first rounded to >= 18 significant bits (2^-18 of its magnitude on top
        # of its own f32 rounding), contributions below 2^(E-45) may vanish (E = exponent of max|g|)
        bound_packed = (2.0 ** -18 + U) * mag * 1.001 + U * total.abs() + fixed + k * gmax * 2.0 ** -43
        # f32 atomics: products rounded, then k - 1 rounded additions of partial sums <= mag
        bound_atomic = U * mag * (k + 1.0) * 1.001
        for name, bound in (("f32_records", bound_exact), ("packed_records", bound_packed),
                            ("f32_atomics", bound_atomic)):
            err = (paths[name][lo:hi] - total).abs()
            worst = float((err - bound).max())
            assert worst <= 0.0, f"{name} level {l}: a slot exceeds its bound by {worst:.3e}"
            # nothing in a slot no corner hashes to
            assert float(paths[name][lo:hi][count == 0].abs().max()) == 0.0 if (count == 0).any() else True
            stats[name].append(_per_slot_stats(paths[name][lo:hi], total, mag,
                                               1e-6 * float(total.abs().max())))
            # the usual per-level bound of the parity suite
            assert_close(paths[name][lo:hi].numpy(), total.numpy(), REL_TOL, f"{name} level {l}")
    summary = {k: dict(max_rel=max(s[0] for s in v), rms_rel=float(np.sqrt(np.mean([s[1] ** 2 for s in v]))),
                       median_rel=float(np.median([s[3] for s in v])), slots=sum(s[2] for s in v))
               for k, v in stats.items()}
    summary["ratio_packed_over_f32_records_rms"] = summary["packed_records"]["rms_rel"] / summary["f32_records"]["rms_rel"]
    summary["ratio_atomics_over_f32_records_rms"] = summary["f32_atomics"]["rms_rel"] / summary["f32_records"]["rms_rel"]
    _report("r3_records_accuracy.json", summary)
    print(json.dumps(summary))
    # the default records are at least as accurate as the reference's own accumulation (f32 atomics /
    # sequential f32 adds); the packed ones are NOT f32: measured here, they stay an option
    assert summary["f32_records"]["rms_rel"] <= summary["f32_atomics"]["rms_rel"] * 1.05
    assert summary["f32_records"]["median_rel"] <= U
    assert summary["ratio_packed_over_f32_records_rms"] <= 64.0


def test_record_formats_on_the_golden_encoders(amd):
    """Every golden encoder fixture through both F = 2 record formats: nothing lands in a slot the
    reference did not touch, the f32 records lose no entry the reference filled above 2^-36 of the
    level's maximum (the fixed-point unit is 2^-40 of max |d_out|), the packed ones none above 1e-9 of it."""
    for name in ("enc_cfg2", "enc_cfg4", "enc_cfg5_4d", "enc_defaults_2d", "enc_v2_cfg5", "enc_v2_notebook"):
        fx = load_golden(name)
        c = dict(fx.meta["ctor"])
        cls = getattr(amd.encoding, c.pop("cls"))
        dim = c.pop("dim")
        for key in ("base_resolution", "finest_resolution"):
            if isinstance(c.get(key), list):
                c[key] = tuple(c[key])
        enc = cls(dim, **c).cuda()
        if enc.n_features_per_level != 2:
            continue
        x, d_out = torch.as_tensor(fx["x"]).cuda(), torch.as_tensor(fx["d_out"]).cuda()
        for records, floor in ((0, 2.0 ** -36), (1, 1e-9)):
            amd.lib.set_option("bwd_records", records)
            g = torch.zeros_like(enc.table.data)
            amd.ops.hashgrid_backward(enc.desc, x, d_out, g, method=2)
            g = g.cpu()
            for l in range(enc.n_levels):
                lo, hi = enc._row_span(l)
                got = g[lo:hi].numpy()
                want = np.zeros_like(got)
                want[fx[f"grad_idx_{l}"]] = fx[f"grad_val_{l}"]
                nz = np.nonzero(np.abs(got).sum(axis=1))[0]
                assert np.isin(nz, fx[f"grad_idx_{l}"]).all(), f"{name} level {l}: stray slot"
                big = np.abs(want) > floor * np.abs(want).max()
                assert (got[big] != 0).all(), f"{name} level {l} records {records}: lost entry"
                assert_close(got, want, REL_TOL, f"{name} level {l} records {records}")
    amd.lib.set_option("bwd_records", 0)


def _train_cfg4_like(amd, records, steps, shape=(96, 96, 96)):
    """`steps` fused steps of the config-4 model (hash L16 F2 T2^19 growth 1.4 + MLP 128) on the analytic
    phantom with shuffled on-device batches; returns (flat parameters, PSNR over all voxels)."""
    amd.lib.set_option("bwd_records", records)
    dev = torch.device("cuda", 0)
    vol = amd.datamodules.phantom_volume(shape, device=dev)
    ds = amd.datamodules.MriImage(volume=vol, device=dev)
    loader = amd.datamodules.DeviceLoader(ds, 1 << 16, shuffle=True, drop_last=True, seed=1337)
    torch.manual_seed(1337)
    net = amd.models.HashMLP(3, 16, 2, 19, 16, FIN4, dim_hidden=128, n_layers=3,
                             activation=torch.nn.ReLU, batch_norm=False, final_activation=False,
                             lr=5e-3).cuda()
    step = amd.trainer.FusedStep(net, net.configure_optimizers())
    assert step.use_tiny
    pipe = amd.datamodules.BatchPipeline(loader)
    for _ in range(steps):
        coords, target = pipe.current()
        step.train_step(coords, target, pipe.produce_next)
        pipe.advance()
    preds = []
    with torch.no_grad():
        for xb, _ in amd.datamodules.DeviceLoader(ds, 1 << 18, shuffle=False):
            preds.append(step.forward(xb)[0].clone())
    db = amd.trainer.psnr(torch.cat(preds), ds.pixels)
    return step.flat.param.detach().double().cpu(), db


def test_packed_records_over_200_steps(amd):
    """200 Adam steps with packed records against the same run with f32 records: what the 18-21-bit
    records cost a TRAINING RUN.  Two f32-record runs are bitwise identical
    (the step is deterministic), so every difference below is the packed format's."""
    p_f32, db_f32 = _train_cfg4_like(amd, 0, 200)
    p_again, db_again = _train_cfg4_like(amd, 0, 200)
    p_pack, db_pack = _train_cfg4_like(amd, 1, 200)
    amd.lib.set_option("bwd_records", 0)
    assert torch.equal(p_f32, p_again) and db_f32 == db_again, "the f32-record run is not reproducible"
    diff = (p_pack - p_f32).abs()
    scale = float(p_f32.abs().max())
    rel_max, rel_l2 = float(diff.max()) / scale, float(diff.norm() / p_f32.norm())
    _report("r3_packed_200_steps.json", dict(psnr_f32_records=db_f32, psnr_packed_records=db_pack,
                                             param_rel_to_max=rel_max, param_rel_l2=rel_l2))
    print(f"200 steps: PSNR f32 records {db_f32:.4f} dB, packed {db_pack:.4f} dB; "
          f"parameters differ by {rel_max:.3e} of max, {rel_l2:.3e} in L2")
    assert abs(db_pack - db_f32) <= 0.01, (db_f32, db_pack)
