"""Round-4 parity edges of the gfx950 HIP path (through the C ABI).  Needs an MI355X.

* the three-term bf16 products (csrc/bf16x3.h) with operands at 2^-100 ... 2^+100 -- "no range assumption" is
  a claim, this is its test -- for the fused decoder (reference models.py:46-66) and a SIREN chain
  (models.py:153-156, 230-233), against the same network evaluated in float64;
* the hash ids themselves (reference encoding.py:69-78): a one-feature table that holds its own slot index,
  queried exactly on grid nodes, so that the forward output IS `fast_hash` -- compared bit for bit with the
  ids the reference produced (tests/golden/hash_ids.npz), negative cells and non-power-of-two tables included.
"""
import numpy as np
import pytest
import torch

from conftest import REL_TOL, load_golden
from oracle import detrand
from oracle import hashgrid as ohash
from oracle import mlp as omlp

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    """conftest.rel_err without its 1e-30 floor on the denominators: the tensors of this file live at 2^-100
    (7.9e-31) and below, where that floor would hide a factor; the reference must not vanish instead."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape and np.abs(b).max() > 0.0
    return np.abs(a - b).max() / np.abs(b).max(), np.linalg.norm(a - b) / np.linalg.norm(b)


def assert_close(a, b, tol=REL_TOL, what=""):
    e_max, e_l2 = rel_err(a, b)
    assert e_max <= tol and e_l2 <= tol, f"{what}: rel-to-max {e_max:.3e}, rel-L2 {e_l2:.3e} > {tol}"


@pytest.fixture(scope="module")
def amd():
    import mri_interpolation_amd as pkg
    from mri_interpolation_amd import _lib, datamodules, encoding, models, ops, optim, trainer
    assert torch.cuda.is_available(), "these tests need the GPU"
    _lib.load()  # fails loudly if libmri_inr.so is missing
    return type("NS", (), dict(pkg=pkg, lib=_lib, ops=ops, encoding=encoding, models=models,
                               optim=optim, trainer=trainer, datamodules=datamodules))


# ----------------------------------------------------------------------- hash ids, directly
@pytest.mark.parametrize("feature_major", [False, True])
def test_hash_ids_on_the_gpu_equal_the_references(amd, feature_major):
    """fast_hash on the GPU, bit for bit.  Level table: T rows of ONE feature, row s holding float(s) (exact:
    T <= 2^19 < 2^24).  Coordinates x = idx / 4096 for the golden's integer cells idx (-2490 ... 2^20 + 3, negative
    ones included; coordinates beyond 1 are extrapolated cells like any other) on a grid of resolution 4096: x and
    pos = x * 4096 = idx are exact in f32 (|idx| < 2^22), the fraction
    is 0, so corner 0 (every axis at its floor vertex) has weight exactly 1 and every other corner weight
    exactly 0: out = table[hash(idx)] * 1 + sum(table[.] * 0) = the hash id, with no rounding anywhere.
    Covers uint32 wrap-multiplication, xor, negative cells and the true `% T` of non-power-of-two tables."""
    fx = load_golden("hash_ids")
    for dim, size in fx.meta["cases"]:
        idx, want = fx[f"idx_d{dim}_t{size}"], fx[f"ids_d{dim}_t{size}"]
        assert np.abs(idx).max() < 1 << 22 and size <= 1 << 24
        x = (idx.astype(np.float64) / 4096.0).astype(np.float32)
        assert np.array_equal(x.astype(np.float64) * 4096.0, idx)  # exact by construction
        desc = amd.ops.make_grid_desc(dim, [[4096.0] * dim], [size], 1)
        table = torch.arange(size, dtype=torch.float32, device="cuda").reshape(size, 1)
        out = amd.ops.hashgrid_forward(desc, torch.from_numpy(x).cuda(), table, feature_major=feature_major)
        got = out.reshape(-1).cpu().numpy()
        assert np.array_equal(got, np.rint(got)), (dim, size)
        np.testing.assert_array_equal(got.astype(np.int64), want, err_msg=f"D={dim} T={size}")
        # the oracle restatement gives the same ids for the same cells (pins the test's own reading of idx)
        np.testing.assert_array_equal(ohash.hash_u32(idx, size), want)


def test_hash_ids_of_every_corner_on_the_gpu(amd):
    """The other 2^D - 1 corners: with x half a cell off the node on every axis each corner has weight 2^-D
    exactly, so out * 2^D = sum of the 2^D corner ids -- compared with the oracle's ids of cell + offset."""
    fx = load_golden("hash_ids")
    for dim, size in fx.meta["cases"]:
        idx = fx[f"idx_d{dim}_t{size}"]
        idx = idx[(idx >= 0).all(axis=1)][:128]  # (truncation toward zero: the floor vertex of -k + 0.5 is -k + 1)
        x = ((idx.astype(np.float64) + 0.5) / 4096.0).astype(np.float32)
        assert np.array_equal(x.astype(np.float64) * 4096.0, idx + 0.5)
        want = np.zeros(len(idx), dtype=np.int64)
        for corner in range(1 << dim):
            off = np.array([(corner >> d) & 1 for d in range(dim)], dtype=np.int64)
            want += ohash.hash_u32(idx + off, size)
        desc = amd.ops.make_grid_desc(dim, [[4096.0] * dim], [size], 1)
        table = torch.arange(size, dtype=torch.float32, device="cuda").reshape(size, 1)
        out = amd.ops.hashgrid_forward(desc, torch.from_numpy(x).cuda(), table).reshape(-1).double().cpu().numpy()
        # every product id * 2^-D is exact; the f32 sum of 2^D of them is exact while it stays below 2^24 / 2^-D
        np.testing.assert_array_equal(np.rint(out * (1 << dim)).astype(np.int64), want, err_msg=f"D={dim} T={size}")


# ----------------------------------------------------------------------- three-term products, extreme scales
SCALES = [-100, -60, 60, 100]


def _decoder_case(k_in, hidden, n, s, seed):
    """in -> H -> H -> 1 ReLU decoder whose first layer is scaled by 2^s and whose output layer by 2^-s: hidden
    activations live at 2^s, the gradients flowing back through them at 2^-s, prediction and loss at O(1);
    dW2 = dz2^T h1 multiplies 2^-s by 2^s, dw3 sits at 2^s, dW1 / db1 / db2 at 2^-s.  The ReLU network is
    positively homogeneous, so in exact arithmetic this is the unscaled network: float64 is the reference."""
    params = omlp.linear_init([k_in, hidden, hidden, 1], seed)
    (w1, b1), (w2, b2), (w3, b3) = params
    up, down = float(2.0 ** s), float(2.0 ** -s)
    params = [(w1 * up, b1 * up), (w2, b2 * up), (w3 * down, b3)]
    x = torch.from_numpy(detrand.uniform(n * k_in, seed + 1, -1, 1).reshape(n, k_in))
    t = torch.from_numpy(detrand.uniform(n, seed + 2, 0, 1).reshape(n, 1))
    return params, x, t


@pytest.mark.parametrize("s", SCALES)
@pytest.mark.parametrize("k_in,hidden", [(32, 128), (32, 64)])
def test_decoder_products_at_extreme_scales(amd, k_in, hidden, s):
    """tiny_mlp forward + MSE + backward (csrc/mlp_x3.hip: every product six bf16 MFMAs on exact three-term
    splits) with operands at 2^+-60 and 2^+-100 against float64, same relative tolerance as everywhere.
    bf16 has f32's exponent range, so the split x = h + m + l needs no scaling: it stays exact while the
    smallest term l ~ 2^-16 |x| is a NORMAL bf16 (|x| >= 2^-110) and h does not round up to infinity
    (|x| <= bf16 max = 0.996 * f32 max); test_where_the_three_term_split_stops_being_exact walks past both."""
    ops = amd.ops
    n = 1500
    params, x, t = _decoder_case(k_in, hidden, n, s, 40 + k_in + hidden)
    p64 = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in params]
    x64 = x.double().requires_grad_(True)
    y64 = omlp.relu_mlp_forward(x64, p64, final_activation=False)
    loss64 = omlp.mse_loss(y64, t.double())
    loss64.backward()

    gp = [(w.cuda(), b.cuda()) for w, b in params]
    grads = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in gp]
    x_fm = x.t().contiguous().cuda()
    d_x = torch.empty_like(x_fm)
    y_gpu = torch.empty(n, 1, device="cuda")
    loss_gpu = torch.zeros(1, device="cuda")
    ops.tiny_mlp_train(x_fm, t.cuda(), gp, grads, loss_gpu, d_x=d_x, y=y_gpu)
    for g in [y_gpu, d_x] + [g for pair in grads for g in pair]:
        assert torch.isfinite(g).all()
    assert_close(y_gpu.cpu().numpy(), y64.detach().numpy(), REL_TOL, f"y at 2^{s}")
    assert abs(float(loss_gpu) - float(loss64)) <= REL_TOL * float(loss64)
    assert_close(d_x.t().cpu().numpy(), x64.grad.numpy(), REL_TOL, f"dx at 2^{s}")
    for i, ((gw, gb), (w, b)) in enumerate(zip(grads, p64)):
        assert_close(gw.cpu().numpy(), w.grad.numpy(), REL_TOL, f"dW{i + 1} at 2^{s}")
        assert_close(gb.cpu().numpy(), b.grad.numpy(), REL_TOL, f"db{i + 1} at 2^{s}")
    assert torch.equal(ops.tiny_mlp_forward(x_fm, gp), y_gpu)  # the inference kernel: same bits


@pytest.mark.parametrize("s", [60, 100])
def test_siren_chain_products_at_extreme_scales(amd, s):
    """SirenNet 3 -> 64 x 3 -> 1 on the fused chain kernels (csrc/siren_chain.hip) with the second sine layer
    scaled by 2^-s (weights and bias: its activation sin(w0 z) ~ w0 z sits at 2^-s) and the third layer's
    WEIGHTS by 2^+s (its pre-activation is O(1) again): forward products of 2^+s by 2^-s, dz of the second
    layer at 2^+s, its weight gradient at 2^+s, the third layer's at 2^-s.  (The mirror image -- a sine layer
    scaled UP -- is sin of 1e30: no evaluation is meaningful there, the reference's included.)"""
    hidden, n_layers, n = 64, 3, 1200
    net = amd.models.SirenNet(dim_in=3, dim_hidden=hidden, dim_out=1, n_layers=n_layers)
    params = omlp.siren_init(3, hidden, 1, n_layers, 77)
    up, down = float(2.0 ** s), float(2.0 ** -s)
    params[1] = (params[1][0] * down, params[1][1] * down)
    params[2] = (params[2][0] * up, params[2][1])
    with torch.no_grad():
        for layer, (w, b) in zip(list(net.layers) + [net.last_layer], params):
            layer.weight.copy_(w)
            layer.bias.copy_(b)
    net.cuda()
    x = torch.from_numpy(detrand.uniform(n * 3, 5, -1, 1).reshape(n, 3))
    t = torch.from_numpy(detrand.uniform(n, 6, -1, 1).reshape(n, 1))
    p64 = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in params]
    y64 = omlp.siren_forward(x.double(), p64)
    loss64 = omlp.mse_loss(y64, t.double())
    loss64.backward()

    step = amd.trainer.FusedStep(net, net.configure_optimizers())
    assert step.use_chain, "this test is about the fused chain kernels"
    pred, ws = step.forward(x.cuda(), train=True)
    step.backward(x.cuda(), t.cuda(), ws)
    assert_close(pred.detach().cpu().numpy(), y64.detach().numpy(), REL_TOL, f"pred at 2^{s}")
    assert abs(float(step.loss) - float(loss64)) <= REL_TOL * float(loss64)
    for i, (layer, (w, b)) in enumerate(zip(list(net.layers) + [net.last_layer], p64)):
        assert torch.isfinite(layer.weight.grad).all()
        assert_close(layer.weight.grad.cpu().numpy(), w.grad.numpy(), REL_TOL, f"gw{i} at 2^{s}")
        assert_close(layer.bias.grad.cpu().numpy(), b.grad.numpy(), REL_TOL, f"gb{i} at 2^{s}")


def test_where_the_three_term_split_stops_being_exact(amd, capsys):
    """Documents the edges instead of leaving them to be found: the decoder's forward pass with the
    first layer (and the biases) scaled by 2^s, s from -118 to +120, prediction at 2^s.  Asserted: f32-accurate
    (1e-5; in practice ~1e-7) for -100 <= s <= 120.  Below that the third term l ~ 2^-16 |x|, then the second,
    m ~ 2^-8 |x|, drop under the smallest NORMAL bf16 (2^-126): the product degrades towards a two-term (2^-16) and
    finally a one-term (2^-8) one, i.e. f32 accuracy ends where |x| < 2^-110 -- sixteen binades above the point
    where f32 itself runs out (subnormals).  The sweep prints what the hardware does there; only finiteness and
    2^-7 are asserted.  Upwards nothing degrades until a value exceeds the largest bf16 (0.996 of the largest f32),
    where h rounds to infinity."""
    ops = amd.ops
    k_in, hidden, n = 32, 128, 512
    base = omlp.linear_init([k_in, hidden, hidden, 1], 91)
    x = torch.from_numpy(detrand.uniform(n * k_in, 92, -1, 1).reshape(n, k_in))
    x_fm = x.t().contiguous().cuda()
    report = []
    for s in (-118, -114, -110, -106, -102, -100, -60, 0, 60, 100, 120):
        up = float(2.0 ** s)
        (w1, b1), (w2, b2), (w3, b3) = base
        params = [(w1 * up, b1 * up), (w2, b2 * up), (w3, b3 * up)]
        y64 = omlp.relu_mlp_forward(x.double(), [(w.double(), b.double()) for w, b in params], final_activation=False)
        y = ops.tiny_mlp_forward(x_fm, [(w.cuda(), b.cuda()) for w, b in params])
        assert torch.isfinite(y).all(), s
        e_max, e_l2 = rel_err(y.cpu().numpy(), y64.numpy())
        report.append((s, e_max, e_l2))
        assert e_max <= (REL_TOL if s >= -100 else 2.0 ** -7), (s, e_max)
    with capsys.disabled():
        print("\nthree-term product, decoder forward, operands scaled by 2^s: "
              + ", ".join(f"s={s}: {a:.1e}" for s, a, _ in report))


# ----------------------------------------------------------------------- table gradient: records routed as pairs
def _table_gradient_case(amd, dim, resolutions, sizes, x, seed, records):
    """Binned table gradient (method 2, records f32 or packed) of an F = 2 grid against the float64 yardstick
    and the global-atomic kernel; returns the binned result."""
    n_levels = len(sizes)
    desc = amd.ops.make_grid_desc(dim, resolutions, sizes, 2)
    n = x.shape[0]
    d_out = torch.from_numpy(detrand.uniform(n * n_levels * 2, seed, -1, 1).reshape(n, n_levels * 2))
    want = ohash.table_gradient_f64(x, d_out, sizes, resolutions, 2)
    rows = sum(sizes)
    amd.lib.set_option("bwd_records", records)
    try:
        got = amd.ops.hashgrid_backward(desc, x.cuda(), d_out.cuda(), torch.zeros(rows, 2, device="cuda"), method=2)
        again = amd.ops.hashgrid_backward(desc, x.cuda(), d_out.cuda(), torch.zeros(rows, 2, device="cuda"), method=2)
    finally:
        amd.lib.set_option("bwd_records", 0)
    atomic = amd.ops.hashgrid_backward(desc, x.cuda(), d_out.cuda(), torch.zeros(rows, 2, device="cuda"), method=1)
    assert torch.equal(got, again), "the binned table gradient is bitwise reproducible"
    off = 0
    for l, (total, mag, count) in enumerate(want):
        g = got[off:off + sizes[l]].double().cpu()
        bound = (2.0 ** -24 if records == 0 else 2.0 ** -17) * (mag + total.abs()) + 1e-30
        assert bool(((g - total).abs() <= bound + 2.0 ** -40 * mag.max()).all()), f"level {l}: a slot misses its bound"
        assert bool((g[count == 0] == 0).all()), f"level {l}: a slot nothing hashes to holds a gradient"
        assert_close(atomic[off:off + sizes[l]].cpu().numpy(), total.numpy(), REL_TOL, f"atomic level {l}")
        off += sizes[l]
    return got


@pytest.mark.parametrize("records", [0, 1])
def test_pair_routing_where_the_two_slots_lie_in_different_slices(amd, records):
    """The table gradient routes the two corners that differ on axis 0 as ONE pair (hashgrid_bwd.hip).  Their slots
    lie in the same 8192-slot slice unless the axis-0 cell index ends in thirteen ones, the cell is -1 (0xFFFFFFFF
    -> 0), or the wrap of `% T` falls between them; then each bin gets a record and a zero pad.  Cases: a grid of
    resolution 20000 queried around cell 8191 / 16383; coordinates one cell outside the grid on axis 0 (every pair
    splits: twice the records, the workgroups write straight to HBM); a non-power-of-two table; D = 2 ... 4."""
    rng = np.random.default_rng(5)
    # (a) D = 3, T = 2^19 (64 slices), cells 8190 .. 8192 and 16382 .. 16384 on axis 0
    n = 3000
    x = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    x[: n // 2, 0] = ((8190 + rng.uniform(0, 3, n // 2)) / 20000.0).astype(np.float32)
    x[n // 2:, 0] = ((16382 + rng.uniform(0, 3, n - n // 2)) / 20000.0).astype(np.float32)
    _table_gradient_case(amd, 3, [[20000.0] * 3], [1 << 19], torch.from_numpy(x), 11, records)
    # (b) every coordinate in cell -1 of axis 0 on the first level (pos in (-1.95, -1.05) truncates toward zero to -1:
    # "floor" vertex 0xFFFFFFFF, upper vertex 0 -- every pair splits), cells -1 .. -3 on the second
    for dim in (2, 3, 4):
        n = 5000
        x = rng.uniform(0, 1, (n, dim)).astype(np.float32)
        x[:, 0] = (-(1.0 + rng.uniform(0.05, 0.95, n)) / 64.0).astype(np.float32)
        _table_gradient_case(amd, dim, [[64.0] * dim, [100.0] * dim], [1 << 17, 1 << 17], torch.from_numpy(x), 12 + dim,
                             records)
    # (c) non-power-of-two tables (true `% T`): pairs around the wrap; a mix of inside / outside coordinates
    n = 20000
    x = rng.uniform(-0.2, 1.2, (n, 3)).astype(np.float32)
    _table_gradient_case(amd, 3, [[61.0] * 3, [86.0] * 3, [330.0] * 3], [226981, 300763, 328509], torch.from_numpy(x), 21,
                         records)


def test_pair_routing_with_more_than_64_slices(amd):
    """T = 2^20 and 2^21: 128 / 256 slices per level -- the copy-out goes bin by bin (the slot word has room for a
    6-bit bin id only) -- beside a 2^19 level that takes the linear copy-out, in one call."""
    rng = np.random.default_rng(6)
    n = 30000
    x = torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    _table_gradient_case(amd, 3, [[512.0] * 3, [700.0] * 3, [900.0] * 3], [1 << 19, 1 << 20, 1 << 21], x, 31, 0)
