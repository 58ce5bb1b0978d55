"""Pin the oracle (oracle/) against golden vectors captured from the reference's own
encoding.py / models.py (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden
from oracle import detrand
from oracle import data as odata
from oracle import hashgrid as ohash
from oracle import mlp as omlp
from oracle import train as otrain

ENC_FIXTURES = ["enc_cfg2", "enc_cfg4", "enc_cfg5_4d", "enc_defaults_2d", "enc_f4_small",
                "enc_v2_hashconfig", "enc_v2_notebook", "enc_v2_cfg5"]


def _ctor_args(ctor):
    base, fin = ctor.get("base_resolution", 16), ctor.get("finest_resolution", 512)
    return (ctor["dim"], ctor.get("n_levels", 16), ctor.get("log2_hashmap_size", 15), base, fin)


def test_hash_ids_bit_exact():
    fx = load_golden("hash_ids")
    assert tuple(fx.meta["primes"]) == ohash.PRIMES
    for dim, size in fx.meta["cases"]:
        idx, want = fx[f"idx_d{dim}_t{size}"], fx[f"ids_d{dim}_t{size}"]
        np.testing.assert_array_equal(ohash.hash_u32(idx, size), want)
        np.testing.assert_array_equal(ohash.hash_torch(torch.from_numpy(idx), size).numpy(), want)


@pytest.mark.parametrize("name", ENC_FIXTURES)
def test_level_geometry(name):
    fx = load_golden(name)
    dim, L, log2t, base, fin = _ctor_args(fx.meta["ctor"])
    res, sizes = ohash.level_geometry(dim, L, log2t, base, fin)
    assert sizes == fx.meta["sizes"]
    want = [r if len(r) == dim else r * dim for r in fx.meta["resolutions"]]
    assert [[float(v) for v in r] for r in res] == want


def test_level_geometry_survey_tables():
    # cfg 2 / cfg 4 / HashConfig level tables measured on the reference.
    res, sizes = ohash.level_geometry(3, 16, 19, 16, 512)
    assert [r[0] for r in res] == [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256,
                                   322, 406, 512]
    assert 2 * sum(sizes) == 10435874
    res, sizes = ohash.level_geometry(3, 16, 19, 16, 16 * 1.4 ** 15)
    assert [r[0] for r in res][-3:] == [1269, 1777, 2489] and 2 * sum(sizes) == 12236382
    res, sizes = ohash.level_geometry(3, 4, 23, (64, 64, 5), (352, 352, 15))
    assert res == [[64, 64, 5], [65, 65, 6], [67, 67, 8], [69, 69, 11]]
    assert sizes == [262144, 274625, 300763, 328509]


@pytest.mark.parametrize("name", ENC_FIXTURES)
def test_encoder_forward_and_table_grad(name):
    fx = load_golden(name)
    dim, L, log2t, base, fin = _ctor_args(fx.meta["ctor"])
    feats = fx.meta["ctor"].get("n_features_per_level", 2)
    res, sizes = ohash.resolutions_for(dim, L, log2t, base, fin)
    tables = ohash.init_tables(sizes, feats, fx.meta["table_seed"], fx.meta["table_scale"])
    for t in tables:
        t.requires_grad_(True)
    out = ohash.encode(torch.from_numpy(fx["x"]), tables, res)
    # same ATen op sequence as the reference on the same torch build: bit-exact
    np.testing.assert_array_equal(out.detach().numpy(), fx["out"])
    out.backward(torch.from_numpy(fx["d_out"]))
    for l, t in enumerate(tables):
        g = t.grad
        nz = torch.nonzero(g.abs().sum(dim=1) != 0).flatten().numpy()
        np.testing.assert_array_equal(nz, fx[f"grad_idx_{l}"])
        assert_close(g[nz].numpy(), fx[f"grad_val_{l}"], 1e-6, f"{name} level {l} grad")


@pytest.mark.parametrize("name", ["enc_f4_small", "enc_v2_hashconfig"])
def test_loop_restatement_matches(name):
    """The torch-free loop statement agrees with the reference output."""
    fx = load_golden(name)
    dim, L, log2t, base, fin = _ctor_args(fx.meta["ctor"])
    feats = fx.meta["ctor"].get("n_features_per_level", 2)
    res, sizes = ohash.level_geometry(dim, L, log2t, base, fin)
    tables = [t.numpy() for t in ohash.init_tables(sizes, feats, fx.meta["table_seed"],
                                                   fx.meta["table_scale"])]
    n = 64
    got = ohash.encode_loops(np.concatenate([fx["x"][:n], fx["x"][-12:]]), tables, res)
    want = np.concatenate([fx["out"][:n], fx["out"][-12:]])
    assert_close(got, want, 1e-6, name)


@pytest.mark.parametrize("name", ["siren_3d_5x256", "siren_2d_4x352", "siren_2d_3x64"])
def test_siren(name):
    fx = load_golden(name)
    m = fx.meta
    model = otrain.SirenModel(m["dim_in"], m["dim_hidden"], 1, m["n_layers"], seed=m["seed"])
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    ps = model.parameters()
    for p in ps:
        p.requires_grad_(True)
    pred = model.forward(x)
    loss = omlp.mse_loss(pred, torch.from_numpy(fx["y"]))
    loss.backward()
    assert_close(pred.detach().numpy(), fx["pred"], 1e-6, "pred")
    assert abs(float(loss.detach()) - float(fx["loss"])) <= 1e-6 * abs(float(fx["loss"]))
    assert_close(x.grad.numpy(), fx["dx"], 1e-6, "dx")
    for i, (w, b) in enumerate(model.params):
        assert_close(b.grad.numpy(), fx[f"gb_{i}"], 1e-6, f"gb{i}")
        gw = w.grad.numpy()
        head = fx[f"gw_head_{i}"]
        assert_close(gw[:head.shape[0]], head, 1e-6, f"gw{i}")
        assert abs(np.linalg.norm(gw.astype(np.float64)) - float(fx[f"gw_norm_{i}"])) \
            <= 1e-6 * float(fx[f"gw_norm_{i}"])


@pytest.mark.parametrize("name", ["relu_mlp_64", "relu_mlp_128"])
@pytest.mark.parametrize("tag", ["act", "lin"])
def test_relu_mlp(name, tag):
    fx = load_golden(name)
    params = omlp.linear_init(fx.meta["dims"], fx.meta["seed"])
    flat = [t for wb in params for t in wb]
    for p in flat:
        p.requires_grad_(True)
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    pred = omlp.relu_mlp_forward(x, params, final_activation=(tag == "act"))
    loss = omlp.mse_loss(pred, torch.from_numpy(fx["y"]))
    loss.backward()
    assert_close(pred.detach().numpy(), fx[f"pred_{tag}"], 1e-6, "pred")
    assert_close(x.grad.numpy(), fx[f"dx_{tag}"], 1e-6, "dx")
    for i, (w, b) in enumerate(params):
        assert_close(w.grad.numpy(), fx[f"gw_{tag}_{i}"], 1e-6, f"gw{i}")
        assert_close(b.grad.numpy(), fx[f"gb_{tag}_{i}"], 1e-6, f"gb{i}")


def test_e2e_hash_adam():
    fx = load_golden("e2e_hash_adam")
    m = fx.meta
    c = m["ctor"]
    model = otrain.HashMlpModel(c["dim"], c["n_levels"], c["n_features_per_level"],
                                c["log2_hashmap_size"], c["base_resolution"],
                                c["finest_resolution"], hidden=m["dims"][1:-1], seed=0)
    model.tables = ohash.init_tables(m["sizes"], c["n_features_per_level"], m["table_seed"],
                                     m["table_scale"])
    model.mlp = omlp.linear_init(m["dims"], m["mlp_seed"])
    opt = None
    for step in range(m["steps"]):
        batch = [(torch.from_numpy(fx[f"x_{step}"]), torch.from_numpy(fx[f"y_{step}"]))]
        losses, opt = otrain.train_steps(model, batch, m["lr"], opt)
        assert abs(losses[0] - float(fx[f"loss_{step}"])) <= 1e-6 * float(fx[f"loss_{step}"])
        for l, t in enumerate(model.tables):
            assert_close(t.numpy(), fx[f"table_{step}_{l}"], 1e-6, f"table {l} step {step}")
        for i, (w, b) in enumerate(model.mlp):
            assert_close(w.numpy(), fx[f"w_{step}_{i}"], 1e-6, f"w{i} step {step}")
            assert_close(b.numpy(), fx[f"b_{step}_{i}"], 1e-6, f"b{i} step {step}")


@pytest.mark.parametrize("name", ["modsiren_2d", "modsiren_3d"])
def test_modulated_siren(name):
    fx = load_golden(name)
    m = fx.meta
    siren = omlp.siren_init(m["dim_in"], m["dim_hidden"], 1, m["n_layers"], m["seed"])
    mod = omlp.modulator_init(m["dim_in"], m["dim_hidden"], m["n_layers"], m["seed"] + 500)
    for w, b in siren + mod:
        w.requires_grad_(True)
        b.requires_grad_(True)
    pred = omlp.modulated_siren_forward(torch.from_numpy(fx["x"]), siren, mod)
    loss = omlp.mse_loss(pred, torch.from_numpy(fx["y"]))
    loss.backward()
    assert_close(pred.detach().numpy(), fx["pred"], 1e-6, "pred")
    assert abs(float(loss.detach()) - float(fx["loss"])) <= 1e-6 * abs(float(fx["loss"]))
    for tag, params in (("siren", siren), ("mod", mod)):
        for i, (w, b) in enumerate(params):
            assert_close(w.grad.numpy(), fx[f"{tag}_gw_{i}"], 1e-6, f"{tag} gw{i}")
            assert_close(b.grad.numpy(), fx[f"{tag}_gb_{i}"], 1e-6, f"{tag} gb{i}")


def test_frequency_encoding():
    fx = load_golden("frequency")
    for dim, n_levels in fx.meta["cases"]:
        x = torch.from_numpy(fx[f"x_{dim}"]).requires_grad_(True)
        out = ohash.frequency_encode(x, n_levels)
        assert out.shape == (x.shape[0], dim * 2 * n_levels)
        np.testing.assert_array_equal(out.detach().numpy(), fx[f"out_{dim}"])
        out.backward(torch.from_numpy(fx[f"g_{dim}"]))
        assert_close(x.grad.numpy(), fx[f"dx_{dim}"], 1e-6, f"dx d{dim}")


def test_e2e_siren_adam():
    fx = load_golden("e2e_siren_adam")
    m = fx.meta
    model = otrain.SirenModel(m["dim_in"], m["dim_hidden"], 1, m["n_layers"], seed=m["seed"])
    opt = None
    for step in range(m["steps"]):
        batch = [(torch.from_numpy(fx[f"x_{step}"]), torch.from_numpy(fx[f"y_{step}"]))]
        losses, opt = otrain.train_steps(model, batch, m["lr"], opt)
        assert abs(losses[0] - float(fx[f"loss_{step}"])) <= 1e-6 * abs(float(fx[f"loss_{step}"]))
        for i, (w, b) in enumerate(model.params):
            assert_close(w.numpy(), fx[f"w_{step}_{i}"], 1e-6, f"w{i} step {step}")
            assert_close(b.numpy(), fx[f"b_{step}_{i}"], 1e-6, f"b{i} step {step}")


def test_e2e_siren256_adam():
    """SIREN at BASELINE config 3's width (3 -> 256 x 5 -> 1): three Adam steps against the
    reference; the GPU test runs the same fixture through the fused chain kernels."""
    fx = load_golden("e2e_siren256_adam")
    m = fx.meta
    model = otrain.SirenModel(m["dim_in"], m["dim_hidden"], 1, m["n_layers"], seed=m["seed"])
    assert_close(model.forward(torch.from_numpy(fx["x_0"])).numpy(), fx["pred_0"], 1e-6, "pred")
    opt = None
    for step in range(m["steps"]):
        batch = [(torch.from_numpy(fx[f"x_{step}"]), torch.from_numpy(fx[f"y_{step}"]))]
        losses, opt = otrain.train_steps(model, batch, m["lr"], opt)
        assert abs(losses[0] - float(fx[f"loss_{step}"])) <= 1e-6 * abs(float(fx[f"loss_{step}"]))
        for i, (w, b) in enumerate(model.params):
            head = fx[f"w_{step}_{i}"]
            # Adam's first steps are lr * g / (|g| + eps): where |g| ~ eps a last-bit difference
            # of the gradient (torch's fused mm backward vs the oracle's) moves the weight by a
            # fraction of lr = 1e-4, i.e. ~1.5e-6 of max |w| = 0.15 here: 5e-6, not 1e-6
            assert_close(w.numpy()[:head.shape[0]], head, 5e-6, f"w{i} step {step}")
            assert abs(np.linalg.norm(w.numpy().astype(np.float64)) - float(fx[f"wnorm_{step}_{i}"])) \
                <= 1e-6 * float(fx[f"wnorm_{step}_{i}"])
            assert_close(b.numpy(), fx[f"b_{step}_{i}"], 5e-6, f"b{i} step {step}")


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_hashmlp_as_intended(mode):
    fx = load_golden("hashmlp_intended")
    m = fx.meta
    res, sizes = ohash.resolutions_for(3, 4, 23, (64, 64, 5), (352, 352, 15))
    assert sizes == m["sizes"]
    tables = ohash.init_tables(sizes, 1, m["table_seed"], m["table_scale"])
    params = omlp.linear_init(m["dims"], m["mlp_seed"])
    bn = []
    for i, (w, _) in enumerate(params):
        n = w.shape[0]
        if mode == "train":
            st = dict(running_mean=torch.zeros(n), running_var=torch.ones(n))
        else:  # eval uses the running stats the train-mode pass left behind
            st = dict(running_mean=torch.from_numpy(fx[f"bn_mean_{i}"]),
                      running_var=torch.from_numpy(fx[f"bn_var_{i}"]))
        st.update(weight=torch.ones(n), bias=torch.zeros(n))
        bn.append(st)
    z = ohash.encode(torch.from_numpy(fx["x"]), tables, res)
    pred = omlp.hashmlp_decoder_forward(z, params, bn, training=(mode == "train"))
    assert_close(pred.numpy(), fx[f"pred_{mode}"], 1e-6, mode)


def test_hashmlp_gelu_notebook_decoder():
    """Notebook cell-37 HashMLP: encoder + Linear -> GELU blocks (no BatchNorm), forward, loss,
    every gradient and two Adam steps, against the reference modules' outputs."""
    fx = load_golden("hashmlp_gelu_notebook")
    m, c = fx.meta, fx.meta["ctor"]
    res, sizes = ohash.resolutions_for(c["dim"], c["n_levels"], c["log2_hashmap_size"],
                                       tuple(c["base_resolution"]), tuple(c["finest_resolution"]))
    assert sizes == m["sizes"]
    tables = ohash.init_tables(sizes, c["n_features_per_level"], m["table_seed"], m["table_scale"])
    params = omlp.linear_init(m["dims"], m["mlp_seed"])
    flat = list(tables) + [t for wb in params for t in wb]
    opt = omlp.Adam(flat, lr=m["lr"])
    for step in range(m["steps"]):
        for p in flat:
            p.requires_grad_(True)
            p.grad = None
        z = ohash.encode(torch.from_numpy(fx[f"x_{step}"]), tables, res)
        pred = omlp.gelu_mlp_forward(z, params, final_activation=True)
        loss = omlp.mse_loss(pred, torch.from_numpy(fx[f"y_{step}"]))
        loss.backward()
        assert_close(pred.detach().numpy(), fx[f"pred_{step}"], 1e-6, f"pred step {step}")
        assert abs(float(loss.detach()) - float(fx[f"loss_{step}"])) <= 1e-6 * float(fx[f"loss_{step}"])
        if step == 0:
            for l, t in enumerate(tables):
                nz = torch.nonzero(t.grad.abs().sum(dim=1) != 0).flatten().numpy()
                np.testing.assert_array_equal(nz, fx[f"grad_idx_{l}"])
                assert_close(t.grad[nz].numpy(), fx[f"grad_val_{l}"], 1e-6, f"table grad {l}")
            for i, (w, b) in enumerate(params):
                assert_close(w.grad.numpy(), fx[f"gw_{i}"], 1e-6, f"gw{i}")
                assert_close(b.grad.numpy(), fx[f"gb_{i}"], 1e-6, f"gb{i}")
        grads = [p.grad for p in flat]
        for p in flat:
            p.requires_grad_(False)
        opt.step(grads)
        for i, (w, b) in enumerate(params):
            assert_close(w.numpy(), fx[f"w_{step}_{i}"], 1e-6, f"w{i} step {step}")
            assert_close(b.numpy(), fx[f"b_{step}_{i}"], 1e-6, f"b{i} step {step}")
        for l, t in enumerate(tables):
            assert_close(t.numpy()[fx[f"grad_idx_{l}"]], fx[f"table_{step}_{l}"], 1e-6,
                         f"table {l} step {step}")


def _bn_state(m, dims):
    """BatchNorm1d members of the reference's decoder blocks as the fixture initialised them."""
    out = []
    for fan_out in dims[1:]:
        out.append(dict(weight=torch.from_numpy(detrand.uniform(fan_out, m["bn_seeds"][0], 0.5, 1.5).copy()),
                        bias=torch.from_numpy(detrand.uniform(fan_out, m["bn_seeds"][1], -0.2, 0.2).copy()),
                        running_mean=torch.zeros(fan_out), running_var=torch.ones(fan_out)))
    return out


def test_hashmlp_batchnorm_decoder_gradients_and_adam():
    """The reference's DEFAULT model (config.model_cls = HashMLP): encoder + Linear -> BatchNorm1d -> GELU
    -> Dropout(0) blocks in train() mode (models.py:712-739, in sequence per SURVEY Q1) -- forward, loss,
    every gradient incl. BatchNorm weight / bias, the running statistics, two Adam steps, and the
    eval-mode prediction with the statistics those steps left: against the reference modules' outputs."""
    fx = load_golden("hashmlp_bn_adam")
    m, c = fx.meta, fx.meta["ctor"]
    res, sizes = ohash.resolutions_for(c["dim"], c["n_levels"], c["log2_hashmap_size"],
                                       tuple(c["base_resolution"]), tuple(c["finest_resolution"]))
    assert sizes == m["sizes"]
    tables = ohash.init_tables(sizes, c["n_features_per_level"], m["table_seed"], m["table_scale"])
    params = omlp.linear_init(m["dims"], m["mlp_seed"])
    bn = _bn_state(m, m["dims"])
    flat = list(tables) + [t for wb in params for t in wb] + [s[k] for s in bn for k in ("weight", "bias")]
    opt = omlp.Adam(flat, lr=m["lr"])
    for step in range(m["steps"]):
        for p in flat:
            p.requires_grad_(True)
            p.grad = None
        b_before = [b.detach().numpy().copy() for _, b in params]  # the biases this step's batch means hold
        z = ohash.encode(torch.from_numpy(fx[f"x_{step}"]), tables, res)
        pred = omlp.hashmlp_decoder_forward(z, params, bn, training=True)
        loss = omlp.mse_loss(pred, torch.from_numpy(fx[f"y_{step}"]))
        loss.backward()
        assert_close(pred.detach().numpy(), fx[f"pred_{step}"], 1e-6, f"pred step {step}")
        assert abs(float(loss.detach()) - float(fx[f"loss_{step}"])) <= 1e-6 * float(fx[f"loss_{step}"])
        if step == 0:
            for l, t in enumerate(tables):
                nz = torch.nonzero(t.grad.abs().sum(dim=1) != 0).flatten().numpy()
                np.testing.assert_array_equal(nz, fx[f"grad_idx_{l}"])
                assert_close(t.grad[nz].numpy(), fx[f"grad_val_{l}"], 2e-6, f"table grad {l}")
            for i, ((w, b), s) in enumerate(zip(params, bn)):
                assert_close(w.grad.numpy(), fx[f"gw_{i}"], 2e-6, f"gw{i}")
                assert_close(s["weight"].grad.numpy(), fx[f"bn_gw_{i}"], 2e-6, f"bn gw{i}")
                assert_close(s["bias"].grad.numpy(), fx[f"bn_gb_{i}"], 2e-6, f"bn gb{i}")
                # a Linear bias in front of a train-mode BatchNorm has an exactly-zero gradient in exact
                # arithmetic (the batch mean absorbs it): what both sides hold is rounding noise, which
                # Adam then turns into steps of ~lr in a direction no two evaluations share -- with no
                # effect on any output (the next batch mean absorbs the bias again).  Hence no assert on
                # the Linear biases after Adam, here or in the GPU test.
                assert np.abs(fx[f"gb_{i}"]).max() <= 1e-5 * np.abs(fx[f"gw_{i}"]).max()
                assert float(b.grad.abs().max()) <= 1e-5 * float(w.grad.abs().max())
        grads = [p.grad for p in flat]
        for p in flat:
            p.requires_grad_(False)
        opt.step(grads)
        for i, ((w, b), s) in enumerate(zip(params, bn)):
            assert_close(w.numpy(), fx[f"w_{step}_{i}"], 1e-5, f"w{i} step {step}")
            assert_close(s["weight"].numpy(), fx[f"bn_w_{step}_{i}"], 1e-6, f"bn w{i} step {step}")
            assert_close(s["bias"].numpy(), fx[f"bn_b_{step}_{i}"], 1e-6, f"bn b{i} step {step}")
            # running mean = momentum-weighted batch means of W z + b: b's noise-driven Adam steps (see
            # above) enter it from the second step on, so compare it with this side's own bias taken out
            b_ref = fx[f"b_{step - 1}_{i}"] if step else omlp.linear_init(m["dims"], m["mlp_seed"])[i][1].numpy()
            b_own = b_before[i]
            own = s["running_mean"].numpy() - 0.1 * b_own
            ref = fx[f"bn_mean_{step}_{i}"] - 0.1 * b_ref
            assert np.abs(own - ref).max() <= 1e-5 * np.abs(fx[f"bn_mean_{step}_{i}"]).max(), \
                f"bn mean{i} step {step}: {np.abs(own - ref).max():.3e}"
            assert_close(s["running_var"].numpy(), fx[f"bn_var_{step}_{i}"], 1e-5, f"bn var{i} step {step}")
        for l, t in enumerate(tables):
            assert_close(t.numpy()[fx[f"grad_idx_{l}"]], fx[f"table_{step}_{l}"], 1e-5,
                         f"table {l} step {step}")
    # eval mode (predict_step's forward) from the REFERENCE's final state -- every tensor it needs for
    # x_0 is in the fixture (the table rows x_0 touches are the rows of grad_idx_*)
    last = m["steps"] - 1
    for l, t in enumerate(tables):
        t[torch.from_numpy(fx[f"grad_idx_{l}"].astype(np.int64))] = torch.from_numpy(fx[f"table_{last}_{l}"])
    params = [(torch.from_numpy(fx[f"w_{last}_{i}"]), torch.from_numpy(fx[f"b_{last}_{i}"]))
              for i in range(len(params))]
    bn = [dict(weight=torch.from_numpy(fx[f"bn_w_{last}_{i}"]), bias=torch.from_numpy(fx[f"bn_b_{last}_{i}"]),
               running_mean=torch.from_numpy(fx[f"bn_mean_{last}_{i}"]),
               running_var=torch.from_numpy(fx[f"bn_var_{last}_{i}"])) for i in range(len(bn))]
    z = ohash.encode(torch.from_numpy(fx["x_0"]), tables, res)
    pred = omlp.hashmlp_decoder_forward(z, params, bn, training=False)
    assert_close(pred.numpy(), fx["pred_eval_after"], 1e-6, "eval-mode prediction from the reference's final state")


def test_sample_volume_dataset_4d():
    """BASELINE config 5's workload: the whole sample volume through the oracle's MriImage
    restatement (datamodules.py:135-166) -- shape, normalisation, grid order, and agreement
    with the one slice fixture of round 1."""
    fx = load_golden("sample_volume")
    assert fx.meta["shape"] == [352, 352, 6, 15] and fx["raw_int16"].dtype == np.int16
    vol = (fx["raw_int16"].astype(np.float64) * fx.meta["scl_slope"]
           + fx.meta["scl_inter"]).astype(np.float32)
    sl = load_golden("sample_slice_z3_t7")
    np.testing.assert_array_equal(fx["raw_int16"][:, :, 3, 7], sl["raw_int16"])
    assert int(fx["raw_int16"].min()) == 0 and int(fx["raw_int16"].max()) == 91
    coords, pix = odata.dataset(vol)
    assert coords.shape == (11151360, 4) and pix.shape == (11151360, 1)
    assert float(pix.min()) == 0.0 and float(pix.max()) == 1.0
    t = torch.linspace(0, 1, 15)
    assert torch.equal(coords[:15, 3], t)                       # last axis fastest
    assert torch.equal(coords[:15 * 6:15, 2], torch.linspace(0, 1, 6))
    assert torch.equal(coords[::352 * 6 * 15, 0], torch.linspace(0, 1, 352))
    # held-out-frame protocol (interp.py:27,35): even frames of the FULL time grid
    even = coords.view(352, 352, 6, 15, 4)[..., ::2, :]
    assert even.shape[3] == 8 and torch.equal(even[0, 0, 0, :, 3], t[::2])


def test_data_grid_and_normalisation():
    fx = load_golden("sample_slice_z3_t7")
    raw = fx["raw_int16"].astype(np.float32) * np.float32(fx.meta["scl_slope"])
    coords, pix = odata.dataset(raw)
    assert coords.shape == (352 * 352, 2) and pix.shape == (352 * 352, 1)
    assert float(pix.min()) == 0.0 and float(pix.max()) == 1.0
    lin = torch.linspace(0, 1, 352)
    # C-order flatten, last axis fastest (datamodules.py:148,162-163)
    assert torch.equal(coords[:352, 1], lin) and torch.equal(coords[::352, 0], lin)
    c2, p2 = odata.dataset(raw, norm_siren=True)
    assert float(c2.min()) == -1.0 and float(p2.min()) == -1.0 and float(p2.max()) == 1.0


def test_phantom_and_slabs():
    v = odata.phantom((16, 12, 10))
    assert v.dtype == np.float32 and v.min() == 0.0 and v.max() == 1.0
    spans = [odata.slab_range(6, r, 4) for r in range(4)]
    assert spans == [(0, 2), (2, 4), (4, 5), (5, 6)]
