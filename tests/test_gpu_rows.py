"""GPU tests of the rows-in-registers SIREN chain kernels (csrc/siren_rows.hip, SirenNet of width 256, round 4): against
the float64 oracle, against the LDS-image kernels of csrc/siren_chain.hip (option "siren_rows" 0), on ragged row counts
(a 128-row group cut anywhere), and on the cold path of the branch-free sincos (arguments beyond 8192).

Reference semantics: models.py:153-156 SirenLayer.forward (`sin(w0 * F.linear(x))`), :230-233 SirenNet.forward, :61-70
training_step (F.mse_loss) -- autograd of those for the gradients.
"""
import numpy as np
import pytest
import torch

from conftest import REL_TOL, assert_close
from oracle import detrand
from oracle import mlp as omlp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    from mri_interpolation_amd import _lib, models, trainer
    assert torch.cuda.is_available(), "these tests need the GPU"
    _lib.load()
    return type("NS", (), dict(lib=_lib, models=models, trainer=trainer))


def _net(amd, dim_in, n_layers, seed, scale_first=1.0, scale_hidden=1.0):
    net = amd.models.SirenNet(dim_in, 256, 1, n_layers)
    params = omlp.siren_init(dim_in, 256, 1, n_layers, seed)
    params[0] = (params[0][0] * scale_first, params[0][1])
    if n_layers > 1:
        params[1] = (params[1][0] * scale_hidden, params[1][1])
    with torch.no_grad():
        for layer, (w, b) in zip(list(net.layers) + [net.last_layer], params):
            layer.weight.copy_(w)
            layer.bias.copy_(b)
    return net.cuda(), params


def _step(amd, net, x, y):
    """One fused pass (loss-mode forward, backward chain, weight gradients): prediction, loss, flat gradient."""
    st = amd.trainer.FusedStep(net, net.configure_optimizers())
    assert st.use_chain and st.chain_loss
    st._chain_loss_pass(x, y, True, 1.0)
    ws = st._workspace(x.shape[0], True)
    return ws["y"][-1].clone(), float(st.loss), st.flat.grad.clone(), st


@pytest.mark.parametrize("dim_in,n_layers,n", [(3, 5, 1), (3, 5, 127), (3, 5, 128), (3, 5, 129), (2, 2, 300),
                                               (4, 3, 4097), (3, 5, 70001), (6, 3, 1000), (1, 8, 257)])
def test_rows_kernels_against_oracle_and_image_kernels(amd, dim_in, n_layers, n):
    """Loss-mode forward + backward chain + weight gradients of the rows kernels: prediction, loss and every gradient
    against the float64 oracle at 1e-5, and against the LDS-image kernels (same arithmetic up to summation order) at
    2e-6.  Row counts cut the 128-row groups and the 32-row wave shares everywhere; dim_in = 6 takes the image
    backward (the rows backward serves dim_in <= 4) with the rows forward in its plain training form; 8 layers is
    MRI_SIREN_MAX_LAYERS."""
    lib = amd.lib.load()
    net, params = _net(amd, dim_in, n_layers, 900 + n_layers + dim_in)
    x = torch.from_numpy(detrand.uniform(n * dim_in, n + 3, -1.0, 1.0).reshape(n, dim_in)).cuda()
    y = torch.from_numpy(detrand.uniform(n, n + 4, -1.0, 1.0).reshape(n, 1)).cuda()
    p64 = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in params]
    y64 = omlp.siren_forward(x.cpu().double(), p64)
    loss64 = omlp.mse_loss(y64, y.cpu().double())
    loss64.backward()
    want = np.concatenate([np.concatenate([w.grad.numpy().ravel(), b.grad.numpy().ravel()]) for w, b in p64])
    try:
        pred, loss, grad, st = _step(amd, net, x, y)
        assert_close(pred.cpu().numpy(), y64.detach().numpy(), REL_TOL, "prediction")
        assert abs(loss - float(loss64)) <= REL_TOL * max(float(loss64), 0.1)
        flat = np.concatenate([np.concatenate([l.weight.grad.cpu().numpy().ravel(), l.bias.grad.cpu().numpy().ravel()])
                               for l in list(net.layers) + [net.last_layer]])
        assert_close(flat, want, REL_TOL, "gradients vs float64")
        infer = st.forward(x, train=False)[0].clone()
        assert_close(infer.cpu().numpy(), y64.detach().numpy(), REL_TOL, "inference form")
        assert lib.mri_set_option(b"siren_rows", 0) == 0
        pred0, loss0, grad0, st0 = _step(amd, net, x, y)
        assert_close(pred.cpu().numpy(), pred0.cpu().numpy(), 2e-6, "prediction vs image kernels")
        assert abs(loss - loss0) <= 2e-6 * max(abs(loss0), 0.1)
        assert_close(grad.cpu().numpy(), grad0.cpu().numpy(), 2e-6, "gradients vs image kernels")
    finally:
        assert lib.mri_set_option(b"siren_rows", 1) == 0


def test_rows_kernels_are_bitwise_reproducible(amd):
    """No float atomics, fixed summation orders (lane butterflies, slabs): the same bits on every run, ragged n included."""
    net, _ = _net(amd, 3, 4, 17)
    x = torch.rand(33333, 3, device="cuda") * 2 - 1
    y = torch.rand(33333, 1, device="cuda")
    runs = []
    for _ in range(3):
        pred, loss, grad, _ = _step(amd, net, x, y)
        runs.append((pred, loss, grad))
        torch.empty(1 << 24, device="cuda").normal_()
    assert all(torch.equal(runs[0][0], r[0]) and runs[0][1] == r[1] and torch.equal(runs[0][2], r[2]) for r in runs[1:])


@pytest.mark.parametrize("where", ["first", "hidden"])
def test_rows_sincos_cold_path(amd, where):
    """The rows kernels evaluate sin / cos branch-free between the MFMAs (Cody-Waite, good to |u| <= 8192) and repair a
    tile whose arguments go beyond on a cold path (library sincosf through LDS).  Weights scaled so that some -- not
    all -- arguments of the first / second sine layer exceed 8192: prediction and gradients against float64 evaluated AT
    THE KERNEL'S f32 ARGUMENTS' precision is meaningless out there (sin of 1e4 amplifies the argument's rounding by 1e4),
    so the yardstick is the image kernels' arithmetic (sincos_fast: same reduction, same library fallback, element
    by element): equal to 2e-6, and finite."""
    lib = amd.lib.load()
    n = 2000
    x = torch.from_numpy(detrand.uniform(n * 3, 71, -1.0, 1.0).reshape(n, 3)).cuda()
    y = torch.from_numpy(detrand.uniform(n, 72, -1.0, 1.0).reshape(n, 1)).cuda()

    def arguments(net):
        with torch.no_grad():
            layers = list(net.layers)
            u = net.w0_initial * torch.nn.functional.linear(x, layers[0].weight, layers[0].bias)
            if where == "hidden":
                u = net.w0 * torch.nn.functional.linear(torch.sin(u), layers[1].weight, layers[1].bias)
            return u.abs()

    # the scale that puts the 70th percentile of the layer's |argument| at 8192: ~30 % of them beyond, in every tile
    scale = 8192.0 / float(torch.quantile(arguments(_net(amd, 3, 3, 55)[0]).flatten()[:1 << 20], 0.7))
    net, _ = _net(amd, 3, 3, 55, *((scale, 1.0) if where == "first" else (1.0, scale)))
    big = (arguments(net) > 8192).float().mean().item()
    assert 0.05 < big < 0.6, big
    try:
        pred, loss, grad, st = _step(amd, net, x, y)
        assert torch.isfinite(pred).all() and torch.isfinite(grad).all()
        infer = st.forward(x, train=False)[0].clone()
        assert lib.mri_set_option(b"siren_rows", 0) == 0
        pred0, loss0, grad0, st0 = _step(amd, net, x, y)
        infer0 = st0.forward(x, train=False)[0].clone()
        assert_close(pred.cpu().numpy(), pred0.cpu().numpy(), 2e-6, "prediction vs image kernels")
        assert_close(infer.cpu().numpy(), infer0.cpu().numpy(), 2e-6, "inference vs image kernels")
        assert_close(grad.cpu().numpy(), grad0.cpu().numpy(), 2e-6, "gradients vs image kernels")
    finally:
        assert lib.mri_set_option(b"siren_rows", 1) == 0


def test_head_done_checks_what_it_continues(amd, monkeypatch):
    """mri_siren_backward(head_done = 1) continues the mri_siren_forward_loss call before it (at width 256 the forward
    call parks w0 cos in dz_last and dLoss/dy in the workspace): a backward call that does not match -- another kernel
    family selected in between, another batch size -- fails loudly instead of reading the wrong thing."""
    from mri_interpolation_amd import ops
    lib = amd.lib.load()
    net, _ = _net(amd, 3, 3, 5)
    x = torch.rand(500, 3, device="cuda") * 2 - 1
    y = torch.rand(500, 1, device="cuda")
    st = amd.trainer.FusedStep(net, net.configure_optimizers())
    real_backward = ops.siren_backward

    def flipped(*args, **kw):
        assert lib.mri_set_option(b"siren_rows", 0) == 0
        try:
            return real_backward(*args, **kw)
        finally:
            assert lib.mri_set_option(b"siren_rows", 1) == 0

    monkeypatch.setattr(ops, "siren_backward", flipped)
    with pytest.raises(RuntimeError, match="continues the mri_siren_forward_loss call"):
        st._chain_loss_pass(x, y, True, 1.0)
    monkeypatch.setattr(ops, "siren_backward", real_backward)
    st._chain_loss_pass(x, y, True, 1.0)  # the matching pair still works
    assert torch.isfinite(st.flat.grad).all()
