"""Tensor-level wrappers over the C ABI (include/mri_inr.h) and the autograd glue.

PyTorch is plumbing here: it owns device memory and the stream; every computation below
is a call into libmri_inr.so.  CPU tensors are rejected -- there is no fallback.
"""
import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_IDENTITY, ACT_RELU, ACT_SINE, DERIV_MUL, DERIV_NONE,  # noqa: F401
                   DERIV_RELU_MASK, GridDesc)


# --------------------------------------------------------------------------- helpers
def _gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mri_interpolation_amd: the hot path runs on MI355X only; got a "
                               f"{t.device} tensor (no CPU fallback exists)")
        if t.dtype != torch.float32 and t.dtype != torch.int64:
            raise TypeError(f"expected float32/int64 tensor, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    # the raw handle of the current stream: torch.cuda.current_stream() builds a Stream object and resolves
    # the device three times over (8 us per call, a dozen calls per training step: tools/host_profile.py)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """2-D view with unit column stride (copies only if needed)."""
    if t.dim() != 2:
        t = t.reshape(-1, t.shape[-1])
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def make_grid_desc(dim: int, resolutions: Sequence[Sequence[float]], sizes: Sequence[int],
                   n_features: int) -> GridDesc:
    """Host descriptor from per-level per-axis resolutions and table sizes."""
    if not (1 <= dim <= _lib.MAX_DIM):
        raise ValueError(f"HashGrid only supports 1..{_lib.MAX_DIM}-D inputs")
    if not (1 <= len(sizes) <= _lib.MAX_LEVELS):
        raise ValueError(f"n_levels must be in 1..{_lib.MAX_LEVELS}")
    g = GridDesc()
    g.dim, g.n_levels, g.n_features = dim, len(sizes), n_features
    off = 0
    for l, (res, size) in enumerate(zip(resolutions, sizes)):
        for d in range(dim):
            g.resolution[l][d] = float(res[d])
        g.table_size[l] = int(size)
        g.table_offset[l] = off
        off += int(size)
    return g


def grid_rows(desc: GridDesc) -> int:
    return sum(desc.table_size[l] for l in range(desc.n_levels))


def _enc_strides(desc: GridDesc, n: int, feature_major: bool):
    f = desc.n_features
    if feature_major:  # (L*F, n)
        return f * n, 1, n
    return f, desc.n_levels * f, 1  # reference layout (n, L*F)


# --------------------------------------------------------------------------- hash grid
def hashgrid_forward(desc: GridDesc, x: torch.Tensor, table: torch.Tensor,
                     out: Optional[torch.Tensor] = None, feature_major: bool = False,
                     row_offset: int = 0):
    """Encode the rows of `x`.  With `row_offset`, `x` is a slice of a larger batch whose features
    live in `out` (the whole batch's buffer): the slice's features go to rows / columns
    [row_offset, row_offset + len(x))."""
    _gpu(x, table, out)
    x = _rowmajor(x).contiguous()
    n, width = x.shape[0], desc.n_levels * desc.n_features
    if x.shape[1] != desc.dim:
        raise ValueError(f"x has {x.shape[1]} columns, encoder expects {desc.dim}")
    if out is None:
        out = torch.empty((width, n) if feature_major else (n, width), device=x.device,
                          dtype=torch.float32)
    n_total = out.shape[1] if feature_major else out.shape[0]
    if row_offset < 0 or row_offset + n > n_total:
        raise ValueError("slice does not fit the output buffer")
    sl, sr, sf = _enc_strides(desc, n_total, feature_major)
    base = out.data_ptr() + 4 * row_offset * sr
    _lib.call("mri_hashgrid_forward", C.byref(desc), _ptr(x), n, _ptr(table), C.c_void_p(base), sl,
              sr, sf, _stream())
    return out


_bwd_workspace = {}  # device index -> scratch (the library clears what it needs per call)


def backward_workspace_bytes(desc: GridDesc, n: int) -> int:
    need = _lib.load().mri_hashgrid_backward_workspace_bytes(C.byref(desc), n)
    if need < 0:
        _lib.check(-1, "mri_hashgrid_backward_workspace_bytes")
    return int(need)


def backward_workspace(desc: GridDesc, n: int, device) -> torch.Tensor:
    """Process-wide scratch of the module / autograd path (FusedStep owns its own buffer).  It
    only ever grows; before a larger one replaces it the device is synchronised, so no kernel of
    any stream can still be using the buffer that goes back to the allocator."""
    need = backward_workspace_bytes(desc, n)
    ws = _bwd_workspace.get(device.index)
    if ws is None or ws.numel() * 8 < need:
        if ws is not None:
            torch.cuda.synchronize(device)
        ws = torch.empty((need + 7) // 8, dtype=torch.int64, device=device)
        _bwd_workspace[device.index] = ws
    return ws


def hashgrid_backward_prepare(desc: GridDesc, x: torch.Tensor, method: int = 0, stream=None,
                              ws: Optional[torch.Tensor] = None):
    """First stage of the binned backward (record counts per table slice): needs only `x`, so
    it can be queued on a side stream while the forward pass runs.  Follow with
    hashgrid_backward(..., prepared=True) on the same workspace `ws` (int64 tensor of at least
    backward_workspace_bytes(); default: the process-wide one)."""
    _gpu(x)
    n = x.shape[0]
    if ws is None:
        ws = backward_workspace(desc, n, x.device)
    st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    _lib.call("mri_hashgrid_backward_prepare", C.byref(desc), _ptr(x), n, method, _ptr(ws),
              ws.numel() * 8, st)


def hashgrid_backward(desc: GridDesc, x: torch.Tensor, d_out: torch.Tensor,
                      d_table: torch.Tensor, feature_major: bool = False, method: int = 0,
                      prepared: bool = False, overwrite: bool = False,
                      level_mask: Optional[int] = None, ws: Optional[torch.Tensor] = None,
                      level_absmax: Optional[torch.Tensor] = None):
    """d_table += scatter of d_out (or d_table = ..., with overwrite=True); `level_mask`
    restricts the call to the levels whose bit is set (one level group of a bucketed,
    data-parallel backward); `level_absmax` (n_levels floats on the device): max |d_out| per level
    when the caller has it already (tiny_mlp_train(..., dx_absmax=)), saving a pass over d_out."""
    _gpu(x, d_out, d_table)
    x = _rowmajor(x).contiguous()
    n = x.shape[0]
    if not d_out.is_contiguous():
        d_out = d_out.contiguous()
    sl, sr, sf = _enc_strides(desc, n, feature_major)
    if method == 1:
        ws = None
    elif ws is None:
        ws = backward_workspace(desc, n, x.device)
    flags = (method | (_lib.BWD_PREPARED if prepared else 0)
             | (_lib.BWD_OVERWRITE if overwrite else 0))
    if level_absmax is not None:
        _gpu(level_absmax)
        if level_absmax.numel() < desc.n_levels or level_absmax.dtype != torch.float32:
            raise ValueError("level_absmax: one float32 per level")
        _lib.call("mri_hashgrid_backward_scaled", C.byref(desc), _ptr(x), _ptr(d_out), n, sl, sr,
                  sf, _ptr(d_table), flags, (0xFFFFFFFF if level_mask is None else level_mask) & 0xFFFFFFFF,
                  _ptr(level_absmax), _ptr(ws), ws.numel() * 8 if ws is not None else 0, _stream())
    elif level_mask is None:
        _lib.call("mri_hashgrid_backward", C.byref(desc), _ptr(x), _ptr(d_out), n, sl, sr, sf,
                  _ptr(d_table), flags, _ptr(ws), ws.numel() * 8 if ws is not None else 0,
                  _stream())
    else:
        _lib.call("mri_hashgrid_backward_levels", C.byref(desc), _ptr(x), _ptr(d_out), n, sl, sr,
                  sf, _ptr(d_table), flags, level_mask & 0xFFFFFFFF, _ptr(ws),
                  ws.numel() * 8 if ws is not None else 0, _stream())
    return d_table


def hashgrid_backward_adam(desc: GridDesc, x: torch.Tensor, d_out: torch.Tensor, table: torch.Tensor,
                           exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, lr: float, beta1: float,
                           beta2: float, eps: float, step: int, grad_scale: float = 1.0,
                           feature_major: bool = False, method: int = 0, prepared: bool = False,
                           ws: Optional[torch.Tensor] = None) -> bool:
    """Table gradient and the table's Adam step in one pass (mri_hashgrid_backward_adam): `table`,
    `exp_avg`, `exp_avg_sq` are updated in place, no gradient tensor is produced.  Returns False
    (nothing done) when a level of the grid is not served by the binned kernels."""
    _gpu(x, d_out, table, exp_avg, exp_avg_sq)
    x = _rowmajor(x).contiguous()
    n = x.shape[0]
    if not d_out.is_contiguous():
        d_out = d_out.contiguous()
    if not (table.is_contiguous() and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous()):
        raise ValueError("hashgrid_backward_adam needs contiguous table / moment buffers")
    sl, sr, sf = _enc_strides(desc, n, feature_major)
    if method == 1:
        return False
    if ws is None:
        ws = backward_workspace(desc, n, x.device)
    flags = method | (_lib.BWD_PREPARED if prepared else 0)
    rc = _lib.load().mri_hashgrid_backward_adam(
        C.byref(desc), _ptr(x), _ptr(d_out), n, sl, sr, sf, _ptr(table), _ptr(exp_avg), _ptr(exp_avg_sq),
        float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale), flags, _ptr(ws),
        ws.numel() * 8, _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    _lib.check(rc, "mri_hashgrid_backward_adam")
    return True


def hashgrid_backward_input(desc: GridDesc, x: torch.Tensor, d_out: torch.Tensor,
                            table: torch.Tensor, feature_major: bool = False) -> torch.Tensor:
    """d loss / d x (n, D): the reference keeps x -> x*res - trunc(x*res) differentiable
    (encoding.py:111-113), so callers that track coordinates (e.g. spatial derivatives of the
    implicit image) get their gradient; the training path never asks for it."""
    _gpu(x, d_out, table)
    x = _rowmajor(x).contiguous()
    n = x.shape[0]
    if not d_out.is_contiguous():
        d_out = d_out.contiguous()
    sl, sr, sf = _enc_strides(desc, n, feature_major)
    dx = torch.empty_like(x)
    _lib.call("mri_hashgrid_backward_input", C.byref(desc), _ptr(x), _ptr(d_out), n, sl, sr, sf,
              _ptr(table), _ptr(dx), _stream())
    return dx


class HashGridFunction(torch.autograd.Function):
    """Autograd node for the encoder: gradient to the table (the training path) and, when the
    caller tracks the coordinates, to x (the reference keeps that path, encoding.py:113)."""

    @staticmethod
    def forward(ctx, x, table, desc):
        ctx.desc = desc
        ctx.save_for_backward(x, table)
        return hashgrid_forward(desc, x, table)

    @staticmethod
    def backward(ctx, d_out):
        x, table = ctx.saved_tensors
        d_out = d_out.contiguous()
        d_x = d_table = None
        if ctx.needs_input_grad[1]:
            d_table = torch.zeros(table.shape, device=d_out.device, dtype=torch.float32)
            hashgrid_backward(ctx.desc, x, d_out, d_table)
        if ctx.needs_input_grad[0]:
            d_x = hashgrid_backward_input(ctx.desc, x, d_out, table).reshape(x.shape)
        return d_x, d_table, None


# --------------------------------------------------------------------------- linear layers
def linear_forward(x, weight, bias, activation=ACT_IDENTITY, w0=1.0, out=None, deriv=None,
                   x_feature_major=False):
    """y = act(w0 * (x W^T + b)); x is (M, K), or (K, M) when x_feature_major."""
    _gpu(x, weight, bias, out, deriv)
    n, k = weight.shape
    if x_feature_major:
        if not x.is_contiguous():
            x = x.contiguous()
        m, xrs, xcs = x.shape[1], 1, x.shape[1]
        if x.shape[0] != k:
            raise ValueError(f"x has {x.shape[0]} features, weight expects {k}")
    else:
        x = _rowmajor(x)
        m, xrs, xcs = x.shape[0], x.stride(0), 1
        if x.shape[1] != k:
            raise ValueError(f"x has {x.shape[1]} features, weight expects {k}")
    if not weight.is_contiguous():
        weight = weight.contiguous()
    if out is None:
        out = torch.empty((m, n), device=x.device, dtype=torch.float32)
    _lib.call("mri_linear_forward", _ptr(x), xrs, xcs, _ptr(weight), _ptr(bias), m, n, k,
              activation, float(w0), _ptr(out), out.stride(0), _ptr(deriv),
              deriv.stride(0) if deriv is not None else 0, _stream())
    return out


def linear_backward_data(dy, weight, deriv_mode=DERIV_NONE, deriv=None, dx=None,
                         dx_feature_major=False):
    _gpu(dy, weight, deriv, dx)
    dy = _rowmajor(dy)
    n, k = weight.shape
    m = dy.shape[0]
    if not weight.is_contiguous():
        weight = weight.contiguous()
    if dx is None:
        dx = torch.empty((k, m) if dx_feature_major else (m, k), device=dy.device,
                         dtype=torch.float32)
    drs, dcs = (1, dx.stride(0)) if dx_feature_major else (dx.stride(0), 1)
    _lib.call("mri_linear_backward_data", _ptr(dy), dy.stride(0), _ptr(weight), m, n, k,
              deriv_mode, _ptr(deriv), deriv.stride(0) if deriv is not None else 0, _ptr(dx),
              drs, dcs, _stream())
    return dx


def linear_backward_weight(dy, x, d_weight, d_bias=None, x_feature_major=False):
    """d_weight += dy^T x, d_bias += colsum(dy)."""
    _gpu(dy, x, d_weight, d_bias)
    dy = _rowmajor(dy)
    n, k = d_weight.shape
    m = dy.shape[0]
    if x_feature_major:
        if not x.is_contiguous():
            x = x.contiguous()
        xrs, xcs = 1, x.shape[1]
    else:
        x = _rowmajor(x)
        xrs, xcs = x.stride(0), 1
    _lib.call("mri_linear_backward_weight", _ptr(dy), dy.stride(0), _ptr(x), xrs, xcs, m, n, k,
              _ptr(d_weight), _ptr(d_bias), _stream())


def apply_deriv(dy, deriv_mode, deriv):
    _gpu(dy, deriv)
    if deriv_mode == DERIV_NONE:
        return dy
    _lib.call("mri_apply_deriv", _ptr(dy), dy.stride(0), deriv_mode, _ptr(deriv),
              deriv.stride(0), dy.shape[0], dy.shape[1], _stream())
    return dy


def deriv_mode_for(activation: int) -> int:
    return {ACT_IDENTITY: DERIV_NONE, ACT_RELU: DERIV_RELU_MASK, ACT_SINE: DERIV_MUL,
            ACT_GELU: DERIV_MUL}[activation]


class LinearActFunction(torch.autograd.Function):
    """y = act(w0 (x W^T + b)) as one MFMA kernel, with its three backward GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, activation, w0):
        lead = x.shape[:-1]
        x2 = _rowmajor(x)
        need_deriv = activation in (ACT_SINE, ACT_GELU)
        deriv = torch.empty((x2.shape[0], weight.shape[0]), device=x.device,
                            dtype=torch.float32) if need_deriv else None
        y = linear_forward(x2, weight, bias, activation, w0, deriv=deriv)
        ctx.activation = activation
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x2, weight, deriv if need_deriv else y)
        ctx.lead = lead
        return y.reshape(*lead, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, g = ctx.saved_tensors
        dz = dy.reshape(-1, weight.shape[0]).contiguous()
        mode = deriv_mode_for(ctx.activation)
        if mode != DERIV_NONE:
            dz = apply_deriv(dz.clone() if dz.data_ptr() == dy.data_ptr() else dz, mode, g)
        dx = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            dx = linear_backward_data(dz, weight).reshape(*ctx.lead, weight.shape[1])
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            d_w = torch.zeros_like(weight)
            d_b = torch.zeros(weight.shape[0], device=dy.device) if ctx.has_bias else None
            linear_backward_weight(dz, x2, d_w, d_b)
        return dx, d_w, d_b, None, None


def linear_act(x, weight, bias, activation=ACT_IDENTITY, w0=1.0):
    return LinearActFunction.apply(x, weight, bias, activation, w0)


class ModulateFunction(torch.autograd.Function):
    """y = x (.) m, the elementwise modulation of reference models.py:318-320 (`x *= mod`)."""

    @staticmethod
    def forward(ctx, x, m):
        x2, m2 = _rowmajor(x), _rowmajor(m)
        ctx.save_for_backward(x2, m2)
        return apply_deriv(x2.clone(), DERIV_MUL, m2)

    @staticmethod
    def backward(ctx, dy):
        x2, m2 = ctx.saved_tensors
        dy = _rowmajor(dy)
        dx = apply_deriv(dy.clone(), DERIV_MUL, m2) if ctx.needs_input_grad[0] else None
        dm = apply_deriv(dy.clone(), DERIV_MUL, x2) if ctx.needs_input_grad[1] else None
        return dx, dm


def modulate(x, m):
    return ModulateFunction.apply(x, m)


# --------------------------------------------------------------------------- frequency encoding
def frequency_forward(x, n_levels: int, out=None):
    _gpu(x, out)
    x = _rowmajor(x)
    n, dim = x.shape
    if out is None:
        out = torch.empty((n, dim * 2 * n_levels), device=x.device, dtype=torch.float32)
    _lib.call("mri_frequency_forward", _ptr(x), x.stride(0), n, dim, n_levels, _ptr(out),
              out.stride(0), _stream())
    return out


def frequency_backward(x, d_out, n_levels: int):
    _gpu(x, d_out)
    x, d_out = _rowmajor(x), _rowmajor(d_out)
    n, dim = x.shape
    dx = torch.empty((n, dim), device=x.device, dtype=torch.float32)
    _lib.call("mri_frequency_backward", _ptr(x), x.stride(0), _ptr(d_out), d_out.stride(0), n,
              dim, n_levels, _ptr(dx), dx.stride(0), _stream())
    return dx


class FrequencyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_levels):
        lead = x.shape[:-1]
        x2 = _rowmajor(x.reshape(-1, x.shape[-1]))
        ctx.save_for_backward(x2)
        ctx.n_levels, ctx.shape = n_levels, x.shape
        return frequency_forward(x2, n_levels).reshape(*lead, x.shape[-1] * 2 * n_levels)

    @staticmethod
    def backward(ctx, d_out):
        (x2,) = ctx.saved_tensors
        dx = frequency_backward(x2, d_out.reshape(x2.shape[0], -1), ctx.n_levels)
        return dx.reshape(ctx.shape), None


# --------------------------------------------------------------------------- fused tiny MLP
def tiny_mlp_supported(k_in: int, hidden: int, dim_out: int) -> bool:
    return bool(_lib.load().mri_tiny_mlp_supported(k_in, hidden, dim_out))


_mlp_workspace = {}


def _tiny_workspace(k_in, hidden, n, device):
    need = _lib.load().mri_tiny_mlp_workspace_bytes(k_in, hidden, n)
    ws = _mlp_workspace.get(device.index)
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:  # may still be in use on another stream: see backward_workspace()
            torch.cuda.synchronize(device)
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
        _mlp_workspace[device.index] = ws
    return ws


def tiny_mlp_forward(x_fm, params, y=None):
    """y (n, 1) = MLP(x) for feature-major x (k_in, n); params = [(w1,b1),(w2,b2),(w3,b3)]."""
    (w1, b1), (w2, b2), (w3, b3) = params
    _gpu(x_fm, w1, b1, w2, b2, w3, b3, y)
    k_in, n = x_fm.shape
    if y is None:
        y = torch.empty((n, 1), device=x_fm.device, dtype=torch.float32)
    _lib.call("mri_tiny_mlp_forward", _ptr(x_fm), n, k_in, w1.shape[0], _ptr(w1), _ptr(b1),
              _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(y), _stream())
    return y


def tiny_mlp_dx_absmax_supported(k_in: int, hidden: int) -> bool:
    """Can tiny_mlp_train report max |d_x| per pair of feature rows (dx_absmax=)?"""
    return bool(_lib.load().mri_tiny_mlp_dx_absmax_supported(int(k_in), int(hidden)))


def tiny_mlp_train(x_fm, target, params, grads, loss_out, d_x=None, y=None,
                   grad_divisor: float = 1.0, overwrite: bool = False, dx_absmax=None):
    """Forward + MSE + backward of the tiny MLP in one kernel; grads accumulate (or are
    overwritten, together with loss_out, when overwrite=True).  `dx_absmax` (k_in / 2 zeroed floats
    on the device) receives max |d_x| per pair of feature rows (hashgrid_backward's level_absmax)."""
    (w1, b1), (w2, b2), (w3, b3) = params
    (g1, gb1), (g2, gb2), (g3, gb3) = grads
    _gpu(x_fm, target, w1, b1, w2, b2, w3, b3, g1, gb1, g2, gb2, g3, gb3, loss_out, d_x, y)
    k_in, n = x_fm.shape
    ws = _tiny_workspace(k_in, w1.shape[0], n, x_fm.device)
    if dx_absmax is not None:
        _gpu(dx_absmax)
        if dx_absmax.numel() < (k_in + 1) // 2 or dx_absmax.dtype != torch.float32:
            raise ValueError("dx_absmax: one float32 per pair of feature rows")
        _lib.call("mri_tiny_mlp_train_dx_absmax", _ptr(x_fm), _ptr(target), n, k_in, w1.shape[0],
                  _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), float(grad_divisor),
                  _ptr(g1), _ptr(gb1), _ptr(g2), _ptr(gb2), _ptr(g3), _ptr(gb3), _ptr(d_x),
                  _ptr(loss_out), _ptr(y), 1 if overwrite else 0, _ptr(dx_absmax), _ptr(ws),
                  ws.numel() * 4, _stream())
        return loss_out
    _lib.call("mri_tiny_mlp_train_overwrite" if overwrite else "mri_tiny_mlp_train", _ptr(x_fm),
              _ptr(target), n, k_in, w1.shape[0], _ptr(w1),
              _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), float(grad_divisor), _ptr(g1),
              _ptr(gb1), _ptr(g2), _ptr(gb2), _ptr(g3), _ptr(gb3), _ptr(d_x), _ptr(loss_out),
              _ptr(y), _ptr(ws), ws.numel() * 4, _stream())
    return loss_out


def hash_tiny_mlp_supported(desc: GridDesc, hidden: int) -> bool:
    """Is there a one-kernel form of encoder + decoder step for this grid and decoder width?"""
    return bool(_lib.load().mri_hash_tiny_mlp_supported(C.byref(desc), int(hidden)))


def hash_tiny_mlp_train(desc: GridDesc, table, coords, target, params, grads, loss_out, d_enc,
                        y=None, grad_divisor: float = 1.0, overwrite: bool = False):
    """hashgrid_forward + tiny_mlp_train in one kernel (mri_hash_tiny_mlp_train): the decoder looks the
    features up itself; `d_enc` (2 L, ld >= n) feature-major receives dLoss / d features."""
    (w1, b1), (w2, b2), (w3, b3) = params
    (g1, gb1), (g2, gb2), (g3, gb3) = grads
    _gpu(table, coords, target, w1, b1, w2, b2, w3, b3, g1, gb1, g2, gb2, g3, gb3, loss_out, d_enc, y)
    coords = _rowmajor(coords).contiguous()
    n = coords.shape[0]
    k_in = 2 * desc.n_levels
    if d_enc is not None and (d_enc.shape[0] != k_in or d_enc.stride(1) != 1 or d_enc.shape[1] < n):
        raise ValueError("d_enc must be a feature-major (2 L, >= n) block")
    ws = _tiny_workspace(k_in, w1.shape[0], n, coords.device)
    _lib.call("mri_hash_tiny_mlp_train", C.byref(desc), _ptr(table), _ptr(coords), _ptr(target), n,
              w1.shape[0], _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3),
              float(grad_divisor), _ptr(g1), _ptr(gb1), _ptr(g2), _ptr(gb2), _ptr(g3), _ptr(gb3),
              _ptr(d_enc), d_enc.stride(0) if d_enc is not None else n, _ptr(loss_out), _ptr(y),
              1 if overwrite else 0, _ptr(ws), ws.numel() * 4, _stream())
    return loss_out


def tiny_mlp_train_slice(x_fm, target, col_offset: int, n: int, params, grads, loss_out, d_x,
                         grad_divisor: float = 1.0, overwrite: bool = False):
    """tiny_mlp_train for columns [col_offset, col_offset + n) of the feature-major batch block
    x_fm (k_in, n_total) and of d_x; the loss mean and gradient scale are those of the whole
    batch, so the slices of a batch add up to the whole-batch call."""
    (w1, b1), (w2, b2), (w3, b3) = params
    (g1, gb1), (g2, gb2), (g3, gb3) = grads
    _gpu(x_fm, target, w1, b1, w2, b2, w3, b3, g1, gb1, g2, gb2, g3, gb3, loss_out, d_x)
    k_in, n_total = x_fm.shape
    if col_offset < 0 or col_offset + n > n_total or target.numel() != n_total:
        raise ValueError("slice does not fit the batch")
    ws = _tiny_workspace(k_in, w1.shape[0], n, x_fm.device)
    at = lambda t: C.c_void_p(t.data_ptr() + 4 * col_offset)  # noqa: E731
    _lib.call("mri_tiny_mlp_train_slice", at(x_fm), n_total, at(target), n, n_total, k_in,
              w1.shape[0], _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3),
              float(grad_divisor), _ptr(g1), _ptr(gb1), _ptr(g2), _ptr(gb2), _ptr(g3), _ptr(gb3),
              at(d_x) if d_x is not None else None, _ptr(loss_out), None, 1 if overwrite else 0,
              _ptr(ws), ws.numel() * 4, _stream())
    return loss_out


# ----------------------------------------------------------- lookup and decoder side by side
def tiny_mlp_round_rows(k_in: int, hidden: int, n: int) -> int:
    """Rows one round of the decoder kernel's workgroups reads (0: no overlapped form)."""
    return int(_lib.load().mri_tiny_mlp_round_rows(k_in, hidden, n))


def hashgrid_signal_blocks(desc: GridDesc, slice_rows: int) -> int:
    """What mri_hashgrid_forward_signal adds to a slice's counter (-1: unsupported grid)."""
    return int(_lib.load().mri_hashgrid_forward_signal_blocks(C.byref(desc), slice_rows))


def hashgrid_forward_signal(desc: GridDesc, x, table, out_fm, slice_rows: int, ready):
    """Feature-major lookup on the CURRENT stream that reports finished slices in `ready`
    (uint64 counters, one per slice of `slice_rows` coordinates, never reset by the library)."""
    _gpu(x, table, out_fm)
    x = _rowmajor(x).contiguous()
    n = x.shape[0]
    if out_fm.shape[0] != desc.n_levels * desc.n_features or out_fm.shape[1] < n \
            or not out_fm.is_contiguous():
        raise ValueError("out_fm must be a contiguous (L*F, >= n) block")
    if ready.dtype != torch.int64 or ready.numel() * slice_rows < n:
        raise ValueError("ready: one int64 counter per slice")
    _lib.call("mri_hashgrid_forward_signal", C.byref(desc), _ptr(x), n, _ptr(table), _ptr(out_fm),
              out_fm.shape[1], slice_rows, C.c_void_p(ready.data_ptr()), _stream())
    return out_fm


def tiny_mlp_train_overlapped(x_fm, target, params, grads, loss_out, d_x, ready, ready_target: int,
                              status, grad_divisor: float = 1.0, overwrite: bool = True):
    """tiny_mlp_train whose workgroups wait for the producer of x_fm (hashgrid_forward_signal,
    queued earlier on ANOTHER stream) slice by slice."""
    (w1, b1), (w2, b2), (w3, b3) = params
    (g1, gb1), (g2, gb2), (g3, gb3) = grads
    _gpu(x_fm, target, w1, b1, w2, b2, w3, b3, g1, gb1, g2, gb2, g3, gb3, loss_out, d_x)
    k_in, n = x_fm.shape
    ws = _tiny_workspace(k_in, w1.shape[0], n, x_fm.device)
    _lib.call("mri_tiny_mlp_train_overlapped", _ptr(x_fm), _ptr(target), n, k_in, w1.shape[0],
              _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), float(grad_divisor),
              _ptr(g1), _ptr(gb1), _ptr(g2), _ptr(gb2), _ptr(g3), _ptr(gb3), _ptr(d_x),
              _ptr(loss_out), 1 if overwrite else 0, C.c_void_p(ready.data_ptr()),
              C.c_uint64(ready_target), C.c_void_p(status.data_ptr()), _ptr(ws), ws.numel() * 4,
              _stream())
    return loss_out


# --------------------------------------------------------------------------- fused SIREN chain
def siren_supported(dim_in: int, hidden: int, n_sine_layers: int, dim_out: int) -> bool:
    return bool(_lib.load().mri_siren_supported(dim_in, hidden, n_sine_layers, dim_out))


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def siren_forward(x, weights, biases, w0_first: float, w0: float, act=None, deriv=None, y=None):
    """y (n, 1) = SirenNet(x) in one persistent kernel (csrc/siren_chain.hip).  weights / biases:
    the sine layers' then the head's; act / deriv: per sine layer (n, hidden) buffers that receive
    sin(.) and w0 cos(.) for the backward pass (both None: inference, nothing but y is written)."""
    _gpu(x, y, *weights, *biases, *(act or []), *(deriv or []))
    x = _rowmajor(x).contiguous()
    n, dim_in = x.shape
    n_sine, hidden = len(weights) - 1, weights[0].shape[0]
    if y is None:
        y = torch.empty((n, 1), device=x.device, dtype=torch.float32)
    for t in list(weights) + list(biases) + list(act or []) + list(deriv or []):
        if not t.is_contiguous():
            raise ValueError("siren_forward needs contiguous parameters and buffers")
    if (act is None) != (deriv is None) or (act is not None and
                                            (len(act) != n_sine or len(deriv) != n_sine)):
        raise ValueError("act and deriv: one (n, hidden) buffer per sine layer, or both None")
    ws = _siren_scratch(x.device, max(n, 1), hidden, n_sine)  # holds the split weights
    _lib.call("mri_siren_forward", _ptr(x), n, dim_in, hidden, n_sine, _ptr_array(weights),
              _ptr_array(biases), float(w0_first), float(w0),
              _ptr_array(act) if act is not None else None,
              _ptr_array(deriv) if deriv is not None else None, _ptr(y), _ptr(ws), ws.numel() * 4,
              _stream())
    return y


_siren_workspace = {}


def _siren_scratch(device, n, hidden, n_sine):
    need = _lib.load().mri_siren_backward_workspace_bytes(n, hidden, n_sine)
    ws = _siren_workspace.get(device.index)
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:
            torch.cuda.synchronize(device)
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
        _siren_workspace[device.index] = ws
    return ws


def _opt_ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def siren_forward_loss(x, target, weights, biases, w0_first: float, w0: float, act, deriv, dz_last,
                       y, d_w_head, d_b_head, d_b_last, loss_out, grad_divisor: float = 1.0,
                       n_total=None):
    """Training forward of the fused SIREN chain with F.mse_loss and the head's backward in the
    same kernel (mri_siren_forward_loss, include/mri_inr.h).  act / deriv: per sine layer
    (n, hidden) buffers, the LAST layer's may be None (it stays on chip); dz_last (n, hidden)
    receives the gradient w.r.t. the last sine layer's pre-activation; d_w_head, d_b_head,
    d_b_last and loss_out are ADDED to.  Follow with siren_backward(..., head_done=True)."""
    _gpu(x, target, y, dz_last, d_w_head, d_b_head, d_b_last, loss_out, *weights, *biases,
         *[t for t in list(act) + list(deriv) if t is not None])
    x = _rowmajor(x).contiguous()
    n, dim_in = x.shape
    n_sine, hidden = len(weights) - 1, weights[0].shape[0]
    if len(act) != n_sine or len(deriv) != n_sine:
        raise ValueError("act and deriv: one entry per sine layer (the last may be None)")
    if target.numel() != n or not target.is_contiguous() or y.numel() != n or \
            dz_last.shape != (n, hidden):
        raise ValueError("siren_forward_loss: target / y (n, 1) contiguous, dz_last (n, hidden)")
    for t in list(weights) + list(biases) + [t for t in list(act) + list(deriv) if t is not None] + \
            [dz_last, y, d_w_head, d_b_head, d_b_last]:
        if not t.is_contiguous():
            raise ValueError("siren_forward_loss needs contiguous parameters and buffers")
    ws = _siren_scratch(x.device, n, hidden, n_sine)
    _lib.call("mri_siren_forward_loss", _ptr(x), _ptr(target), n, n if n_total is None else n_total,
              dim_in, hidden, n_sine, _ptr_array(weights), _ptr_array(biases), float(w0_first),
              float(w0), float(grad_divisor), _opt_ptr_array(act), _opt_ptr_array(deriv),
              _ptr(dz_last), _ptr(y), _ptr(d_w_head), _ptr(d_b_head), _ptr(d_b_last), _ptr(loss_out),
              _ptr(ws), ws.numel() * 4, _stream())
    return y


def siren_backward(x, dy, weights, act, deriv, dz, d_weights, d_biases, head_done: bool = False):
    """Gradients of the fused SIREN chain, ADDED to d_weights / d_biases (sine layers, then the
    head).  dy: (n, 1) loss gradient w.r.t. the prediction; act / deriv: what siren_forward
    stored; dz: per sine layer (n, hidden) scratch (dz[0] may be None).  head_done: the head's
    backward already ran in siren_forward_loss, which left its result in dz[-1] (dy and the last
    act / deriv are not read, and may be None)."""
    _gpu(x, dy, *weights, *[t for t in list(act) + list(deriv) + list(dz) if t is not None],
         *d_weights, *d_biases)
    x = _rowmajor(x).contiguous()
    n, dim_in = x.shape
    n_sine, hidden = len(weights) - 1, weights[0].shape[0]
    if not (len(act) == len(deriv) == len(dz) == n_sine and
            len(d_weights) == len(d_biases) == n_sine + 1):
        raise ValueError("siren_backward: one act / deriv / dz per sine layer, one gradient per layer")
    ws = _siren_scratch(x.device, n, hidden, n_sine)
    _lib.call("mri_siren_backward", _ptr(x), _ptr(dy), n, dim_in, hidden, n_sine,
              _ptr_array(weights), _opt_ptr_array(act), _opt_ptr_array(deriv), _opt_ptr_array(dz),
              _ptr_array(d_weights), _ptr_array(d_biases), 1 if head_done else 0, _ptr(ws),
              ws.numel() * 4, _stream())


# --------------------------------------------------------------------------- loss / optimiser
def mse_loss(pred, target, loss_out, d_pred=None, grad_divisor: float = 1.0):
    """loss_out[0] += mean((pred - target)^2); d_pred = 2 (pred - target) / (N * divisor)."""
    _gpu(pred, target, loss_out, d_pred)
    if not (pred.is_contiguous() and target.is_contiguous()):
        raise ValueError("mse_loss needs contiguous tensors")
    if pred.numel() != target.numel():
        raise ValueError("pred and target differ in size")
    _lib.call("mri_mse_loss", _ptr(pred), _ptr(target), pred.numel(), float(grad_divisor),
              _ptr(loss_out), _ptr(d_pred), _stream())
    return loss_out


class MSELossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        loss = torch.zeros(1, device=pred.device, dtype=torch.float32)
        d_pred = torch.empty_like(pred, memory_format=torch.contiguous_format)
        mse_loss(pred.contiguous(), target.contiguous(), loss, d_pred)
        ctx.save_for_backward(d_pred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        return d_pred * g, None


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step,
              grad_scale: float = 1.0):
    _gpu(param, grad, exp_avg, exp_avg_sq)
    _lib.call("mri_adam_step", _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
              param.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step),
              float(grad_scale), _stream())


# --------------------------------------------------------------------------- batch producer
def sample_indices(seed: int, first: int, lo: int, hi: int, n: int, out=None, device="cuda"):
    if out is None:
        out = torch.empty(n, device=device, dtype=torch.int64)
    _gpu(out)
    _lib.call("mri_sample_indices", C.c_uint64(seed & (2 ** 64 - 1)), first, lo, hi, n, _ptr(out),
              _stream())
    return out


def gather_batch(idx, shape: Sequence[int], axes: torch.Tensor, axis_offset: Sequence[int],
                 volume: Optional[torch.Tensor], coords=None, target=None):
    _gpu(idx, axes, volume, coords, target)
    n, dim = idx.numel(), len(shape)
    if coords is None:
        coords = torch.empty((n, dim), device=idx.device, dtype=torch.float32)
    if target is None and volume is not None:
        target = torch.empty((n, 1), device=idx.device, dtype=torch.float32)
    shp = (C.c_int64 * dim)(*[int(s) for s in shape])
    off = (C.c_int64 * dim)(*[int(o) for o in axis_offset])
    _lib.call("mri_gather_batch", _ptr(idx), n, dim, shp, _ptr(axes), off, _ptr(volume),
              _ptr(coords), _ptr(target), _stream())
    return coords, target
