"""Oracle: multiresolution hash-grid encoding (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference `encoding.py`:
  PRIMES                      encoding.py:40
  fast_hash                   encoding.py:69-78
  _HashGrid.__init__/forward  encoding.py:81-128
  MultiResHashGrid            encoding.py:131-191   (isotropic, int resolutions)
  _HashGridV2/MultiResHashGridV2  encoding.py:194-336 (per-axis resolutions)

Two restatements live here:
  * `encode` / `level_lookup`: PyTorch-CPU fp32, the same ATen op sequence the
    reference issues (mul, long, sub, where, prod, embedding, mul, sum, cat), so
    that fp32 rounding is the reference's; autograd through it gives the table
    gradient the reference's `nn.Embedding` would receive.
  * `hash_u32` / `encode_loops`: NumPy uint32 / pure-Python loops, an
    independent statement of the integer arithmetic for small cases.
"""
import math
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

# encoding.py:40 -- one multiplier per axis, axis 0 is left unmultiplied.
PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)

Resolution = Union[int, float, Sequence[float]]


# --------------------------------------------------------------------------- geometry
def level_geometry(dim: int, n_levels: int, log2_hashmap_size: int,
                   base_resolution, finest_resolution) -> Tuple[List[List[int]], List[int]]:
    """Per-level resolutions (one int per axis) and table sizes.

    Isotropic (int base):   encoding.py:168-176 -- note the growth exponent divides
    by (base_resolution - 1), not (n_levels - 1).
    Anisotropic (tuples):   encoding.py:310-321 -- table size from max(res)**dim.
    """
    cap = 2 ** log2_hashmap_size
    res_per_level, sizes = [], []
    if isinstance(base_resolution, (int, float)) and not isinstance(base_resolution, bool):
        b = math.exp((math.log(finest_resolution) - math.log(base_resolution))
                     / (base_resolution - 1))
        for l in range(n_levels):
            r = math.floor(base_resolution * (b ** l))
            res_per_level.append([r] * dim)
            sizes.append(min(r ** dim, cap))
    else:
        growth = [math.exp((math.log(fr) - math.log(br)) / (br - 1))
                  for br, fr in zip(base_resolution, finest_resolution)]
        for l in range(n_levels):
            r = [math.floor(br * (g ** l)) for g, br in zip(growth, base_resolution)]
            res_per_level.append(r)
            sizes.append(min(max(r) ** dim, cap))
    return res_per_level, sizes


# --------------------------------------------------------------------------- hash
def hash_u32(idx: np.ndarray, table_size: int) -> np.ndarray:
    """uint32 restatement of fast_hash (encoding.py:69-78).

    idx: integer array (..., D), may hold negative values (two's complement wrap,
    exactly what `(ind * prime) & 0xFFFFFFFF` does on int64).
    """
    d = idx.shape[-1]
    v = idx.astype(np.int64).astype(np.uint32)  # wrap negatives mod 2^32
    acc = np.zeros(idx.shape[:-1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for a in range(d):
            acc ^= v[..., a] * np.uint32(PRIMES[a])
    return (acc % np.uint32(table_size)).astype(np.int64)


def hash_torch(idx: torch.Tensor, table_size: int) -> torch.Tensor:
    """int64 ATen sequence the reference runs (mul, and, xor, remainder)."""
    d = idx.shape[-1]
    primes = torch.tensor(PRIMES[:d], dtype=torch.int64)
    v = (idx * primes) & 0xFFFFFFFF
    acc = v[..., 0]
    for a in range(1, d):
        acc = acc ^ v[..., a]
    return acc % table_size


# --------------------------------------------------------------------------- lookup
def corner_mask(dim: int) -> torch.Tensor:
    """(2^D, D) bool: True where corner n takes the floor vertex on axis d
    (bit d of n clear) -- encoding.py:102-106."""
    n = torch.arange(1 << dim).unsqueeze(1)
    d = torch.arange(dim).unsqueeze(0)
    return ((n >> d) & 1) == 0


def level_lookup(x: torch.Tensor, table: torch.Tensor, resolution: Resolution) -> torch.Tensor:
    """One level: encoding.py:108-128 (and :232-270 for per-axis resolution).

    x (..., D) float32; table (T, F) float32.  Returns (..., F).
    """
    dim = x.shape[-1]
    if isinstance(resolution, (int, float)):
        pos = x * resolution
    else:
        pos = x * torch.tensor(list(resolution), dtype=x.dtype)  # float32 in the reference (encoding.py:205)
    cell = pos.long()                      # truncation toward zero
    frac = pos - cell.to(pos.dtype).detach()  # `.float()` in the reference; float64 only as a test yardstick
    cell = cell.unsqueeze(-2)
    frac = frac.unsqueeze(-2)
    floor_side = corner_mask(dim).reshape((1,) * (x.dim() - 1) + (1 << dim, dim))
    vertex = torch.where(floor_side, cell, cell + 1)
    w_axis = torch.where(floor_side, 1 - frac, frac)
    w = w_axis.prod(dim=-1, keepdim=True)
    slot = hash_torch(vertex, table.shape[0])
    rows = F.embedding(slot, table)
    return torch.sum(rows * w, dim=-2)


def encode(x: torch.Tensor, tables: Sequence[torch.Tensor],
           resolutions: Sequence[Resolution]) -> torch.Tensor:
    """All levels, features ordered [l0f0, l0f1, l1f0, ...] -- encoding.py:190-191.

    `resolutions[l]` is an int for the isotropic encoder (the reference multiplies
    by a Python scalar) or a per-axis list for V2 (a float32 tensor)."""
    return torch.cat([level_lookup(x, t, r) for t, r in zip(tables, resolutions)], dim=-1)


def level_slots_and_weights(x: torch.Tensor, table_size: int, resolution: Resolution):
    """(slots (n, 2^D) int64, weights (n, 2^D) float32) of one level: the integer and f32 steps of
    `level_lookup` up to, not including, the table access (encoding.py:108-126)."""
    dim = x.shape[-1]
    if isinstance(resolution, (int, float)):
        pos = x * resolution
    else:
        pos = x * torch.tensor(list(resolution), dtype=torch.float32)
    cell = pos.long()
    frac = pos - cell.float()
    cell, frac = cell.unsqueeze(-2), frac.unsqueeze(-2)
    floor_side = corner_mask(dim).reshape((1,) * (x.dim() - 1) + (1 << dim, dim))
    vertex = torch.where(floor_side, cell, cell + 1)
    w = torch.where(floor_side, 1 - frac, frac).prod(dim=-1)
    return hash_torch(vertex, table_size), w


def table_gradient_f64(x: torch.Tensor, d_out: torch.Tensor, sizes: Sequence[int],
                       resolutions: Sequence[Resolution], n_features: int):
    """The float64 yardstick of the table gradient (autograd of encoding.py:127-128): per level,
    (sum, sum of magnitudes, number) of the products w * g over the corners that hash to each slot, with the f32 weights
    and f32 incoming gradients the reference's backward multiplies, but every product and every
    addition in float64.  `sum of magnitudes` bounds what ANY f32 evaluation can lose to
    cancellation: an f32 product is within 2^-24 of its magnitude.  d_out: (n, L * F)."""
    out = []
    for l, (t, r) in enumerate(zip(sizes, resolutions)):
        slot, w = level_slots_and_weights(x, t, r)
        g = d_out[:, l * n_features:(l + 1) * n_features].double()
        contrib = w.double().unsqueeze(-1) * g.unsqueeze(-2)          # (n, 2^D, F)
        flat_slot = slot.reshape(-1)
        total = torch.zeros(t, n_features, dtype=torch.float64)
        mag = torch.zeros(t, n_features, dtype=torch.float64)
        total.index_add_(0, flat_slot, contrib.reshape(-1, n_features))
        mag.index_add_(0, flat_slot, contrib.abs().reshape(-1, n_features))
        count = torch.bincount(flat_slot, minlength=t)
        out.append((total, mag, count))
    return out


def resolutions_for(dim, n_levels, log2_hashmap_size, base_resolution, finest_resolution):
    """(resolution argument per level for `encode`, table sizes)."""
    res, sizes = level_geometry(dim, n_levels, log2_hashmap_size,
                                base_resolution, finest_resolution)
    if isinstance(base_resolution, (int, float)):
        return [r[0] for r in res], sizes
    return res, sizes


def init_tables(sizes: Sequence[int], n_features: int, seed: int,
                scale: float = 1e-4) -> List[torch.Tensor]:
    """Deterministic stand-in for the reference's U(-1e-4, 1e-4) init
    (encoding.py:95-96); values come from oracle.detrand, not torch's RNG."""
    from . import detrand
    out = []
    for l, t in enumerate(sizes):
        a = detrand.uniform(t * n_features, seed * 1000 + l, -scale, scale)
        out.append(torch.from_numpy(a.reshape(t, n_features).copy()))
    return out


# --------------------------------------------------------------------------- loops
def encode_loops(x: np.ndarray, tables: Sequence[np.ndarray],
                 resolutions: Sequence[Sequence[float]]) -> np.ndarray:
    """Pure-Python/NumPy-scalar statement for SMALL inputs: one coordinate, one
    level, one corner at a time, float32 scalar arithmetic in the same order as
    encoding.py:111-128.  Independent of torch."""
    x = np.asarray(x, dtype=np.float32)
    n, dim = x.shape
    feats = tables[0].shape[1]
    out = np.zeros((n, len(tables) * feats), dtype=np.float32)
    for l, (table, res) in enumerate(zip(tables, resolutions)):
        size = table.shape[0]
        for i in range(n):
            pos = [np.float32(x[i, a]) * np.float32(res[a]) for a in range(dim)]
            cell = [int(np.trunc(p)) for p in pos]
            frac = [np.float32(p) - np.float32(c) for p, c in zip(pos, cell)]
            acc = np.zeros(feats, dtype=np.float32)
            for corner in range(1 << dim):
                w = np.float32(1.0)
                h = 0
                for a in range(dim):
                    hi = (corner >> a) & 1
                    v = cell[a] + hi
                    w = np.float32(w * (frac[a] if hi else np.float32(1.0) - frac[a]))
                    h ^= (v * PRIMES[a]) & 0xFFFFFFFF
                acc = acc + table[h % size] * w
            out[i, l * feats:(l + 1) * feats] = acc
    return out


def frequency_encode(x: torch.Tensor, n_levels: int) -> torch.Tensor:
    """encoding.py:43-66 (`Frequency`): per input axis, [sin(2^0 x) .. sin(2^(L-1) x),
    cos(2^0 x) .. cos(2^(L-1) x)]; output (..., dim * 2L), axis-major."""
    freqs = 2.0 ** torch.linspace(0.0, n_levels - 1, n_levels)
    v = x.unsqueeze(-1) * freqs
    return torch.cat((torch.sin(v), torch.cos(v)), dim=-1).flatten(-2, -1)
