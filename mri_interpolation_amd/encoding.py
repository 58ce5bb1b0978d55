"""Multiresolution hash-grid encoders with the reference's class surface, computed by the
gfx950 kernels in csrc/hashgrid.hip.

Mirrors reference `encoding.py`: `MultiResHashGrid` (encoding.py:131-191, isotropic integer
resolutions) and `MultiResHashGridV2` (encoding.py:273-336, per-axis resolutions): same
constructor arguments, `forward(x) -> (b..., n_levels * n_features_per_level)`, attributes
`input_dim`, `output_dim`, `levels[i].embedding.weight`, `levels[i].resolution`,
`levels[i].hashmap_size`, and the same state-dict keys (`levels.{i}.embedding.weight`).

MI355X-first storage: all level tables live in ONE flat (sum T_l, F) parameter (`table`) so
that the gather, the gradient scatter, the RCCL reduction and Adam each see one buffer;
`levels[i].embedding.weight` is a view of it.
"""
import math
from typing import List, Sequence, Tuple, Union

import torch
from torch import nn

from . import ops

PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)


def level_table(dim: int, n_levels: int, log2_hashmap_size: int, base_resolution,
                finest_resolution) -> Tuple[List[List[int]], List[int]]:
    """Resolution per level and axis, and table size per level.

    Isotropic: b = exp((ln finest - ln base) / (base - 1)), res_l = floor(base * b**l),
    T_l = min(res_l**dim, 2**log2T)                      (reference encoding.py:168-176;
    the exponent really divides by base-1).
    Anisotropic: the same per axis, T_l = min(max_d(res_l,d)**dim, 2**log2T)
                                                          (reference encoding.py:310-321).
    """
    limit = 1 << log2_hashmap_size
    if isinstance(base_resolution, (int, float)):
        bases, finests = [base_resolution] * dim, [finest_resolution] * dim
        isotropic = True
    else:
        bases, finests = list(base_resolution), list(finest_resolution)
        isotropic = False
        if len(bases) != dim or len(finests) != dim:
            raise ValueError(f"resolution tuples must have {dim} entries "
                             f"(got {len(bases)} and {len(finests)})")
    scale = [math.exp((math.log(f) - math.log(b)) / (b - 1)) for b, f in zip(bases, finests)]
    res, sizes = [], []
    for l in range(n_levels):
        r = [math.floor(b * (s ** l)) for b, s in zip(bases, scale)]
        res.append(r)
        edge = r[0] if isotropic else max(r)
        sizes.append(min(edge ** dim, limit))
    return res, sizes


class _TableView:
    """Stands in for `nn.Embedding` of one level: `.weight` is a view of the flat table."""

    def __init__(self, owner, index):
        self._owner, self._index = owner, index

    @property
    def weight(self) -> torch.Tensor:
        lo, hi = self._owner._row_span(self._index)
        return self._owner.table[lo:hi]

    @property
    def num_embeddings(self):
        return self._owner.sizes[self._index]

    @property
    def embedding_dim(self):
        return self._owner.n_features_per_level


class _LevelView:
    """One level as the reference's `_HashGrid` exposes it (encoding.py:81-106)."""

    def __init__(self, owner, index):
        self._owner, self._index = owner, index
        self.dim = owner.dim
        self.n_features = owner.n_features_per_level
        self.hashmap_size = owner.sizes[index]
        r = owner.resolutions[index]
        self.resolution = r[0] if owner.isotropic else torch.tensor(r, dtype=torch.float32)
        self.embedding = _TableView(owner, index)


class _HashGridBase(nn.Module):
    def __init__(self, dim, n_levels, n_features_per_level, log2_hashmap_size, base_resolution,
                 finest_resolution, isotropic):
        super().__init__()
        if dim > len(PRIMES):
            raise AssertionError(f"HashGrid only supports < {len(PRIMES)}-D inputs")
        self.dim = dim
        self.n_levels = n_levels
        self.n_features_per_level = n_features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.finest_resolution = finest_resolution
        self.isotropic = isotropic
        self.resolutions, self.sizes = level_table(dim, n_levels, log2_hashmap_size,
                                                   base_resolution, finest_resolution)
        self._starts = [0]
        for s in self.sizes:
            self._starts.append(self._starts[-1] + s)
        table = torch.empty(self._starts[-1], n_features_per_level)
        nn.init.uniform_(table, a=-0.0001, b=0.0001)  # reference encoding.py:95-96
        self.table = nn.Parameter(table)
        self.desc = ops.make_grid_desc(dim, self.resolutions, self.sizes, n_features_per_level)
        self.levels = [_LevelView(self, i) for i in range(n_levels)]
        # isotropic levels only: V2 holds its resolutions as float32 tensors, which never raise
        self._too_fine = isotropic and any(r[0] >= 2 ** 63 for r in self.resolutions)
        self.input_dim = dim
        self.output_dim = n_levels * n_features_per_level

    def _row_span(self, i):
        return self._starts[i], self._starts[i + 1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._too_fine:
            # the reference multiplies by the level's Python-int resolution (encoding.py:110);
            # torch refuses an int beyond int64 with exactly this error, at the first forward
            raise OverflowError("int too big to convert")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if torch.is_grad_enabled() and (self.table.requires_grad or x2.requires_grad):
            out = ops.HashGridFunction.apply(x2, self.table, self.desc)
        else:
            out = ops.hashgrid_forward(self.desc, x2, self.table.detach())
        return out.reshape(*lead, self.output_dim)

    # state-dict layout of the reference: one `levels.{i}.embedding.weight` per level
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for i in range(self.n_levels):
            lo, hi = self._row_span(i)
            t = self.table[lo:hi]
            destination[f"{prefix}levels.{i}.embedding.weight"] = t if keep_vars else t.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        flat = state_dict.pop(prefix + "table", None)
        if flat is not None:
            with torch.no_grad():
                self.table.copy_(flat)
            return
        for i in range(self.n_levels):
            key = f"{prefix}levels.{i}.embedding.weight"
            if key not in state_dict:
                if strict:
                    missing_keys.append(key)
                continue
            lo, hi = self._row_span(i)
            if tuple(state_dict[key].shape) != (hi - lo, self.n_features_per_level):
                error_msgs.append(f"size mismatch for {key}")
                continue
            with torch.no_grad():
                self.table[lo:hi].copy_(state_dict[key])


class MultiResHashGrid(_HashGridBase):
    """Drop-in for reference `encoding.MultiResHashGrid` (encoding.py:131-191)."""

    def __init__(self, dim: int, n_levels: int = 16, n_features_per_level: int = 2,
                 log2_hashmap_size: int = 15, base_resolution: int = 16,
                 finest_resolution: int = 512):
        super().__init__(dim, n_levels, n_features_per_level, log2_hashmap_size,
                         base_resolution, finest_resolution, isotropic=True)


class MultiResHashGridV2(_HashGridBase):
    """Drop-in for reference `encoding.MultiResHashGridV2` (encoding.py:273-336): per-axis
    base / finest resolutions (tuples of length dim)."""

    def __init__(self, dim: int, n_levels: int = 16, n_features_per_level: int = 2,
                 log2_hashmap_size: int = 15,
                 base_resolution: Union[int, Sequence[int]] = 16,
                 finest_resolution: Union[int, Sequence[int]] = 512):
        if isinstance(base_resolution, (int, float)):
            base_resolution = (base_resolution,) * dim
            finest_resolution = (finest_resolution,) * dim
        super().__init__(dim, n_levels, n_features_per_level, log2_hashmap_size,
                         tuple(base_resolution), tuple(finest_resolution), isotropic=False)


class Frequency(nn.Module):
    """NeRF positional encoding (reference encoding.py:43-66): per input axis
    [sin(2^0 x) .. sin(2^(L-1) x), cos(2^0 x) .. cos(2^(L-1) x)], one HIP kernel
    (csrc/frequency.hip) forward and one backward."""

    def __init__(self, dim: int, n_levels: int = 10):
        super().__init__()
        self.n_levels = n_levels
        assert self.n_levels > 0
        freqs = 2.0 ** torch.linspace(0.0, n_levels - 1, n_levels)
        self.register_buffer("freqs", freqs, persistent=False)
        self.input_dim = dim
        self.output_dim = dim * n_levels * 2

    def forward(self, x: torch.Tensor):
        return ops.FrequencyFunction.apply(x, self.n_levels)

