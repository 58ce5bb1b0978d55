"""Oracle: coordinate-batch producer (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference `datamodules.py`:
  MriImage.__init__   datamodules.py:135-166  (axes = torch.linspace, meshgrid 'ij',
                                               C-order flatten, min-max normalise)
  MriDataModule.upsampling  datamodules.py:229-252  (dense grid for interpolation)
and the synthetic phantom defined in SURVEY.md section 8(d) for BASELINE configs 2-4.
"""
from typing import Sequence, Tuple

import numpy as np
import torch

# 12 Gaussian blobs, parameters drawn once from
# numpy.random.default_rng(1337) (amplitude, cx, cy, cz, sigma), frozen here as data.
PHANTOM_BLOBS = (
    (0.914671, 0.705558, 0.440001, 0.208670, 0.167284),
    (0.429870, 0.424467, 0.387700, 0.165612, 0.108726),
    (0.944630, 0.773307, 0.166709, 0.735074, 0.175188),
    (0.962596, 0.471719, 0.178302, 0.719093, 0.185785),
    (0.912156, 0.693448, 0.651990, 0.195792, 0.077915),
    (0.381020, 0.590153, 0.840326, 0.689920, 0.083039),
    (0.435612, 0.299856, 0.210111, 0.831542, 0.050020),
    (0.539216, 0.668344, 0.835137, 0.700320, 0.054043),
    (0.647054, 0.243490, 0.233573, 0.305336, 0.168622),
    (0.928860, 0.588131, 0.640457, 0.688756, 0.073182),
    (0.304521, 0.433649, 0.395709, 0.780849, 0.044156),
    (0.459188, 0.214497, 0.293642, 0.376995, 0.179894),
)


def axes_for(shape: Sequence[int], norm_siren: bool = False):
    """datamodules.py:140-146: one torch.linspace per axis, [0,1] or [-1,1]."""
    lo = -1.0 if norm_siren else 0.0
    return [torch.linspace(lo, 1.0, s) for s in shape]


def coords_grid(shape: Sequence[int], norm_siren: bool = False) -> torch.Tensor:
    """(prod(shape), D) float32, last axis fastest -- datamodules.py:148,162-163."""
    grid = torch.stack(torch.meshgrid(*axes_for(shape, norm_siren), indexing="ij"), dim=-1)
    return grid.reshape(-1, len(shape))


def normalise(pixels: torch.Tensor, norm_siren: bool = False) -> torch.Tensor:
    """(N,1) targets -- datamodules.py:151-161,166."""
    p = pixels.flatten()
    p = (p - torch.min(p)) / (torch.max(p) - torch.min(p))
    if norm_siren:
        p = p * 2 - 1
    return p.unsqueeze(-1)


def dataset(volume: np.ndarray, norm_siren: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """What MriImage holds after __init__: (coords (N,D), pixels (N,1))."""
    pix = torch.from_numpy(np.ascontiguousarray(volume, dtype=np.float32))
    return coords_grid(volume.shape, norm_siren), normalise(pix, norm_siren)


def phantom(shape: Sequence[int]) -> np.ndarray:
    """Analytic volume of SURVEY.md 8(d), evaluated at linspace(0,1,s) voxel
    centres in float64 then min-max normalised to float32 [0,1]."""
    ax = [np.linspace(0.0, 1.0, s) for s in shape]
    x, y, z = np.meshgrid(*ax, indexing="ij")
    v = 0.05 * np.sin(2.0 * np.pi * (7.0 * x + 11.0 * y + 13.0 * z))
    for a, cx, cy, cz, s in PHANTOM_BLOBS:
        v = v + a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) / (2.0 * s * s))
    v = (v - v.min()) / (v.max() - v.min())
    return v.astype(np.float32)


def slab_range(n_slow: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slab of the slowest axis owned by `rank`."""
    base, rem = divmod(n_slow, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
