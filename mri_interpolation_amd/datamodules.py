"""Coordinate-batch producer with the reference's `MriImage` / `MriDataModule` surface.

Mirrors reference `datamodules.py`: `MriImage` (datamodules.py:123-172) holds
coords = meshgrid(linspace(0,1,s_d), 'ij') flattened C-order and pixels min-max normalised;
`MriDataModule` (datamodules.py:175-252) hands out a shuffled train loader, an ordered test
loader and a dense-grid `upsampling` loader.

MI355X-first: the reference's real wall-clock limiter is the per-sample `__getitem__` +
collate DataLoader.  Here the normalised volume lives in HBM and each batch
is generated ON DEVICE by two kernels (csrc/train_ops.hip): a keyed permutation of the voxel
range (a without-replacement shuffle, one key per epoch) and a gather that turns flat voxel
indices into (coords, targets) through per-axis `torch.linspace` tables built on the host
(so coordinates equal the reference's bit for bit, SURVEY.md section 7 hard part 6).
"""
import itertools
import math
from typing import Optional, Sequence

import numpy as np
import torch

from . import nifti, ops, parallel

# Synthetic phantom of SURVEY.md 8(d) (amplitude, cx, cy, cz, sigma), frozen constants.
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


def phantom_volume(shape: Sequence[int], device="cuda") -> torch.Tensor:
    """Analytic test volume (12 Gaussian blobs + a sinusoid), min-max normalised to [0, 1]
    float32, evaluated at linspace(0,1,s) voxel centres in float64 on `device`.  3-D shapes give
    the phantom of SURVEY.md 8(d); a 4-D shape (x, y, z, t) adds time: the blobs drift along x
    and breathe in amplitude, frame by frame (a stand-in for the dynamic sample volume)."""
    if len(shape) == 4:
        frames = []
        for t in torch.linspace(0.0, 1.0, shape[3], dtype=torch.float64).tolist():
            frames.append(_phantom_frame(shape[:3], device, t))
        v = torch.stack(frames, dim=-1)
    else:
        v = _phantom_frame(shape, device, None)
    v = (v - v.min()) / (v.max() - v.min())
    return v.to(torch.float32)


def _phantom_frame(shape, device, t):
    ax = [torch.linspace(0.0, 1.0, s, dtype=torch.float64, device=device) for s in shape]
    x, y, z = torch.meshgrid(*ax, indexing="ij")
    v = 0.05 * torch.sin(2.0 * math.pi * (7.0 * x + 11.0 * y + 13.0 * z))
    for m, (a, cx, cy, cz, s) in enumerate(PHANTOM_BLOBS):
        if t is not None:
            cx = cx + 0.08 * math.sin(2.0 * math.pi * (t + m / 12.0))
            a = a * (1.0 + 0.25 * math.cos(2.0 * math.pi * (t + m / 7.0)))
        v = v + a * torch.exp(-((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) / (2.0 * s * s))
    return v


class MriImage:
    """Dataset for implicit-representation training: coordinates in x, intensity in y."""

    def __init__(self, config=None, image_path: Optional[str] = None, norm_siren: bool = False,
                 volume=None, device=None, frames: Optional[slice] = None, *args, **kwargs):
        """`frames` keeps a subset of the LAST axis (e.g. slice(0, None, 2) = even time frames)
        while coordinates still come from the FULL grid -- the held-out-frame protocol of
        reference interp.py:35 / legacy_code/hash_experimentation.py:284,313-317.  Intensities
        are normalised with the min / max of the full volume."""
        if volume is None:
            path = image_path if image_path else config.image_path
            volume = nifti.load(path)
        self.device = torch.device(device if device is not None else "cuda")
        vol = torch.as_tensor(volume, dtype=torch.float32).to(self.device)
        full_shape = tuple(int(s) for s in vol.shape)
        self.norm_siren = norm_siren
        lo = -1.0 if norm_siren else 0.0
        # axes built with torch.linspace on the CPU, as the reference does, then uploaded
        axes = [torch.linspace(lo, 1.0, s) for s in full_shape]
        v_min, v_max = torch.min(vol), torch.max(vol)
        if frames is not None:
            axes[-1] = axes[-1][frames]
            vol = vol[..., frames].contiguous()
        self.shape = tuple(int(s) for s in vol.shape)
        self.dim_in = len(self.shape)
        self.axis_offset = [0]
        for a in axes[:-1]:
            self.axis_offset.append(self.axis_offset[-1] + a.numel())
        self.axes = torch.cat(axes).to(self.device)
        pix = vol.flatten()
        pix = (pix - v_min) / (v_max - v_min)
        if norm_siren:
            pix = pix * 2 - 1
        self.pixels = pix.unsqueeze(-1).contiguous()  # (N, 1), C-order flatten

    def __len__(self):
        return self.pixels.shape[0]

    def batch(self, idx: torch.Tensor, coords=None, target=None):
        """(coords (n, D), pixels (n, 1)) for flat voxel indices `idx` (int64 on device)."""
        return ops.gather_batch(idx, self.shape, self.axes, self.axis_offset, self.pixels,
                                coords, target)

    def __getitem__(self, idx):
        single = isinstance(idx, int)
        t = torch.as_tensor([idx] if single else idx, dtype=torch.int64, device=self.device)
        c, p = self.batch(t.reshape(-1))
        return (c[0], p[0]) if single else (c, p)

    @property
    def coords(self) -> torch.Tensor:
        """All coordinates (N, D), generated on demand."""
        idx = torch.arange(len(self), device=self.device)
        return self.batch(idx)[0]


class DeviceLoader:
    """Iterable of on-device (coords, targets) batches over a voxel range [lo, hi)."""

    def __init__(self, dataset: MriImage, batch_size: int, shuffle: bool, lo: int = 0,
                 hi: Optional[int] = None, seed: int = 1337, drop_last: bool = False,
                 steps: Optional[int] = None):
        """`steps` fixes the number of batches per epoch and makes EVERY batch `batch_size` rows:
        the epoch's permutation of [lo, hi) is walked cyclically (a short range wraps around).
        Data-parallel ranks whose slabs differ in size use it to run the same number of
        equal-sized steps (mean of equal-sized local means = global mean; no rank leaves the
        collective early)."""
        self.ds, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.lo, self.hi = lo, len(dataset) if hi is None else hi
        self.seed, self.epoch, self.drop_last = seed, 0, drop_last
        self.steps = None if steps is None else int(steps)
        if self.steps is not None and self.hi <= self.lo:
            raise ValueError("a loader with a fixed step count needs a non-empty voxel range")

    def __len__(self):
        if self.steps is not None:
            return self.steps
        n = self.hi - self.lo
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def indices(self, first: int, n: int, epoch: Optional[int] = None, out=None) -> torch.Tensor:
        epoch = self.epoch if epoch is None else epoch
        if self.shuffle:
            return ops.sample_indices(self.seed + 7919 * epoch, first, self.lo, self.hi, n,
                                      out=out, device=self.ds.device)
        idx = torch.arange(first, first + n, device=self.ds.device)
        if self.steps is not None:
            idx = idx % (self.hi - self.lo)
        idx = idx + self.lo
        return idx if out is None else out.copy_(idx)

    def span(self, b: int):
        """(position of batch b's first voxel in the epoch's permutation of [lo, hi), its size)."""
        first = b * self.batch_size
        if self.steps is not None:  # cyclic walk: always a full batch
            return first, self.batch_size
        return first, min(self.batch_size, self.hi - self.lo - first)

    def batches(self, n_batches: Optional[int] = None):
        """Generator over `n_batches` consecutive batches (default: one epoch; -1: endless),
        running on into the following epochs; advances `self.epoch` as epochs complete."""
        per_epoch = len(self)
        if per_epoch == 0:
            return
        epoch0 = self.epoch
        count = itertools.count() if (n_batches is not None and n_batches < 0) \
            else range(per_epoch if n_batches is None else n_batches)
        for k in count:
            epoch, b = epoch0 + k // per_epoch, k % per_epoch
            first, n = self.span(b)
            self.epoch = epoch
            yield self.ds.batch(self.indices(first, n, epoch))
            if b == per_epoch - 1:
                self.epoch = epoch + 1

    def __iter__(self):
        return self.batches()


def sharded_steps(shape, batch_size: int, world: int) -> int:
    """Batches per epoch of a data-parallel run: what the LARGEST slab needs, so that every voxel
    is visited at least once per epoch (slabs differ whenever the slow axis does not divide by
    the world size; the reference's single loader rounds up the same way, datamodules.py:198)."""
    longest = max(hi - lo for lo, hi in (parallel.voxel_range(shape, r, world)
                                         for r in range(world)))
    return -(-longest // int(batch_size))


def sharded_loader(dataset: MriImage, batch_size: int, rank: int = 0, world: int = 1,
                   seed: int = 1337) -> DeviceLoader:
    """Shuffled training loader over `rank`'s slab.  One process: the reference's loader (last
    batch of an epoch may be short).  Several: every rank runs sharded_steps() batches of exactly
    `batch_size` rows per epoch, shorter slabs wrapping around their permutation."""
    lo, hi = parallel.voxel_range(dataset.shape, rank, world)
    steps = sharded_steps(dataset.shape, batch_size, world) if world > 1 else None
    return DeviceLoader(dataset, batch_size, shuffle=True, lo=lo, hi=hi, seed=seed + rank,
                        steps=steps)


class BatchPipeline:
    """Reused batch buffers for a training loop: while step k consumes its batch, later batches are
    produced into the other buffer by `produce_next()` / `produce_late()`, which the caller runs on
    whatever stream has room beside the step -- FusedStep queues them on its side stream behind the
    counting stage of the table gradient, whose completion the main stream awaits anyway (no extra
    cross-stream wait, which costs ~12 us of queue bubble on this GPU).  The counterpart of the
    reference DataLoader's worker processes (datamodules.py:198-205).

    `group` = R batches are produced by ONE pair of launches (sample + gather over R x batch_size
    positions of the epoch's permutation -- the same indices, coordinates and targets the R single
    launches give, bit for bit).  The two small kernels of a production run beside the lookup and cost
    the step ~8 us at BASELINE config 4 (bench.py --fixed-batch); with R = 8 they run once in 8 steps --
    measured: the lookup then takes 0.101 instead of 0.108 ms, the step the same 0.528 ms (eight times
    longer kernels every eighth step take from their neighbours what the short ones took), so the
    default stays 1.
    Groups never straddle an epoch (its last group is shorter).  Two buffers of R batches alternate.

    Ordering contract: group G+1 is produced during the FIRST step of group G, on a stream that waited
    for the main stream at that step's start (group G-1, the buffer's previous user, is then done), and
    every later step is ordered behind that stream (FusedStep: the batch event of the previous step)."""

    def __init__(self, loader: DeviceLoader, group: int = 1):
        self.loader = loader
        self.group = max(1, int(group))
        dev, bs = loader.ds.device, loader.batch_size
        rows = bs * self.group
        self.slots = [(torch.empty(rows, dtype=torch.int64, device=dev),
                       torch.empty(rows, loader.ds.dim_in, device=dev),
                       torch.empty(rows, 1, device=dev)) for _ in range(2)]
        self.per_epoch = len(loader)
        if self.per_epoch == 0:
            raise ValueError("empty loader")
        self.groups_per_epoch = -(-self.per_epoch // self.group)
        self.epoch0 = loader.epoch
        self.k = 0
        self._made = [-1, -1]  # group number each buffer holds
        self._due = None
        self._produce_group(0)  # on the current stream

    def _locate(self, k: int):
        """batch k -> (group number, epoch, first batch of the group in its epoch, batches in the group,
        position of batch k inside the group)."""
        e, b = divmod(k, self.per_epoch)
        j = b // self.group
        b0 = j * self.group
        return e * self.groups_per_epoch + j, self.epoch0 + e, b0, min(self.group, self.per_epoch - b0), b - b0

    def _produce_group(self, g: int):
        e, j = divmod(g, self.groups_per_epoch)
        b0 = j * self.group
        count = min(self.group, self.per_epoch - b0)
        first, _ = self.loader.span(b0)
        total = sum(self.loader.span(b0 + r)[1] for r in range(count))  # only an epoch's last batch is short
        idx, coords, target = self.slots[g % 2]
        self.loader.indices(first, total, self.epoch0 + e, out=idx[:total])
        self.loader.ds.batch(idx[:total], coords[:total], target[:total])
        self._made[g % 2] = g

    def _view(self, k: int):
        g, _, b0, _, r = self._locate(k)
        if self._made[g % 2] != g:  # a caller that never ran produce_late(): produce now, on this stream
            self._produce_group(g)
        lo = r * self.loader.batch_size
        n = self.loader.span(b0 + r)[1]
        _, coords, target = self.slots[g % 2]
        return coords[lo:lo + n], target[lo:lo + n]

    def current(self):
        """(coords, targets) of batch k; valid until the buffer's next production."""
        return self._view(self.k)

    def produce_next(self):
        """To be queued during step k: makes sure batch k+1 exists and returns its coordinates (the
        tensor current() yields after advance()), so that a caller can start coordinate-only work of
        the next step -- FusedStep counts the table-gradient records a step ahead.  With group = 1
        batch k+1 is produced here; with larger groups it was produced a group ago, and the NEXT
        group's production is left to produce_late() when this is the first step of a group."""
        g, _, _, _, r = self._locate(self.k)
        g_next = self._locate(self.k + 1)[0]
        if self._made[g_next % 2] != g_next:
            self._produce_group(g_next)  # batch k+1 does not exist yet (group = 1; one-batch groups)
        elif r == 0 and g_next == g and self._made[(g + 1) % 2] != g + 1:
            self._due = g + 1            # first of several steps on this group: the next one, late
        return self._view(self.k + 1)[0]

    def produce_late(self):
        """The deferred production of the next group, to be queued BEHIND whatever the step needs soon
        (FusedStep: behind the count of the next batch's records): its kernels are `group` times longer."""
        if self._due is not None:
            g, self._due = self._due, None
            self._produce_group(g)

    def advance(self):
        self.k += 1
        self.loader.epoch = self.epoch0 + self.k // self.per_epoch

    @property
    def batch_in_epoch(self) -> int:
        return self.k % self.per_epoch


class GridLoader:
    """Ordered batches of dense-grid coordinates (no targets: zeros, as the reference's
    mock loader yields -- datamodules.py:229-252)."""

    def __init__(self, shape, batch_size, norm_siren=False, device="cuda"):
        self.grid = MriImage(volume=np.zeros(shape, dtype=np.float32), norm_siren=norm_siren,
                             device=device)
        self.grid.pixels.zero_()
        self.loader = DeviceLoader(self.grid, batch_size, shuffle=False)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        return iter(self.loader)


class MriDataModule:
    """Takes ONE MRI image and returns coords and pixels (no train/val/test split)."""

    def __init__(self, config=None, volume=None, norm_siren: bool = False, *args, **kwargs):
        self.config = config
        self.volume = volume
        self.norm_siren = norm_siren
        self.dataset = self.train_ds = self.val_ds = self.test_ds = None

    def prepare_data(self) -> None:
        self.dataset = MriImage(config=self.config, volume=self.volume,
                                norm_siren=self.norm_siren)
        self.train_ds = self.test_ds = self.val_ds = self.dataset

    def setup(self, stage=None):
        pass

    def train_dataloader(self, rank: int = 0, world: int = 1) -> DeviceLoader:
        """Shuffled loader over this rank's slab of the volume (whole volume for world 1)."""
        return sharded_loader(self.train_ds, self.config.batch_size, rank, world,
                              seed=getattr(self.config, "seed", 1337))

    def val_dataloader(self) -> DeviceLoader:
        return DeviceLoader(self.val_ds, self.config.batch_size, shuffle=False)

    def test_dataloader(self) -> DeviceLoader:
        return DeviceLoader(self.test_ds, self.config.batch_size, shuffle=False)

    def upsampling(self, shape, batch_size, norm_siren: bool = False) -> GridLoader:
        return GridLoader(shape, batch_size, norm_siren)
