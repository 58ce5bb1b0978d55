"""Does the order of a training batch matter to the lookup / table-gradient kernels?  Times both on a
shuffled config-4 batch and on the same batch sorted in Morton (Z) order: python tools/sorted_batch_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import _lib, encoding, ops
_lib.load()
torch.manual_seed(0)
n = 1 << 18
enc = encoding.MultiResHashGrid(3, n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                                base_resolution=16, finest_resolution=16 * 1.4 ** 15).cuda()
idx = torch.randperm(256 ** 3, device="cuda")[:n]          # a 1/64 sample of the 256^3 voxels
v = torch.stack([idx // 65536, (idx // 256) % 256, idx % 256], dim=1)
x = (v.float() / 255.0).contiguous()
def part(a):  # spread the low 8 bits of a
    a = a.long()
    a = (a | (a << 16)) & 0x0000FF0000FF
    a = (a | (a << 8)) & 0x00F00F00F00F
    a = (a | (a << 4)) & 0x0C30C30C30C3
    a = (a | (a << 2)) & 0x249249249249
    return a
morton = part(v[:, 0]) | (part(v[:, 1]) << 1) | (part(v[:, 2]) << 2)
xs = x[torch.argsort(morton)].contiguous()
xr = x[torch.argsort(idx)].contiguous()                    # raster order
d = torch.randn(enc.output_dim, n, device="cuda") * 1e-3
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
out = torch.empty(enc.output_dim, n, device="cuda")
g = torch.zeros_like(enc.table.data)
for name, c in (("shuffled", x), ("raster  ", xr), ("morton  ", xs)):
    f = timed(lambda: ops.hashgrid_forward(enc.desc, c, enc.table.data, out=out, feature_major=True))
    b = timed(lambda: ops.hashgrid_backward(enc.desc, c, d, g, feature_major=True, method=2, overwrite=True))
    print("%s: lookup %.4f ms   table gradient (count + scatter + accumulate on one stream) %.4f ms" % (name, f, b), flush=True)

# per level (round 3): where the order helps the lookup and where it hurts the table gradient -- one level per call
# (level_mask; the lookup of one level alone through a one-level grid descriptor is not available, so the lookup
# is timed for the whole grid above and per level only the table gradient is split)
if "--levels" in sys.argv:
    print("table gradient per level (count + scatter + accumulate of that level alone), ms: shuffled / morton")
    for l in range(enc.desc.n_levels):
        t = []
        for c in (x, xs):
            t.append(timed(lambda: ops.hashgrid_backward(enc.desc, c, d, g, feature_major=True, method=2, overwrite=True,
                                                         level_mask=1 << l), reps=20))
        print("  level %2d  res %5d  slots %7d : %.4f / %.4f" % (l, int(enc.desc.resolution[l][0]),
                                                               int(enc.desc.table_size[l]), t[0], t[1]), flush=True)
if "--split" in sys.argv:  # the dense levels (0-2) and the record levels (3-15) as two calls
    for name, mask in (("dense levels 0-2 ", 0x7), ("record levels 3-15", 0xFFF8)):
        t = [timed(lambda: ops.hashgrid_backward(enc.desc, c, d, g, feature_major=True, method=2, overwrite=True,
                                                 level_mask=mask)) for c in (x, xr, xs)]
        print("%s: shuffled %.4f  raster %.4f  morton %.4f ms" % (name, *t), flush=True)
if "--transposed" in sys.argv:
    # Morton order, then inside blocks of `blk` consecutive coordinates a transposition: a wave's 64 lanes take every
    # (blk / 64)-th coordinate of the block (64 different neighbourhoods: no equal slots inside a wave), consecutive
    # waves take their Morton neighbours (the same cache lines a moment later, on the same CU)
    order = torch.argsort(morton)
    for blk in (64, 256, 1024, 4096, 16384, 65536):
        q = torch.arange(n, device="cuda")
        inner = q % blk
        w = blk // 64
        pos = (q - inner) + (inner % w) * 64 + inner // w if w > 0 else q
        xt = torch.empty_like(x)
        xt[pos] = x[order]
        f = timed(lambda: ops.hashgrid_forward(enc.desc, xt, enc.table.data, out=out, feature_major=True))
        b = timed(lambda: ops.hashgrid_backward(enc.desc, xt, d, g, feature_major=True, method=2, overwrite=True))
        bd = timed(lambda: ops.hashgrid_backward(enc.desc, xt, d, g, feature_major=True, method=2, overwrite=True, level_mask=0x7))
        print("morton, transposed in blocks of %6d: lookup %.4f ms   table gradient %.4f ms   (dense levels alone %.4f)" % (blk, f, b, bd), flush=True)
if "--coarse" in sys.argv:
    # how many key bits does the lookup's gain need?  stable sort on the top `bits` bits of the 24-bit Morton key,
    # then the transposition in blocks of 16384
    for bits in (9, 12, 15, 18, 24):
        order = torch.sort(morton >> (24 - bits), stable=True).indices
        q = torch.arange(n, device="cuda")
        inner = q % 16384
        pos = (q - inner) + (inner % 256) * 64 + inner // 256
        xt = torch.empty_like(x)
        xt[pos] = x[order]
        f = timed(lambda: ops.hashgrid_forward(enc.desc, xt, enc.table.data, out=out, feature_major=True))
        b = timed(lambda: ops.hashgrid_backward(enc.desc, xt, d, g, feature_major=True, method=2, overwrite=True))
        print("top %2d key bits, transposed: lookup %.4f ms   table gradient %.4f ms" % (bits, f, b), flush=True)
