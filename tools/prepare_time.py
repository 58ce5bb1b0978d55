"""The counting stage of the table gradient (mri_hashgrid_backward_prepare) alone on the GPU, config-4 batch:
python tools/prepare_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import _lib, encoding, ops
_lib.load()
dev = torch.device("cuda", 0)
n = 1 << 18
enc = encoding.MultiResHashGrid(3, n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                                finest_resolution=16 * 1.4 ** 15).cuda()
x = torch.rand(n, 3, device=dev)
ws = ops.backward_workspace(enc.desc, n, dev)
for _ in range(5):
    ops.hashgrid_backward_prepare(enc.desc, x, 2, ws=ws)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    ops.hashgrid_backward_prepare(enc.desc, x, 2, ws=ws)
b.record()
torch.cuda.synchronize()
print("prepare (2 memsets + count + chunk scan + prefix) alone: %.1f us per call" % (a.elapsed_time(b) / 50 * 1e3))
