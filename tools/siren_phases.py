#!/usr/bin/env python3
"""Where a wave of the LDS-image SIREN forward kernel (csrc/siren_chain.hip) spends its cycles (tools-only profile build;
for the rows kernels of width 256 see tools/rows_phases.py).

    python tools/build_variant.py --name=libmri_sprof.so -DSIREN_PROFILE     # here
    MRI_LIB=tools/libmri_sprof.so python tools/siren_phases.py [train]      # on the GPU box
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mri_interpolation_amd import _lib, models, ops, trainer

lib = _lib.load()
lib.mri_set_option(b"siren_rows", 0)  # the LDS-image kernel (csrc/siren_chain.hip); the rows kernels: tools/rows_phases.py
train = len(sys.argv) > 1 and sys.argv[1] == "train"
net = models.SirenNet(3, 256, 1, 5).cuda()
st = trainer.FusedStep(net, net.configure_optimizers())
n = 1 << 20
x = torch.rand(n, 3, device="cuda") * 2 - 1
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
st.forward(x, train=train)
torch.cuda.synchronize()
lib.mri_debug_set_siren_profile.argtypes = [C.c_void_p]
assert lib.mri_debug_set_siren_profile(C.c_void_p(buf.data_ptr())) == 0
st.forward(x, train=train)
torch.cuda.synchronize()
t = buf.view(256, 8, 8).double()
names = ["tile top (x)", "first layer", "chunk wait+barrier", "DMA issue + drip", "reads + MFMAs",
         "epilogue arithmetic", "epilogue barrier + image", "head"]
total = t.sum(dim=2).mean()
tiles = n / 64 / 256
print(f"{'train' if train else 'inference'} forward: {total:.0f} cycles per wave, {total / tiles:.0f} per tile")
for i, nm in enumerate(names):
    v = t[:, :, i].mean()
    print(f"  {nm:28s} {v / tiles:9.0f} cycles per tile  {100 * v / total:5.1f} %")
print("  (reads + MFMAs: 4 layers x 8 chunks x 32 MFMAs x 64 cycles = 65536 cycles of issue per wave and tile,"
      " two waves share a SIMD's pipe)")
