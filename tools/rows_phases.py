#!/usr/bin/env python3
"""Where a wave of the rows-in-registers SIREN forward kernel (csrc/siren_rows.hip) spends its cycles
(tools-only profile build).

    python tools/build_variant.py --name=libmri_sprof.so -DSIREN_PROFILE     # here
    MRI_LIB=tools/libmri_sprof.so python tools/rows_phases.py [train]       # on the GPU box
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mri_interpolation_amd import _lib, models, trainer

lib = _lib.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "infer"  # infer | train | loss | bwd
train = mode != "infer"
net = models.SirenNet(3, 256, 1, 5).cuda()
st = trainer.FusedStep(net, net.configure_optimizers())
st.chain_loss = False
n = 1 << 20
x = torch.rand(n, 3, device="cuda") * 2 - 1
buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device="cuda")
y = torch.rand(n, 1, device="cuda")
def run():
    if mode in ("loss", "bwd"):
        st._chain_loss_pass(x, y, True, 1.0)
    else:
        st.forward(x, train=train)
if mode in ("loss", "bwd"):
    st.chain_loss = True
run()
torch.cuda.synchronize()
lib.mri_debug_set_rows_profile.argtypes = [C.c_void_p, C.c_int]
assert lib.mri_debug_set_rows_profile(C.c_void_p(buf.data_ptr()), 1 if mode == "bwd" else 0) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
run()
b.record()
torch.cuda.synchronize()
t = buf.view(256, 4, 8).double()
names = ["first layer", "operand split", "MFMAs + interleaved epilogue", "chunk wait + barrier + DMA issue",
         "last tile's epilogue", "head (+ loss tail)"]
if mode == "bwd":  # (the forward kernel of the pass wrote first; the backward kernel's counters are what is left)
    names = ["head phase", "operand split", "MFMAs + epilogue", "chunk wait + barrier", "tail stores + bias sums",
             "first layer's weight gradient", "derivative tile wait"]
total = t.sum(dim=2).mean()
groups = n / 128 / 256
print(f"{mode}: {a.elapsed_time(b):.3f} ms with counters; {total:.0f} cycles per wave, "
      f"{total / groups:.0f} per 128-row group (MFMA issue: 4 layers x 768 x 32 = 98304)")
for i, nm in enumerate(names):
    v = t[:, :, i].mean()
    print(f"  {nm:34s} {v / groups:9.0f} cycles per group  {100 * v / total:5.1f} %")
print("  per wave totals:", " ".join(f"{float(v) / groups:.0f}" for v in t.sum(dim=2).mean(dim=0)))
