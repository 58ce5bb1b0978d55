#!/usr/bin/env python3
"""Where the two-barrier decoder kernel (csrc/mlp_x3.hip: tiny_mlp_x3p_kernel) spends its cycles: shader-clock
counters of a tools-only build (-DMRI_X3_PROFILE), config-4 shape.

    python tools/x3_segments.py --build-only     # here (cross-compile: tools/libmri_x3prof.so), the .so travels
    python tools/x3p_segments.py                 # on the GPU box
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("MRI_X3PROF_LIB") or os.path.join(ROOT, "tools", "libmri_x3prof.so")
import torch

lib = C.CDLL(LIB)
lib.mri_set_option(b"mlp_x3", C.c_int32(1))
n, k_in, h = 1 << 18, 32, 128
dev = "cuda"
x = torch.randn(k_in, n, device=dev) * 0.1
t = torch.rand(n, device=dev)
w1 = torch.randn(h, k_in, device=dev) / k_in ** 0.5
w2 = torch.randn(h, h, device=dev) / h ** 0.5
w3 = torch.randn(1, h, device=dev) / h ** 0.5
b1, b2, b3 = torch.zeros(h, device=dev), torch.zeros(h, device=dev), torch.zeros(1, device=dev)
grads = [torch.zeros_like(p) for p in (w1, b1, w2, b2, w3, b3)]
loss = torch.zeros(1, device=dev)
dx = torch.empty(k_in, n, device=dev)
lib.mri_tiny_mlp_workspace_bytes.restype = C.c_int64
ws = torch.empty(lib.mri_tiny_mlp_workspace_bytes(k_in, h, C.c_int64(n)) // 4, device=dev)
blocks = 256
prof = torch.zeros(blocks * 8 * 32, dtype=torch.int64, device=dev)
assert lib.mri_debug_set_x3_profile(C.c_void_p(prof.data_ptr())) == 0
P = C.c_void_p
args = [P(x.data_ptr()), P(t.data_ptr()), C.c_int64(n), C.c_int32(k_in), C.c_int32(h)]
args += [P(p.data_ptr()) for p in (w1, b1, w2, b2, w3, b3)]
args += [C.c_float(1.0)] + [P(g.data_ptr()) for g in grads] + [P(dx.data_ptr()), P(loss.data_ptr()), P(None), P(ws.data_ptr()),
                                                              C.c_int64(ws.numel() * 4), P(None)]
fn = lib.mri_tiny_mlp_train
lib.mri_last_error.restype = C.c_char_p
for _ in range(3):
    assert fn(*args) == 0, lib.mri_last_error()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
fn(*args)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b)
p = prof.cpu().reshape(blocks, 8, 32).double()
tiles = n / 32 / blocks
names = ["A: x loads, layer 1 (i+1)", "A: dx (i-1)", "A: dLoss/dy, masks, db2, dW3", "A: dW2", "A: barrier wait", "B: x -> LDS", "B: dz1",
         "B: dW1", "B: layer 2 (i+1)", "B: barrier wait"]
q = p.mean(dim=(0, 1)) / tiles
tot = float(q[:10].sum())
print(f"train, two-barrier kernel: {ms * 1e3:.1f} us with counters; {tiles:.0f} tiles per workgroup; {tot:.0f} cycles per tile")
for i, nm in enumerate(names):
    lo = p[:, :4, i].mean() / tiles
    hi = p[:, 4:, i].mean() / tiles
    print(f"  {nm:32s} {float(q[i]):7.0f}   (waves 0-3 {float(lo):6.0f}, waves 4-7 {float(hi):6.0f})")
print(f"  prologue {float(q[20]):.0f}  tail {float(q[18]):.0f}  epilogue {float(q[21]):.0f} cycles")
