"""Decoder kernel timings at the config-4 shape (32 -> 128 -> 128 -> 1, B = 2^18), f32-MFMA team
kernel vs the bf16x3 kernel: python tools/mlp_time.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import _lib, ops
_lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
torch.manual_seed(0)
k_in, H = 32, int(sys.argv[2]) if len(sys.argv) > 2 else 128
params = [(torch.randn(H, k_in, device="cuda") * 0.2, torch.randn(H, device="cuda") * 0.1),
          (torch.randn(H, H, device="cuda") * 0.1, torch.randn(H, device="cuda") * 0.1),
          (torch.randn(1, H, device="cuda") * 0.1, torch.randn(1, device="cuda") * 0.1)]
x = torch.rand(k_in, n, device="cuda") * 2 - 1
t = torch.rand(n, 1, device="cuda")
def run(x3):
    _lib.set_option("mlp_x3", x3)
    grads = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in params]
    dx = torch.empty_like(x); y = torch.empty(n, 1, device="cuda"); loss = torch.zeros(1, device="cuda")
    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    tr = timed(lambda: ops.tiny_mlp_train(x, t, params, grads, loss, d_x=dx, y=y, overwrite=True))
    fw = timed(lambda: ops.tiny_mlp_forward(x, params, y=y))
    return tr, fw, [g.clone() for wb in grads for g in wb], dx.clone(), y.clone()
def rel(u, v): return float((u - v).abs().max() / v.abs().max().clamp_min(1e-30))
a = run(0)
print("H %d" % H, "n %d  f32 MFMA kernel: train %.4f ms  forward %.4f ms" % (n, a[0], a[1]), flush=True)
for mode in ((1, 2) if H == 128 else (1,)):
    b = run(mode)
    print("bf16x3 (%d waves): train %.4f ms  forward %.4f ms | vs f32 kernel: y %.2e  dx %.2e  grads %s" % (
        8 // mode, b[0], b[1], rel(b[4], a[4]), rel(b[3], a[3]), " ".join("%.1e" % rel(u, v) for u, v in zip(b[2], a[2]))), flush=True)
