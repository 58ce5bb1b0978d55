import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import _lib, models, ops, trainer
_lib.load()
net = models.HashMLP(3, 16, 2, 19, 16, 16 * 1.4 ** 15, dim_hidden=128, n_layers=3, activation=torch.nn.ReLU,
                     batch_norm=False, final_activation=False).cuda()
enc = net.encoder
n = 1 << 18
x = torch.rand(n, 3, device="cuda")
out = torch.empty(32, n, device="cuda")
rows = ops.tiny_mlp_round_rows(32, 128, n)
ready = torch.zeros(-(-n // rows), dtype=torch.int64, device="cuda")
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("rows", rows, "blocks/slice", ops.hashgrid_signal_blocks(enc.desc, rows))
print("plain  fwd %.4f ms" % timed(lambda: ops.hashgrid_forward(enc.desc, x, enc.table.data, out=out, feature_major=True)))
ref = out.clone()
print("signal fwd %.4f ms" % timed(lambda: ops.hashgrid_forward_signal(enc.desc, x, enc.table.data, out, rows, ready)))
print("equal", torch.equal(ref, out), "ready", ready[:3].tolist())
