#!/usr/bin/env python3
"""Lookup + decoder step as two kernels against the one-kernel form (mri_hash_tiny_mlp_train), config-4 shape.

    python tools/enc_dec_time.py [hidden 128|64] [dim 3|4] [n] [log2 T]
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import encoding, ops

hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
log2t = int(sys.argv[4]) if len(sys.argv) > 4 else 19
enc = encoding.MultiResHashGrid(dim, 16, 2, log2t, 16, 16 * 1.4 ** 15 if dim == 3 else 512).cuda()
k_in = 32
x = torch.rand(n, dim, device="cuda")
t = torch.rand(n, device="cuda")
mk = lambda *s: torch.randn(*s, device="cuda")  # noqa: E731
params = [(mk(hidden, k_in) / k_in ** 0.5, mk(hidden) * 0.1), (mk(hidden, hidden) / hidden ** 0.5, mk(hidden) * 0.1),
          (mk(1, hidden) / hidden ** 0.5, mk(1) * 0.1)]
grads = [tuple(torch.zeros_like(p) for p in wb) for wb in params]
loss = torch.zeros(1, device="cuda")
feats = torch.empty(k_in, n, device="cuda")
d = torch.empty(k_in, n, device="cuda")


def timed(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def two():
    ops.hashgrid_forward(enc.desc, x, enc.table.data, out=feats, feature_major=True)
    ops.tiny_mlp_train(feats, t, params, grads, loss, d_x=d, overwrite=True)


def one():
    ops.hash_tiny_mlp_train(enc.desc, enc.table.data, x, t, params, grads, loss, d, overwrite=True)


t_l = timed(lambda: ops.hashgrid_forward(enc.desc, x, enc.table.data, out=feats, feature_major=True))
t_d = timed(lambda: ops.tiny_mlp_train(feats, t, params, grads, loss, d_x=d, overwrite=True))
print(f"hidden {hidden} dim {dim} n {n} T 2^{log2t}: lookup {t_l:.4f} + decoder {t_d:.4f} = two kernels {timed(two):.4f} ms | one kernel {timed(one):.4f} ms")
