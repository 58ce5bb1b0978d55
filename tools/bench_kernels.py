#!/usr/bin/env python3
"""Micro-benchmarks of individual hot-path kernels (HIP events), for tuning.

    python tools/bench_kernels.py hash      # per-level forward / backward timings
    python tools/bench_kernels.py gemm      # layer GEMMs at the BASELINE shapes
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mri_interpolation_amd import _lib, encoding, ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def hash_bench():
    n = 1 << 18
    x = torch.rand(n, 3, device="cuda")
    for label, kw in [("L1 res16 T4096", dict(n_levels=1, log2_hashmap_size=19, base_resolution=16, finest_resolution=512)),
                      ("L1 res128 T2^19", dict(n_levels=1, log2_hashmap_size=19, base_resolution=128, finest_resolution=512)),
                      ("L1 res2489 T2^19", dict(n_levels=1, log2_hashmap_size=19, base_resolution=2489, finest_resolution=4000)),
                      ("L16 cfg2", dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, finest_resolution=512)),
                      ("L16 cfg4", dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, finest_resolution=16 * 1.4 ** 15))]:
        enc = encoding.MultiResHashGrid(3, n_features_per_level=2, **kw).cuda()
        width = enc.output_dim
        out = torch.empty(width, n, device="cuda")
        d = torch.randn(width, n, device="cuda")
        g = torch.zeros_like(enc.table.data)
        t_f = timeit(lambda: ops.hashgrid_forward(enc.desc, x, enc.table.data, out=out, feature_major=True))
        t_a = timeit(lambda: ops.hashgrid_backward(enc.desc, x, d, g, feature_major=True, method=1))
        t_l = timeit(lambda: ops.hashgrid_backward(enc.desc, x, d, g, feature_major=True, method=2))
        print(f"{label:20s} sizes {enc.sizes[:3]}.. fwd {t_f:8.3f} ms  bwd atomic {t_a:8.3f} ms  bwd lds {t_l:8.3f} ms", flush=True)
    for aff in (0, 1):
        _lib.set_option("xcd_affinity", aff)
        enc = encoding.MultiResHashGrid(3, 16, 2, 19, 16, 16 * 1.4 ** 15).cuda()
        out = torch.empty(32, n, device="cuda")
        print("xcd_affinity", aff, "fwd cfg4", timeit(lambda: ops.hashgrid_forward(enc.desc, x, enc.table.data, out=out, feature_major=True)))


def gemm_bench():
    for m, n, k in [(1 << 18, 128, 32), (1 << 18, 128, 128), (1 << 18, 1, 128), (1 << 18, 64, 64),
                    (1 << 20, 256, 256), (1 << 20, 256, 3), (1 << 20, 1, 256)]:
        x = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") / k ** 0.5
        b = torch.randn(n, device="cuda")
        y = torch.empty(m, n, device="cuda")
        dy = torch.randn(m, n, device="cuda")
        dx = torch.empty(m, k, device="cuda")
        dw = torch.zeros(n, k, device="cuda")
        db = torch.zeros(n, device="cuda")
        t_f = timeit(lambda: ops.linear_forward(x, w, b, ops.ACT_RELU, out=y))
        t_d = timeit(lambda: ops.linear_backward_data(dy, w, ops.DERIV_RELU_MASK, x, dx=dx))
        t_w = timeit(lambda: ops.linear_backward_weight(dy, x, dw, db))
        fl = 2.0 * m * n * k / 1e9
        print(f"M={m} N={n} K={k}: fwd {t_f:.3f} ms ({fl / t_f:.1f} TF)  bwd_data {t_d:.3f} ms ({fl / t_d:.1f} TF)  "
              f"bwd_weight {t_w:.3f} ms ({fl / t_w:.1f} TF)", flush=True)




def mlp_bench():
    """Fixed vs per-coordinate cost of the fused tiny-MLP kernel."""
    for hidden, k_in in [(128, 32), (64, 32)]:
        params = [(torch.randn(hidden, k_in, device="cuda") * 0.1, torch.zeros(hidden, device="cuda")),
                  (torch.randn(hidden, hidden, device="cuda") * 0.1, torch.zeros(hidden, device="cuda")),
                  (torch.randn(1, hidden, device="cuda") * 0.1, torch.zeros(1, device="cuda"))]
        grads = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in params]
        loss = torch.zeros(1, device="cuda")
        for logn in (14, 16, 17, 18, 19, 20):
            n = 1 << logn
            x = torch.randn(k_in, n, device="cuda")
            t = torch.rand(n, 1, device="cuda")
            dx = torch.empty_like(x)
            ms = timeit(lambda: ops.tiny_mlp_train(x, t, params, grads, loss, d_x=dx), iters=20)
            fl = 6.0 * (k_in * hidden + hidden * hidden + hidden) * n
            print(f"H={hidden} n=2^{logn}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TF", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "hash"
    {"hash": hash_bench, "gemm": gemm_bench, "mlp": mlp_bench}[what]()
