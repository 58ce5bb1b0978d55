#!/usr/bin/env python3
"""Life of a scatter workgroup of the table gradient (csrc/hashgrid_bwd.hip): per-workgroup stamps of
a tools-only build (-DMRI_BWD_PROFILE), config-4 shape (L16 F2 T 2^19, B = 2^18).

    python tools/bwd_segments.py --build-only     # here (cross-compile), the .so travels
    python tools/bwd_segments.py                  # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libmri_bwdprof.so")


def build():
    import importlib
    b = importlib.import_module("mri_interpolation_amd.build")
    srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES]
    subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DMRI_BWD_PROFILE", "-shared", "-o", LIB] + srcs)


def main():
    if "--build-only" in sys.argv:
        return build()
    os.environ["MRI_LIB"] = LIB
    import torch
    from mri_interpolation_amd import _lib, encoding, ops
    n = 1 << 18
    enc = encoding.MultiResHashGrid(3, 16, 2, 19, 16, 16 * 1.4 ** 15).cuda()
    x = torch.rand(n, 3, device="cuda")
    d = torch.randn(16, 2, n, device="cuda") * 1e-3
    g = torch.zeros_like(enc.table.data)
    lib = _lib.load()
    chunks, rows = n // 512, 16
    prof = torch.zeros((65536 + 8192) * 8, dtype=torch.int64, device="cuda")
    run = lambda: ops.hashgrid_backward(enc.desc, x, d, g, feature_major=True, method=2, overwrite=True)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert lib.mri_debug_set_bwd_profile(C.c_void_p(prof.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
    full = prof.cpu()
    p = full[:rows * chunks * 8].reshape(rows, chunks, 8)
    live = p[:, :, 7] > 0
    q = p[live].double()
    print(f"{int(live.sum())} scatter workgroups stamped")
    names = ["issue loads", "run tables + coordinate landed (barrier)", "scan, clear (2 barriers)",
             "records staged (barrier)", "copy-out issued", "stores acknowledged"]
    seg = (q[:, 2:7] - q[:, 1:6])
    for i, nm in enumerate(names[1:]):
        print(f"  {nm:45s} {float(seg[:, i].mean()):9.0f} cycles  (median {float(seg[:, i].median()):.0f})")
    life = (q[:, 7] - q[:, 0]) * 10.0  # 100 MHz wall clock -> ns
    span = float(q[:, 7].max() - q[:, 0].min()) * 10.0
    print(f"  lifetime {float(life.mean()) / 1e3:.2f} us mean, kernel span {span / 1e3:.1f} us, "
          f"mean workgroups in flight {float(life.sum()) / span:.0f} (= {float(life.sum()) / span / 256:.2f} per CU)")
    t0 = float(q[:, 0].min())
    start, end = (q[:, 0] - t0) * 10.0 / 1e3, (q[:, 7] - t0) * 10.0 / 1e3  # us
    print("  in flight per CU over time (5 us buckets):",
          " ".join(f"{float(((start < t + 2.5) & (end > t + 2.5)).sum()) / 256:.1f}" for t in range(0, int(span / 1e3) + 1, 5)))
    per_level = (p[:, :, 7] - p[:, :, 0]).double() * 10.0 / 1e3
    print("  mean lifetime per grid row (us):", " ".join(f"{float(per_level[r][live[r]].mean()):.1f}" if live[r].any() else "-" for r in range(rows)))
    print("  first start per grid row (us):", " ".join(f"{(float(p[r, :, 0][live[r]].min()) - t0) * 0.01:.0f}" if live[r].any() else "-" for r in range(rows)))
    cyc = float((q[:, 6] - q[:, 1]).mean())
    print(f"  {cyc:.0f} shader cycles per lifetime -> {cyc / (float(life.mean()) * 1e-3):.0f} MHz")
    a = full[65536 * 8:(65536 + 4096) * 8].reshape(-1, 8)
    a = a[a[:, 7] > 0].double()
    print(f"{a.shape[0]} accumulate workgroups stamped (the dense ones of the merged launch are not)")
    for i, nm in enumerate(["zero the slice (barrier)", "records: loads + LDS atomics", "last wave (barrier)",
                            "convert + store issued", "stores acknowledged"]):
        d = a[:, i + 2] - a[:, i + 1]
        print(f"  {nm:45s} {float(d.mean()):9.0f} cycles  (median {float(d.median()):.0f})")
    la = (a[:, 7] - a[:, 0]) * 10.0
    sp = float(a[:, 7].max() - a[:, 0].min()) * 10.0
    print(f"  lifetime {float(la.mean()) / 1e3:.2f} us mean (max {float(la.max()) / 1e3:.1f}), span {sp / 1e3:.1f} us, "
          f"in flight {float(la.sum()) / sp / 256:.2f} per CU")
    dd = full[(65536 + 4096) * 8:].reshape(-1, 8)
    dd = dd[dd[:, 7] > 0].double()
    print(f"{dd.shape[0]} dense workgroups stamped")
    for i, nm in enumerate(["zero the slice (barrier)", "corners of the range: hash, LDS atomics", "last wave (barrier)",
                            "merge into the int64 area (global atomics) issued", "atomics acknowledged"]):
        d = dd[:, i + 2] - dd[:, i + 1]
        print(f"  {nm:52s} {float(d.mean()):9.0f} cycles  (median {float(d.median()):.0f})")
    ld = (dd[:, 7] - dd[:, 0]) * 10.0
    print(f"  lifetime {float(ld.mean()) / 1e3:.2f} us mean (max {float(ld.max()) / 1e3:.1f})")


if __name__ == "__main__":
    main()
