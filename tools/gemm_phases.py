#!/usr/bin/env python3
"""Where a 256 x 256 layer of the generic GEMM kernel spends its cycles (per-wave phase counters).

Builds tools/libmri_gprof.so = csrc/linear*.hip compiled with -DMRI_GEMM_PROFILE (the shipped
library has no counters) and runs one forward layer at M = 2^20 per activation.

    python tools/gemm_phases.py --build-only    # here
    python tools/gemm_phases.py                 # on the GPU box
"""
import ctypes as C
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libmri_gprof.so")


def build():
    b = importlib.import_module("mri_interpolation_amd.build")
    srcs = [os.path.join(b.CSRC, s) for s in ("linear.hip", "linear_small.hip", "train_ops.hip")]
    subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DMRI_GEMM_PROFILE", "-shared", "-o", LIB] + srcs)


def main():
    if not os.path.exists(LIB) or "--build-only" in sys.argv:
        build()
        if "--build-only" in sys.argv:
            return
    import torch
    lib = C.CDLL(LIB)
    m, n, k = 1 << 20, 256, 256
    x = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, device="cuda") / k ** 0.5
    b = torch.randn(n, device="cuda")
    y = torch.empty(m, n, device="cuda")
    d = torch.empty(m, n, device="cuda")
    blocks = (m // 128) * 2 + 64
    prof = torch.zeros(blocks * 4 * 8, dtype=torch.int64, device="cuda")
    assert lib.mri_debug_set_gemm_profile(C.c_void_p(prof.data_ptr())) == 0
    P = C.c_void_p
    names = ["prologue", "issue loads", "reads+MFMA", "wait+store", "barrier", "epilogue"]
    for act, label in ((1, "relu"), (2, "sine")):
        args = [P(x.data_ptr()), C.c_int64(k), C.c_int64(1), P(w.data_ptr()), P(b.data_ptr()),
                C.c_int64(m), C.c_int32(n), C.c_int32(k), C.c_int32(act), C.c_float(30.0),
                P(y.data_ptr()), C.c_int64(n), P(d.data_ptr() if act == 2 else None), C.c_int64(n), P(None)]
        for _ in range(2):
            assert lib.mri_linear_forward(*args) == 0
        torch.cuda.synchronize()
        prof.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.mri_linear_forward(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        p = prof.cpu().reshape(-1, 8).double()
        p = p[p.sum(1) > 0]
        mean = p.mean(0)
        tot = float(mean[:6].sum())
        print(f"{label}: {ms:.3f} ms, {2.0 * m * n * k / ms / 1e9:.1f} TF; {p.shape[0]} waves; "
              f"{tot:.0f} cycles per tile and wave (8 chunks: 32768 cycles of MFMA issue)")
        for i, nm in enumerate(names):
            print(f"   {nm:12s} {float(mean[i]):8.0f}  ({100 * float(mean[i]) / tot:4.1f} %)")
    # backward-data of the same layer: dx = (dy W) (.) g
    dy = torch.randn(m, n, device="cuda")
    dx = torch.empty(m, k, device="cuda")
    args = [P(dy.data_ptr()), C.c_int64(n), P(w.data_ptr()), C.c_int64(m), C.c_int32(n), C.c_int32(k),
            C.c_int32(1), P(d.data_ptr()), C.c_int64(n), P(dx.data_ptr()), C.c_int64(k), C.c_int64(1),
            P(None)]
    for _ in range(2):
        assert lib.mri_linear_backward_data(*args) == 0
    torch.cuda.synchronize()
    prof.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.mri_linear_backward_data(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    p = prof.cpu().reshape(-1, 8).double()
    p = p[p.sum(1) > 0]
    mean = p.mean(0)
    tot = float(mean[:6].sum())
    print(f"backward-data (x derivative): {ms:.3f} ms, {2.0 * m * n * k / ms / 1e9:.1f} TF; {tot:.0f} cycles")
    for i, nm in enumerate(names):
        print(f"   {nm:12s} {float(mean[i]):8.0f}  ({100 * float(mean[i]) / tot:4.1f} %)")
    # backward-weight: dW += dy^T x (batch contraction, split over workgroups)
    dw = torch.zeros(n, k, device="cuda")
    db = torch.zeros(n, device="cuda")
    args = [P(dy.data_ptr()), C.c_int64(n), P(x.data_ptr()), C.c_int64(k), C.c_int64(1), C.c_int64(m),
            C.c_int32(n), C.c_int32(k), P(dw.data_ptr()), P(db.data_ptr()), P(None)]
    for _ in range(2):
        assert lib.mri_linear_backward_weight(*args) == 0
    torch.cuda.synchronize()
    prof.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.mri_linear_backward_weight(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    p = prof.cpu().reshape(-1, 8).double()
    p = p[p.sum(1) > 0]
    mean = p.mean(0)
    tot = float(mean[:6].sum())
    print(f"backward-weight: {ms:.3f} ms, {2.0 * m * n * k / ms / 1e9:.1f} TF; {p.shape[0]} waves, {tot:.0f} cycles each")
    for i, nm in enumerate(names):
        print(f"   {nm:12s} {float(mean[i]):8.0f}  ({100 * float(mean[i]) / tot:4.1f} %)")


if __name__ == "__main__":
    main()
