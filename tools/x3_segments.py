#!/usr/bin/env python3
"""Where the bf16x3 decoder kernel (csrc/mlp_x3.hip) spends its cycles: per-segment shader-clock
counters of a tools-only build (-DMRI_X3_PROFILE), config-4 shape.

    python tools/x3_segments.py --build-only     # here (cross-compile), the .so travels
    python tools/x3_segments.py [mode 1|2] [fwd]  # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("MRI_X3PROF_LIB") or os.path.join(ROOT, "tools", "libmri_x3prof.so")  # A/B: another build


def build():
    import importlib
    b = importlib.import_module("mri_interpolation_amd.build")
    srcs = [os.path.join(b.CSRC, s) for s in ("mlp_x3.hip", "mlp_fused.hip", "train_ops.hip")]
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]  # e.g. -DMRI_X3_SKEW with MRI_X3PROF_LIB=tools/other.so
    subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DMRI_X3_PROFILE"] + extra + ["-shared", "-o", LIB] + srcs)


def main():
    if "--build-only" in sys.argv:
        return build()
    import torch
    lib = C.CDLL(LIB)
    args_in = [a for a in sys.argv[1:] if not a.startswith("--")]
    mode = int(args_in[0]) if args_in else 1
    fwd = "fwd" in args_in
    lib.mri_set_option(b"mlp_x3", C.c_int32(mode))
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
    y = torch.empty(n, device=dev)
    lib.mri_tiny_mlp_workspace_bytes.restype = C.c_int64
    ws = torch.empty(lib.mri_tiny_mlp_workspace_bytes(k_in, h, C.c_int64(n)) // 4, device=dev)
    blocks = 256
    prof = torch.zeros(blocks * 8 * 32, dtype=torch.int64, device=dev)
    assert lib.mri_debug_set_x3_profile(C.c_void_p(prof.data_ptr())) == 0
    P = C.c_void_p
    if fwd:
        args = [P(x.data_ptr()), C.c_int64(n), C.c_int32(k_in), C.c_int32(h)]
        args += [P(p.data_ptr()) for p in (w1, b1, w2, b2, w3, b3)] + [P(y.data_ptr()), P(None)]
        fn = lib.mri_tiny_mlp_forward
    else:
        args = [P(x.data_ptr()), P(t.data_ptr()), C.c_int64(n), C.c_int32(k_in), C.c_int32(h)]
        args += [P(p.data_ptr()) for p in (w1, b1, w2, b2, w3, b3)]
        args += [C.c_float(1.0)] + [P(g.data_ptr()) for g in grads] + [P(dx.data_ptr()), P(loss.data_ptr()),
                                                                      P(None), P(ws.data_ptr()), C.c_int64(ws.numel() * 4), P(None)]
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
    waves = 8 // mode
    p = prof.cpu().reshape(blocks, 8, 32).double()[:, :waves]
    tiles = n / 32 / blocks
    names = ["S7 dx + next layer 1 (-> B0)", "(unused)", "S2 layer 2 (-> B2)", "S5 y, dz2, dW2 (-> B3)", "S6 dz1, dW1 (-> B4)"]
    q = p.mean(dim=(0, 1)) / tiles
    tot = float(q[:10].sum())
    print(f"{'forward' if fwd else 'train'} mode {mode} ({waves} waves): {ms * 1e3:.1f} us with counters; {tiles:.0f} tiles per workgroup; "
          f"{tot:.0f} cycles per tile ({tot * tiles / (ms * 1e-3) / 1e9:.2f} GHz if the loop were the kernel)")
    for i, nm in enumerate(names):
        print(f"  {nm:28s} work {float(q[2 * i]):7.0f}  wait {float(q[2 * i + 1]):6.0f}")
    print(f"  prologue {float(q[20]):.0f}  tail {float(q[18]):.0f}  epilogue {float(q[21]):.0f} cycles")
    pw = p[:, :, :10].sum(dim=2).mean(dim=0) / tiles
    print("  per wave total:", " ".join(f"{float(v):.0f}" for v in pw))
    work = p[:, :, 0:10:2].mean(dim=0) / tiles
    wait = p[:, :, 1:10:2].mean(dim=0) / tiles
    for i, nm in ((2, "S2"), (3, "S5"), (4, "S6")):
        print(f"  {nm} per wave: work", " ".join(f"{float(v):.0f}" for v in work[:, i]), "| wait",
              " ".join(f"{float(v):.0f}" for v in wait[:, i]))


if __name__ == "__main__":
    main()
