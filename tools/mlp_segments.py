#!/usr/bin/env python3
"""Where the fused tiny-MLP kernel spends its cycles: per-segment shader-clock counters.

Builds (or reuses) tools/libmri_prof.so = csrc/mlp_fused.hip compiled with -DMRI_MLP_PROFILE
(the shipped library has no counters), runs mri_tiny_mlp_train on a BASELINE config 4 batch and
prints, per team, the mean cycles per tile of every barrier-separated segment: `work` is mark ->
barrier entry, `wait` is the time spent inside the barrier.

    python tools/mlp_segments.py --build-only     # here (cross-compile), the .so travels
    python tools/mlp_segments.py                  # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("MRI_PROF_LIB") or os.path.join(ROOT, "tools", "libmri_prof.so")


def build():
    import importlib
    b = importlib.import_module("mri_interpolation_amd.build")
    srcs = [os.path.join(b.CSRC, s) for s in ("mlp_fused.hip", "train_ops.hip")]
    cmd = [b._hipcc()] + b.FLAGS + ["-DMRI_MLP_PROFILE", "-shared", "-o", LIB] + srcs
    subprocess.check_call(cmd)


def main():
    if not os.path.exists(LIB) or "--build-only" in sys.argv:
        build()
        if "--build-only" in sys.argv:
            return
    import torch
    lib = C.CDLL(LIB)
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
    assert lib.mri_debug_set_mlp_profile(C.c_void_p(prof.data_ptr())) == 0
    P = C.c_void_p
    args = [P(x.data_ptr()), P(t.data_ptr()), C.c_int64(n), C.c_int32(k_in), C.c_int32(h)]
    args += [P(p.data_ptr()) for p in (w1, b1, w2, b2, w3, b3)]
    args += [C.c_float(1.0)] + [P(g.data_ptr()) for g in (grads[0], grads[1], grads[2], grads[3])]
    args += [P(grads[4].data_ptr()), P(grads[5].data_ptr()), P(dx.data_ptr()), P(loss.data_ptr()),
             P(None), P(ws.data_ptr()), C.c_int64(ws.numel() * 4), P(None)]
    lib.mri_last_error.restype = C.c_char_p
    for _ in range(3):
        rc = lib.mri_tiny_mlp_train(*args)
        assert rc == 0, lib.mri_last_error()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    lib.mri_tiny_mlp_train(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    p = prof.cpu().reshape(blocks, 8, 32).double()
    tiles = n / 32 / (2 * blocks)  # per team
    names = ["S0 stage x", "S1 layer1", "S2 layer2", "S3 out dot", "S4 loss", "S5 dz2", "S6 dW2+dz1",
             "S7 relu'", "S8 dW1+dx"]
    print(f"kernel {ms * 1e3:.1f} us (with counters); {tiles:.0f} tiles per team; cycles per tile")
    # slot 2i = work of the segment that ENDS at barrier Si (i.e. segment S(i-1)), 2i+1 = wait
    for team in (0, 1):
        q = p[:, team * 4:(team + 1) * 4].mean(dim=(0, 1)) / tiles
        tot = float(q[:19].sum())
        print(f"team {team}: total {tot:8.0f} cycles / tile  ({tot * tiles / (ms * 1e-3) / 1e9:.2f} GHz)")
        for i in range(9):
            seg = names[(i - 1) % 9]
            print(f"   {seg:12s} work {float(q[2 * i]):7.0f}   wait at S{i} {float(q[2 * i + 1]):7.0f}")
        print(f"   {'tail':12s} work {float(q[18]):7.0f}")
        pro, epi = float(q[20] * tiles), float(q[21] * tiles)
        print(f"   once per launch: prologue (weights -> LDS, first x tile) {pro:8.0f} cycles, "
              f"epilogue (team merge, slab) {epi:8.0f} cycles; loop {tot * tiles:9.0f}")


if __name__ == "__main__":
    main()
