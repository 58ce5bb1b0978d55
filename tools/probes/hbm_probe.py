"""Streaming bandwidth of plain torch ops on this chip (a yardstick for the HBM-bound kernels): python tools/probes/hbm_probe.py"""
import torch
x = torch.empty(1 << 29, device="cuda").normal_()  # 2 GiB
y = torch.empty_like(x)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
gb = x.numel() * 4 / 1e9
ms = t(lambda: x.sum()); print(f"sum   (read  {gb:.2f} GB): {ms:.3f} ms  {gb / ms:.2f} TB/s")
ms = t(lambda: y.copy_(x)); print(f"copy  (r+w {2 * gb:.2f} GB): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
ms = t(lambda: y.fill_(1.0)); print(f"fill  (write {gb:.2f} GB): {ms:.3f} ms  {gb / ms:.2f} TB/s")
ms = t(lambda: torch.add(x, y, out=y)); print(f"add   (2r+w {3 * gb:.2f} GB): {ms:.3f} ms  {3 * gb / ms:.2f} TB/s")
