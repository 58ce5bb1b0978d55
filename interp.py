#!/usr/bin/env python3
"""Linear-interpolation baseline along time (keeps the reference entry point `interp.py`).

The reference (interp.py:24-52) takes slice [:, :, 3, :] of the dynamic volume, keeps the even
frames, builds an `itk.LinearInterpolateImageFunction` over them and evaluates it at
(t/2, y, x) for every voxel of the full grid in a Python loop, then saves
`itk_interpolated.nii.gz`.  Evaluating a linear interpolator of the even frames at t/2 is
plain linear interpolation along t (ITK clamps nothing here: t/2 <= (T-1)/2 is inside the
buffer when T is odd; for the last odd frame of an even T the nearest frame is used).
This is NumPy only (ITK is not installed) and is not on the hot path: it is the non-neural
baseline next to which the network's held-out-frame PSNR is reported.

    python interp.py --image_path sample_ankle_dyn_mri.nii.gz [--z 3] [--out itk_interpolated.nii.gz]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def interpolate_even_frames(data: np.ndarray) -> np.ndarray:
    """data (..., T): rebuild every frame from the even ones by linear interpolation in t."""
    values = data[..., ::2].astype(np.float64)
    n_even = values.shape[-1]
    t = np.arange(data.shape[-1]) / 2.0          # continuous index into the even frames
    lo = np.clip(np.floor(t).astype(int), 0, n_even - 1)
    hi = np.clip(lo + 1, 0, n_even - 1)
    w = t - lo
    w[hi == lo] = 0.0
    return values[..., lo] * (1.0 - w) + values[..., hi] * w


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    p.add_argument("--image_path", required=True)
    p.add_argument("--z", type=int, default=3, help="slice of the third axis (reference: 3)")
    p.add_argument("--out", default="itk_interpolated.nii.gz")
    args = p.parse_args(argv)
    from mri_interpolation_amd import nifti
    data = nifti.load(args.image_path)
    data = data / data.max()                      # reference interp.py:26
    if data.ndim == 4:
        data = data[:, :, args.z, :]              # reference interp.py:27
    out = interpolate_even_frames(data)
    nifti.save(out.astype(np.float64), args.out)  # reference writes float64 (np.zeros default)
    odd = out[..., 1::2], data[..., 1::2]
    print(f"saved {args.out}; PSNR on the {odd[0].shape[-1]} held-out odd frames: "
          f"{psnr(*odd):.2f} dB")


if __name__ == "__main__":
    main()
