"""float64 yardstick for the places where two correct float32 evaluations differ by more than the
1e-5 parity bar (sums of 2^18 ... 2^20 signed terms; Adam's lr g / (|g| + eps) on a near-zero g).

Instead of widening the tolerance, the HIP path and the float32 oracle (or the reference's own fixture
values) are BOTH measured against the same op sequence evaluated in float64 on the same float32
inputs: the claim under test is "the kernel is no worse an f32 evaluation than the reference's".
"""
import numpy as np

from conftest import rel_err

# a tensor on which the f32 reference happens to sit within a few ulps of the float64 result must not
# fail the kernel for being a normal f32 evaluation: errors below this are "equal"
FLOOR = 2.0 ** -22


def assert_no_worse(kernel, f32_ref, f64, what="", factor=2.0, floor=FLOOR):
    """err(kernel, f64) <= factor * err(f32_ref, f64) + floor, as max-abs / max-abs AND relative L2."""
    k_max, k_l2 = rel_err(np.asarray(kernel, dtype=np.float64), np.asarray(f64, dtype=np.float64))
    r_max, r_l2 = rel_err(np.asarray(f32_ref, dtype=np.float64), np.asarray(f64, dtype=np.float64))
    assert k_max <= factor * r_max + floor and k_l2 <= factor * r_l2 + floor, \
        (f"{what}: kernel is {k_max:.3e} (max) / {k_l2:.3e} (L2) from float64, the f32 reference "
         f"{r_max:.3e} / {r_l2:.3e}")
    return k_max, r_max
