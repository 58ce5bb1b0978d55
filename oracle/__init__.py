"""CPU oracle for the coordinate-MLP hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU fp32 + NumPy uint32) of the
reference algorithm for the hot path named in BASELINE.json: multiresolution
hash-grid lookup (reference `encoding.py`), SIREN / ReLU tiny-MLP forward and
backward, MSE loss and Adam (reference `models.py`), and the coordinate-batch
producer (reference `datamodules.py`).  Every function cites the reference
file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / the reported CPU baseline.
Nothing under `mri_interpolation_amd/` imports it: the product path runs the
HIP kernels in `mri_interpolation_amd/csrc` and fails loudly without them.

Parity pin: the reference has no tests.  The oracle is
pinned against golden vectors produced in the build container by importing the
reference's own `encoding.py` / `models.py` (see `tests/golden/make_golden.py`
and `tests/test_oracle_golden.py`).
"""

from . import detrand, hashgrid, mlp, data, train  # noqa: F401
