"""Build the gfx950 hot-path library `libmri_inr.so` in-tree with hipcc.

    python -m mri_interpolation_amd.build [--force]

The library is plain HIP behind the C ABI of include/mri_inr.h (no torch types), so it
is compiled with hipcc directly -- no torch extension machinery, and the .so travels to
the GPU box with the source tree.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libmri_inr.so")
SOURCES = ["hashgrid.hip", "hashgrid_bwd.hip", "linear.hip", "linear_small.hip", "mlp_fused.hip", "mlp_x3.hip", "train_ops.hip",
           "frequency.hip", "siren_chain.hip", "siren_rows.hip", "fused_step.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
         # the reference multiplies and adds separately (encoding.py:111-128, torch Adam);
         # keep those roundings instead of contracting to fma
         "-ffp-contract=off",
         # hardware float atomics (global_atomic_add_f32 / ds_add_f32), never a CAS loop
         "-munsafe-fp-atomics",
         "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include")]
# per-source extras.  siren_chain.hip: hipcc's SLP vectoriser turns the operand split's subtractions
# into v_pk_add_f32, the slower form beside bf16 MFMAs (MI355X_MICROARCH.md, packed f32 VALU):
# config-3 step 13.25 -> 12.58 ms without it (the explicitly packed sincos stays: 12.72 unpacked).
# The decoder kernels measured the same either way.
EXTRA_FLAGS = {"siren_chain.hip": ["-fno-slp-vectorize"], "siren_rows.hip": ["-fno-slp-vectorize",
                                                                                     # the order fixed before register allocation (sched_group_barrier)
                                                                                     # is the one wanted: the post-RA scheduler re-solves the groups with
                                                                                     # the allocator's copies in them and undoes the interleave
                                                                                     "-mllvm", "-enable-post-misched=false"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X hot path cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libmri_inr.so; returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "hashgrid_common.h", "device_math.h", "bf16x3.h",
                                               "mlp_fused.h")] + [os.path.join(ROOT, "include", "mri_inr.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE,
                                                stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    if force or procs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
