#!/usr/bin/env python3
"""Build the library with extra compiler flags into tools/libmri_variant.so for A/B runs:

    python tools/build_variant.py -DMRI_FWD_NT          # here
    MRI_LIB=tools/libmri_variant.so python bench.py     # on the GPU box

Every source is compiled with the flags the shipped build gives it (mri_interpolation_amd/build.py: FLAGS and the
per-source EXTRA_FLAGS, e.g. -fno-slp-vectorize for siren_chain.hip -- until round 4 this tool dropped those and
handicapped its own baselines by 5 % on config 3) plus the arguments given here.
"""
import importlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("mri_interpolation_amd.build")
args = sys.argv[1:]
name = "libmri_variant.so"
if args and args[0].startswith("--name="):
    name = args.pop(0)[len("--name="):]
out = os.path.join(ROOT, "tools", name)
with tempfile.TemporaryDirectory() as tmp:
    procs, objs = [], []
    for src in b.SOURCES:
        obj = os.path.join(tmp, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [b._hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get(src, []) + args + ["-c", os.path.join(b.CSRC, src), "-o", obj]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
