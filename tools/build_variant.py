#!/usr/bin/env python3
"""Build the library with extra compiler flags into tools/libmri_variant.so for A/B runs:

    python tools/build_variant.py -DMRI_FWD_NT          # here
    MRI_LIB=tools/libmri_variant.so python bench.py     # on the GPU box
"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("mri_interpolation_amd.build")
args = sys.argv[1:]
name = "libmri_variant.so"
if args and args[0].startswith("--name="):
    name = args.pop(0)[len("--name="):]
out = os.path.join(ROOT, "tools", name)
srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES]
subprocess.check_call([b._hipcc()] + b.FLAGS + args + ["-shared", "-o", out] + srcs)
print(out)
