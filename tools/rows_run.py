#!/usr/bin/env python3
"""A few forward passes of the config-3 SIREN for counter runs: python3 tools/rows_run.py [train] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mri_interpolation_amd import _lib, models, trainer

_lib.load()
train = "train" in sys.argv
reps = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
net = models.SirenNet(3, 256, 1, 5).cuda()
st = trainer.FusedStep(net, net.configure_optimizers())
st.chain_loss = False
x = torch.rand(1 << 20, 3, device="cuda") * 2 - 1
for _ in range(reps):
    st.forward(x, train=train)
torch.cuda.synchronize()
print("done")
