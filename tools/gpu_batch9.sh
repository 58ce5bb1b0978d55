#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "siren" > gpurun_out/r2/tests6.out 2>&1
echo tests rc=$?; tail -3 gpurun_out/r2/tests6.out
python - <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from mri_interpolation_amd import _lib, models, trainer
_lib.load()
net = models.SirenNet(3, 256, 1, 5).cuda()
import copy
nets = [net, copy.deepcopy(net), copy.deepcopy(net)]
sts = [trainer.FusedStep(m, m.configure_optimizers()) for m in nets]
sts[2].use_chain = False
x = torch.rand(1 << 18, 3, device="cuda") * 2 - 1
y = torch.rand(1 << 18, 1, device="cuda")
outs = []
for i, st in enumerate(sts):
    _lib.set_option("siren_two_per_cu", 1 if i == 1 else 0)
    p, ws = st.forward(x, train=True); st.backward(x, y, ws)
    outs.append((p.clone(), st.flat.grad.clone(), st.forward(x, train=False)[0].clone()))
for i in (0, 1):
    print("geometry", i, "vs layer-wise:", [float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs[i], outs[2])])
_lib.set_option("siren_two_per_cu", 0)
PY
python tools/siren_time.py
MRI_LIB=tools/libmri_sprof.so python tools/siren_phases.py
