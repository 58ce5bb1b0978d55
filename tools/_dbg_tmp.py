import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import mri_interpolation_amd as amd
from conftest import load_golden
from test_gpu_parity import build_encoder, cuda
for name in ("enc_cfg2", "enc_cfg5_4d"):
    fx = load_golden(name)
    enc = build_encoder(amd, fx)
    x, d_out = cuda(fx["x"]), cuda(fx["d_out"])
    print(name, "n", x.shape, "d_out absmax per level", d_out.abs().reshape(x.shape[0], enc.n_levels, -1).amax(dim=(0, 2)).tolist()[:8])
    for fm in (False, True):
        d_table = torch.zeros_like(enc.table.data)
        g = d_out.t().contiguous() if fm else d_out
        amd.ops.hashgrid_backward(enc.desc, x, g, d_table, feature_major=fm, method=2)
        ref = torch.zeros_like(enc.table.data)
        amd.ops.hashgrid_backward(enc.desc, x, g, ref, feature_major=fm, method=1)
        for l in range(enc.n_levels):
            lo, hi = enc._row_span(l)
            a, b = ref[lo:hi], d_table[lo:hi]
            print(" fm", fm, "level", l, "rows", hi - lo, "max", float(a.abs().max()), "rel diff", float((a - b).abs().max() / a.abs().max()))
fx = load_golden("enc_cfg2")
enc = build_encoder(amd, fx)
x, d_out = cuda(fx["x"]), cuda(fx["d_out"])
a = torch.zeros_like(enc.table.data); b = torch.zeros_like(enc.table.data)
amd.ops.hashgrid_backward(enc.desc, x, d_out, a, method=1)
amd.ops.hashgrid_backward(enc.desc, x, d_out, b, method=2)
lo, hi = enc._row_span(6)
a6, b6 = a[lo:hi], b[lo:hi]
diff = (a6 - b6).abs().sum(1)
idx = diff.argsort(descending=True)[:12]
print("n", x.shape[0])
for i in idx.tolist():
    print("slot", i, "slice", i >> 13, "in-slice", i & 8191, "ref", a6[i].tolist(), "got", b6[i].tolist())
print("nonzero ref", int((a6.abs().sum(1) > 0).sum()), "nonzero got", int((b6.abs().sum(1) > 0).sum()), "bad", int((diff > 1e-5).sum()))
