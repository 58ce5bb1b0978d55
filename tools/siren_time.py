"""Timings of the fused SIREN step (config 3 shape) on the GPU box: python tools/siren_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import _lib, models, ops, trainer
lib = _lib.load()
if os.environ.get('MRI_SIREN_ROWS') == '0':
    lib.mri_set_option(b'siren_rows', 0)
hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = models.SirenNet(3, hidden, 1, 5).cuda()
st = trainer.FusedStep(net, net.configure_optimizers())
n = 1 << 20
x = torch.rand(n, 3, device="cuda") * 2 - 1
y = torch.rand(n, 1, device="cuda")
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for chain in (True, False):
    st.use_chain = chain and st.chain is not None
    pred, ws = st.forward(x, train=True)
    print("hidden %d %s: infer %.3f  fwd %.3f  bwd %.3f  step %.3f ms" % (hidden, "chain     " if st.use_chain else "layer-wise",
        timed(lambda: st.forward(x, train=False)), timed(lambda: st.forward(x, train=True)),
        timed(lambda: st.backward(x, y, ws)), timed(lambda: st.train_step(x, y))), flush=True)
    if st.use_chain and st.chain_loss:
        st.chain_loss = False
        print("   step with separate loss / head kernels %.3f ms" % timed(lambda: st.train_step(x, y)), flush=True)
        st.chain_loss = True
