import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mri_interpolation_amd import _lib, models, ops, trainer
_lib.load()
n = 1 << 20
x = torch.rand(n, 3, device="cuda") * 2 - 1
y = torch.rand(n, 1, device="cuda")
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for L in (1, 2, 3, 5, 7):
    net = models.SirenNet(3, 256, 1, L).cuda()
    st = trainer.FusedStep(net, net.configure_optimizers())
    ws = st.forward(x, train=True)[1]
    print("L %d: infer %.3f  fwd %.3f  bwd(all) %.3f ms" % (L, timed(lambda: st.forward(x, train=False)),
          timed(lambda: st.forward(x, train=True)), timed(lambda: st.backward(x, y, ws))), flush=True)
