import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.test_kernels_gpu import rnd, run_tape, rel_err, to_pix, to_nchw
from unidefense_amd import tape as T
dev = torch.device("cuda:0")
N, Ci, Co, H = 8, 448, 512, 16
x = rnd(N, Ci, H, H, seed=1); w = rnd(Co, Ci, 1, 1, seed=2, scale=0.05)
g_, b_ = rnd(Co, seed=3) * 0.1 + 1, rnd(Co, seed=4) * 0.1
xr, wr, gr, br = [t.double().requires_grad_() for t in (x, w, g_, b_)]
y = F.conv2d(xr, wr)
y = F.batch_norm(y, None, None, gr, br, True, 0.1, 1e-5)
yr = F.max_pool2d(y, 3, 2, 1)
gy = rnd(*yr.shape, seed=5)
yr.backward(gy.double())
rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
def fn(t, a, w_, g1, b1):
    h = T.conv1x1(t, a, w_)
    h = T.batchnorm_act(t, h, g1, b1, rm, rv, 1e-5, 0.1, True, 0)
    return T.maxpool3s2(t, h)
outs, gin, gp = run_tape(fn, [to_pix(x).to(dev)], [w.to(dev), g_.to(dev).requires_grad_(), b_.to(dev).requires_grad_()], lambda o: [to_pix(gy).to(dev)])
print("y", rel_err(to_nchw(outs[0]), yr), "dx", rel_err(to_nchw(gin[0]), xr.grad), "dw", rel_err(gp[0], wr.grad), "dg", rel_err(gp[1], gr.grad), "db", rel_err(gp[2], br.grad))
# same without maxpool
xr, wr, gr, br = [t.double().requires_grad_() for t in (x, w, g_, b_)]
yr = F.batch_norm(F.conv2d(xr, wr), None, None, gr, br, True, 0.1, 1e-5)
gy = rnd(*yr.shape, seed=6); yr.backward(gy.double())
outs, gin, gp = run_tape(lambda t, a, w_, g1, b1: T.batchnorm_act(t, T.conv1x1(t, a, w_), g1, b1, rm, rv, 1e-5, 0.1, True, 0),
                         [to_pix(x).to(dev)], [w.to(dev), g_.to(dev).requires_grad_(), b_.to(dev).requires_grad_()], lambda o: [to_pix(gy).to(dev)])
print("no-pool: y", rel_err(to_nchw(outs[0]), yr), "dx", rel_err(to_nchw(gin[0]), xr.grad), "dw", rel_err(gp[0], wr.grad), "dg", rel_err(gp[1], gr.grad))
# conv only
xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
yr = F.conv2d(xr, wr); gy = rnd(*yr.shape, seed=7); yr.backward(gy.double())
outs, gin, gp = run_tape(lambda t, a, w_: T.conv1x1(t, a, w_), [to_pix(x).to(dev)], [w.to(dev)], lambda o: [to_pix(gy).to(dev)])
print("conv only: y", rel_err(to_nchw(outs[0]), yr), "dx", rel_err(to_nchw(gin[0]), xr.grad), "dw", rel_err(gp[0], wr.grad))
