"""CPU: sensitivity of the UDR18 gradients (float64 oracle) to fp32-rounding-sized noise injected at every conv /
linear / fft output (relative 6e-8, i.e. half an fp32 ulp), forward only (backward exact)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import losses, param_fill, r18
from tests import oracle_util as ou
from tests.test_r18 import make_rng_r18, r18_state
n = 8
x = param_fill.make_input(n, 128, 42).double(); tgt = param_fill.make_labels(n); rng = make_rng_r18(n, 142)
lam = ou.SMOOTH_LAMBDAS
def grads():
    sd = r18_state(torch.float64, requires_grad=True)
    out = r18.forward_r18(sd, x, training=True, drop_rate=0.5, rng=rng)
    losses.pass1_loss(out, tgt, n // 2, n // 2, lam)["total_loss"].backward()
    return {k: v.grad for k, v in sd.items() if v.grad is not None}
g0 = grads()
orig_conv, orig_lin = F.conv2d, F.linear
gen = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
EPS = float(sys.argv[2]) if len(sys.argv) > 2 else 6e-8
def noisy(fn):
    def f(*a, **k):
        y = fn(*a, **k)
        return y * (1.0 + EPS * torch.randn(y.shape, generator=gen, dtype=y.dtype))
    return f
F.conv2d, F.linear = noisy(orig_conv), noisy(orig_lin)
try:
    g1 = grads()
finally:
    F.conv2d, F.linear = orig_conv, orig_lin
rows = sorted((((g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-300)).item(), k) for k in g0)
print("relative gradient change with %.0e relative noise on every conv/linear output (float64 otherwise):" % EPS)
for r in rows[-14:]:
    print("  %.3e  %s" % r)
print("median %.3e" % rows[len(rows) // 2][0])
for k in ("extractor.layer2.0.downsample.0.weight", "extractor.layer2.0.conv2.weight", "extractor.layer3.1.conv1.freq_conv.weight", "extractor.layer3.1.bn1.bias"):
    print("  %-45s %.3e" % (k, dict((b, a) for a, b in rows)[k]))
