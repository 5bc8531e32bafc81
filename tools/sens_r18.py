"""CPU: sensitivity of the UDR18 parameter gradients to 1e-7-relative perturbations of the input (float64 oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import losses, param_fill, r18
from tests import oracle_util as ou
from tests.test_r18 import make_rng_r18, r18_state
n = 8
x = param_fill.make_input(n, 128, 42).double(); tgt = param_fill.make_labels(n); rng = make_rng_r18(n, 142)
lam = ou.SMOOTH_LAMBDAS
def grads(xx, pin=None):
    sd = r18_state(torch.float64, requires_grad=True)
    r = dict(rng) if pin is None else dict(rng, pool_sel=pin)
    out = r18.forward_r18(sd, xx, training=True, drop_rate=0.5, rng=r)
    losses.pass1_loss(out, tgt, n // 2, n // 2, lam)["total_loss"].backward()
    return {k: v.grad for k, v in sd.items() if v.grad is not None}
g0 = grads(x)
torch.manual_seed(1)
g1 = grads(x * (1 + 1e-7 * torch.randn_like(x)))
rows = sorted(((( g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-300)).item(), k) for k in g0)
print("relative gradient change under a 1e-7 relative input perturbation (float64 arithmetic):")
for r in rows[-12:]:
    print("  %.3e  %s" % r)
print("median %.3e" % rows[len(rows) // 2][0])
