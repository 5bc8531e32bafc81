"""CPU: two-pass train step restated with the oracle + torch AdamW, against the reference-engine golden."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import eb4, losses, param_fill
from tests import oracle_util as ou

g = np.load("tests/golden/udeb4_step_n4.npz")
n, size, in_seed, mask_seed, num_steps = [int(v) for v in g["meta"]]
tag, cur_step = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("early", 1)
x = param_fill.make_input(n, size, in_seed); tgt = param_fill.make_labels(n)
rngs = [ou.make_rng(n, mask_seed, 0.5), ou.make_rng(n, mask_seed + 1, 0.5)]
sd = ou.oracle_state(0.0, 0.3, requires_grad=True)
named = [(k, v) for k, v in sd.items() if v.requires_grad]
nd = [p for k, p in named if p.ndim <= 1 or k.endswith(".bias")]
dd = [p for k, p in named if not (p.ndim <= 1 or k.endswith(".bias"))]
opt = torch.optim.AdamW([{"params": nd, "weight_decay": 0.0}, {"params": dd, "weight_decay": 5e-6}], lr=1e-4, betas=(0.9, 0.999), amsgrad=True)
lam = ou.LAMBDAS
out = eb4.forward_eb4(sd, x, training=True, drop_rate=0.5, rng=rngs[0])
l1 = losses.pass1_loss(out, tgt, n // 2, n // 2, lam)
for k in ("total_loss", "cls_loss", "triplet_loss"):
    print("pass1", k, float(l1[k]), float(g[f"{tag}_loss_{k}"]))
fm_gt = out["loss_dict"]["freq_mask"].detach(); sm_gt = out["loss_dict"]["spat_mask"].detach()
fac_gt = out["loss_dict"]["factorization"].detach()
l1["total_loss"].backward(); opt.step()
xp = F.interpolate(F.interpolate(x, scale_factor=0.75, mode="nearest"), size=x.shape[-2:], mode="nearest")
out2 = eb4.forward_eb4(sd, x, training=True, drop_rate=0.5, rng=dict(rngs[1], noise_x=xp))
l2 = losses.pass2_loss(out2, tgt, n // 2, n // 2, lam, fm_gt, sm_gt, fac_gt, cur_step > 0.1 * num_steps)
for k in ("freq_mask_loss", "spat_mask_loss", "fac_loss"):
    a, b = float(l2[k]), float(g[f"{tag}_loss_{k}"])
    print("pass2", k, a, b, "rel", abs(a - b) / abs(b))
