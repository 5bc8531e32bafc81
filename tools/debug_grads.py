"""GPU debug: activation-gradient comparison (HIP tape vs oracle fp64 / fp32) stage by stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import eb4, losses, param_fill
from tests import oracle_util as ou
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seeds = (2, 3) if n == 4 else (21, 22)
lam = ou.SMOOTH_LAMBDAS
dev = torch.device("cuda:0")
x = param_fill.make_input(n, 256, seeds[0]); tgt = param_fill.make_labels(n); rng = ou.make_rng(n, seeds[1], 0.5)


def run_oracle(dtype):
    sd = ou.oracle_state(0.0, 0.3, dtype=dtype, requires_grad=True)
    out = eb4.forward_eb4(sd, x.to(dtype), training=True, drop_rate=0.5, rng=rng)
    for t in out["_feats"].values():
        t.retain_grad()
    ls = losses.pass1_loss(out, tgt, n // 2, n // 2, lam)
    ls["total_loss"].backward()
    return out, sd


o64, sd64 = run_oracle(torch.float64)
o32, sd32 = run_oracle(torch.float32)
m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
param_fill.fill_module_(m, 0.0, 0.3)
m = m.to(dev).train()
m._debug_watch = True
out = m(x.to(dev), rng=rng)
ld, t = out["loss_dict"], tgt.to(dev)
trip = sum(LOSSES["aw_triplet"](f, t) for f in ld["triplet"])
total = LOSSES["cross_entropy"](out["cls_out"], t) + lam["lambda_mask"] * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) \
    + lam["lambda_triplet"] * trip
total.backward()
cap = m._debug_tape.captured


def rel(a, b):
    return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300)).item()


print("activation gradients: name, HIP-vs-fp64, cpu-fp32-vs-fp64")
for k in ("pooled", "x_b6", "att_out", "x_b5", "x_b4", "dec2", "dec1", "x_b3", "x_b2", "x_b1", "x_b0"):
    g = cap.get(k)
    r64 = o64["_feats"][k].grad
    if g is None or r64 is None:
        print(k, "missing", g is None, r64 is None); continue
    if g.dim() == 4 and k not in ("dec3",):
        g = g.permute(0, 3, 1, 2)
    print("  %-8s %.3e   %.3e" % (k, rel(g, r64), rel(o32["_feats"][k].grad, r64)))
# forward activations too
for k in ("pooled", "x_b6", "att_out", "x_b5", "x_b4"):
    pass
