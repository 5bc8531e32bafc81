"""GPU debug: the attention module's backward in situ (HIP) vs the oracle's attention run in fp64 on the
HIP-side inputs and upstream gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import eb4, param_fill
from tests import oracle_util as ou
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seeds = (2, 3) if n == 4 else (21, 22)
lam = ou.SMOOTH_LAMBDAS
dev = torch.device("cuda:0")
x = param_fill.make_input(n, 256, seeds[0]); tgt = param_fill.make_labels(n); rng = ou.make_rng(n, seeds[1], 0.5)
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
tape = m._debug_tape
cap = tape.captured
feats = {k: v for k, v in tape.watch.items()}
# forward activations from the HIP run (kept alive by the autograd node's ctx until backward cleared them:
# re-run the forward without tape to get them)
m._debug_watch = False
with torch.no_grad():
    m.train()
    # same masks => same activations (BN running stats do not affect train-mode outputs)
    r = m._run(x.to(dev), None, rng)
f = r["_feats"]
emb = f["x_b5"].permute(0, 3, 1, 2).double().cpu().requires_grad_()
dec3 = f["dec3"].double().cpu()
sd = ou.oracle_state(0.0, 0.3, dtype=torch.float64, requires_grad=True)
att = eb4.attention(dec3, x.double(), emb, sd, True, "ortho", rng["emb_keep"], 0.5)
g_att = cap["att_out"].permute(0, 3, 1, 2).double().cpu()
loss = (att["out"] * g_att).sum() + lam["lambda_mask"] * (att["freq_mask"].mean() + att["spat_mask"].mean())
loss.backward()


def rel(a, b):
    return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300)).item()


print("fwd att_out", rel(f["att_out"].permute(0, 3, 1, 2), att["out"].detach()))
print("fwd freq_mask", rel(r["freq_mask"].permute(0, 3, 1, 2), att["freq_mask"].detach()))
print("fwd spat_mask", rel(r["spat_mask"].permute(0, 3, 1, 2), att["spat_mask"].detach()))
print("d x_b5 (attention bwd in situ)", rel(cap["x_b5"].permute(0, 3, 1, 2), emb.grad))
params = dict(m.named_parameters())
for k in ("freq_filter.layer1.0.weight", "freq_filter.layer1.1.weight", "freq_filter.layer1.1.bias",
          "freq_filter.layer2.0.weight", "spat_filter.layer1.0.weight", "spat_filter.layer1.1.weight",
          "spat_filter.layer1.1.bias", "spat_filter.layer2.0.weight", "fuse_coef"):
    print("d", k, rel(params[k].grad, sd[k].grad))
# argmax agreement of the two filters
with torch.no_grad():
    pf = eb4.rfft2_cat(emb.detach(), "ortho")
    proj = torch.nn.functional.conv2d(pf, sd["freq_filter.layer1.0.weight"])
    proj = eb4.swish(eb4.batch_norm(proj, sd, "freq_filter.layer1.1", True, 1e-5))
    top2 = proj.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1e-30)
    print("freq proj top-2 relative gap: min %.3e" % gap.min().item(), "n<1e-5:", int((gap < 1e-5).sum()))
    proj = torch.nn.functional.conv2d(emb.detach(), sd["spat_filter.layer1.0.weight"], None, 1, 1)
    proj = eb4.swish(eb4.batch_norm(proj, sd, "spat_filter.layer1.1", True, 1e-5))
    top2 = proj.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1e-30)
    print("spat proj top-2 relative gap: min %.3e" % gap.min().item(), "n<1e-5:", int((gap < 1e-5).sum()))
