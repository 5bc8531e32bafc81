"""GPU debug: UDR18 activation-gradient comparison (HIP tape vs oracle fp64 / fp32) stage by stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import losses, param_fill, r18
from tests import oracle_util as ou
from tests.test_r18 import make_rng_r18, r18_state
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model

n, seeds = 8, (42, 142)
lam = ou.SMOOTH_LAMBDAS
dev = torch.device("cuda:0")
x = param_fill.make_input(n, 128, seeds[0]); tgt = param_fill.make_labels(n); rng = make_rng_r18(n, seeds[1])


def run_oracle(dtype):
    sd = r18_state(dtype, requires_grad=True)
    out = r18.forward_r18(sd, x.to(dtype), training=True, drop_rate=0.5, rng=rng)
    for t in out["_feats"].values():
        if t.requires_grad:
            t.retain_grad()
    losses.pass1_loss(out, tgt, n // 2, n // 2, lam)["total_loss"].backward()
    return out, sd


o64, sd64 = run_oracle(torch.float64)
o32, sd32 = run_oracle(torch.float32)
m = load_model("UDR18")(num_classes=2, drop_rate=0.5)
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
for k in ("att_out", "emb", "ext", "dec1"):
    g = cap.get(k)
    r64 = o64["_feats"][k].grad
    if g is None or r64 is None:
        print(k, "missing", g is None, r64 is None); continue
    print("  %-8s %.3e   %.3e" % (k, rel(g.permute(0, 3, 1, 2), r64), rel(o32["_feats"][k].grad, r64)))
params = dict(m.named_parameters())
for k in ("emb_block1.downsample.0.weight", "emb_block1.downsample.1.weight", "emb_block1.downsample.1.bias",
          "emb_block1.conv1.weight", "emb_block1.norm1.weight", "emb_block1.conv2.weight", "emb_block1.conv2.freq_conv.weight",
          "emb_block1.norm2.weight", "emb_block2.conv1.weight", "dec_block1.0.weight", "extractor.layer3.1.conv2.weight"):
    print("  d %-40s %.3e   %.3e" % (k, rel(params[k].grad, sd64[k].grad), rel(sd32[k].grad, sd64[k].grad)))
