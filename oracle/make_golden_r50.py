"""TEST INFRASTRUCTURE — goldens for UniDefenseModelRes50 recorded from the REFERENCE imported from /root/reference
(this container only).  BASELINE configs[3] names 320x320, whose 2^k*5 FFT sizes the HIP path does not cover
yet: recorded at 256x256, bs 4.

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_r50 [size]     (256 -> udr50_n4.npz, 320 -> udr50_n4_s320.npz:
BASELINE configs[3]'s resolution, feature maps 80/40/20/10 = the 5*2^k FFT sizes)
Writes tests/golden/udr50_n4*.npz: eval-mode outputs, train-mode outputs + pass-1 losses + all parameter gradient
norms/heads ('full' and 'smooth' loss variants, like make_golden.py) — once with the reference in float32 (keys as
in udr18_n8.npz) and once in FLOAT64 (keys prefixed 'f64_').  The float64 record is the tight pin: a 53-layer
ReLU network with batch-4 statistics amplifies fp32 rounding (and the ReLU / max-pool near-tie flips it causes) to
~2e-3 on some gradients even between two correct fp32 evaluations, while two float64 evaluations of the same
function agree to 1e-9.  Randomness is pinned by injecting seeded keep-masks at the reference's three F.dropout
sites (model/unidefense.py:586, :552, :604).
Seeds (46, 146): smallest relative top-2 gap in the dynamic filters' torch.max is 3.6e-4 (train) / 2.8e-4 (eval)
for this batch — with 2048 / 4096 channels most seeds have gaps of 1e-5 (scan: seeds 40..51).
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import ref_import, param_fill          # noqa: E402
from oracle.make_golden import LAMBDAS             # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N, SIZE, IN_SEED, MASK_SEED, DROP = 4, 256, 46, 146, 0.5


def make_rng_r50(n, seed, drop_rate=DROP, size=256):
    g = torch.Generator().manual_seed(seed)

    def bern(shape, keep):
        return (torch.rand(shape, generator=g) < keep).float()
    return {"dec_keep": bern((n, 1024, size // 16, size // 16), 0.8),
            "emb_keep": bern((n, 2048, size // 32, size // 32), 1.0 - drop_rate),
            "feat_keep": bern((n, 2048), 1.0 - drop_rate)}


def pack(out, store, prefix):
    ld = out["loss_dict"]
    store[prefix + "cls_out"] = out["cls_out"].detach().numpy()
    store[prefix + "rec_pool8"] = torch.nn.functional.adaptive_avg_pool2d(out["rec"].detach(), 8).numpy()
    store[prefix + "factorization"] = ld["factorization"].detach().numpy()[:, :64]
    for k in ("freq_mask", "spat_mask", "spatial", "freq"):
        store[prefix + k] = ld[k].detach().numpy()
    for i, t in enumerate(ld["triplet"]):
        store[prefix + f"triplet{i}"] = t.detach().numpy()


def record(m, ref_loss, x, tgt, rng, store, pre):
    m.eval()
    with torch.no_grad():
        pack(m(x), store, pre + "eval_")
    F = torch.nn.functional
    orig = F.dropout
    names = []
    for variant, lam in (("full", LAMBDAS), ("smooth", dict(LAMBDAS, lambda_recons=0.0, lambda_freq=0.0))):
        queue = [("dec_keep", 0.2), ("emb_keep", DROP), ("feat_keep", DROP)]

        def fake_dropout(inp, p=0.5, training=True, inplace=False):
            name, pp = queue.pop(0)
            assert abs(pp - p) < 1e-12 and training, (name, p, pp)
            scale = rng[name].to(inp.dtype) / (1.0 - p)
            return inp.mul_(scale) if inplace else inp * scale
        F.dropout = fake_dropout
        try:
            m.train()
            out = m(x)
        finally:
            F.dropout = orig
        assert not queue
        ld = out["loss_dict"]
        n_real = N // 2
        trip = sum(ref_loss.LOSSES["aw_triplet"](f, tgt) for f in ld["triplet"])
        cls = ref_loss.LOSSES["cross_entropy"](out["cls_out"], tgt)
        real_rec = ld["spatial"].narrow(0, 0, n_real).mean()
        real_freq = ld["freq"].narrow(0, 0, n_real).mean()
        total = cls + lam["lambda_mask"] * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) + \
            lam["lambda_triplet"] * trip + lam["lambda_recons"] * real_rec + lam["lambda_freq"] * real_freq
        m.zero_grad()
        total.backward()
        if variant == "full":
            pack(out, store, pre + "train_")
        for k, v in (("total_loss", total), ("cls_loss", cls), ("triplet_loss", trip), ("real_rec_loss", real_rec),
                     ("real_freq_loss", real_freq)):
            store[f"{pre}{variant}_loss_{k}"] = np.array(v.item())
        names, norms, heads = [], [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            norms.append(p.grad.double().norm().item())
            h = torch.zeros(8, dtype=p.dtype)
            f = p.grad.flatten()[:8]
            h[: f.numel()] = f
            heads.append(h.numpy())
        store["grad_names"] = np.array(names)
        store[f"{pre}{variant}_grad_norms"] = np.array(norms)
        store[f"{pre}{variant}_grad_heads"] = np.stack(heads)
    return len(names)


def main(size=SIZE):
    os.makedirs(OUT, exist_ok=True)
    ref_model, ref_loss = ref_import.import_reference()
    torch.manual_seed(0)
    m = ref_model.load_model("UDR50")(extractor="resnet50", num_classes=2, drop_rate=DROP)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    x = param_fill.make_input(N, size, seed=IN_SEED)
    tgt = param_fill.make_labels(N)
    rng = make_rng_r50(N, MASK_SEED, size=size)
    store = {}
    n = record(m, ref_loss, x, tgt, rng, store, "")
    m = m.double()
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)      # also resets the BN running statistics the fp32 passes moved
    record(m, ref_loss, x.double(), tgt, rng, store, "f64_")
    store["meta"] = np.array([N, size, IN_SEED, MASK_SEED], dtype=np.int64)
    name = "udr50_n4.npz" if size == SIZE else f"udr50_n4_s{size}.npz"
    np.savez_compressed(os.path.join(OUT, name), **store)
    print("wrote", name, n, "grads")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else SIZE)
