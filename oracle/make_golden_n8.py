"""TEST INFRASTRUCTURE — tests/golden/udeb4_train_n8.npz: the REFERENCE's train-mode forward + pass-1 loss + backward of
UniDefenseModelEb4 (model/unidefense.py:174-256) at N = 8, both loss variants, in the format of udeb4_train_n4.npz
(oracle/make_golden.py): batch statistics over eight samples are conditioned like the bench's, not like the N = 2…4 fixtures'.

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_n8            (from the repo root, this container only)

The seeds are the first (input, mask) pair whose smallest top-2 gap at the dynamic filters' torch.max is > 1e-3 in the oracle
(oracle.eb4 "_max_gap"), searched from (81, 181) upwards (the widest gap found if none reaches 1e-3) and recorded in `meta`."""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import eb4, make_golden as mg, param_fill, ref_import  # noqa: E402


def main():
    n, drop_rate = 8, 0.5
    sd = param_fill.fill_state_dict(eb4.eb4_state_shapes(2), 0.0, 0.3, torch.float32)
    seeds, best = None, (0.0, None)
    for s in range(81, 480):
        x = param_fill.make_input(n, 256, seed=s)
        rng = mg.make_rng(n, seed=100 + s, drop_rate=drop_rate)
        with torch.no_grad():
            gap = eb4.forward_eb4(sd, x, training=True, drop_rate=drop_rate, rng=rng)["_max_gap"].item()
        print(f"seeds ({s}, {100 + s}): smallest arg-max gap {gap:.2e}")
        if gap > best[0]:
            best = (gap, (s, 100 + s))
        if gap > 1e-3:
            seeds = (s, 100 + s)
            break
    if seeds is None:          # eight samples: twice the arg-max sites of the N = 4 fixture; the widest gap found is recorded
        seeds = best[1]
    print("using", seeds, "gap", best[0])
    ref_model, ref_loss = ref_import.import_reference()
    torch.manual_seed(0)
    m = ref_model.load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=drop_rate)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    x = param_fill.make_input(n, 256, seed=seeds[0])
    tgt = param_fill.make_labels(n)
    rng = mg.make_rng(n, seed=seeds[1], drop_rate=drop_rate)
    store = {}
    for variant, lam in (("full", mg.LAMBDAS), ("smooth", dict(mg.LAMBDAS, lambda_recons=0.0, lambda_freq=0.0))):
        out, losses = mg.run_reference_train(m, ref_loss, x, tgt, rng, drop_rate, lam)
        if variant == "full":
            mg.pack_outputs(out, "", store)
        for k, v in losses.items():
            store[f"{variant}_loss_" + k] = np.array(v.item(), dtype=np.float64)
        names, norms, heads, maxabs = [], [], [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            norms.append(p.grad.double().norm().item())
            maxabs.append(p.grad.abs().max().item())
            h = torch.zeros(8)
            f = p.grad.flatten()[:8]
            h[: f.numel()] = f
            heads.append(h.numpy())
        store["grad_names"] = np.array(names)
        store[f"{variant}_grad_norms"] = np.array(norms, dtype=np.float64)
        store[f"{variant}_grad_maxabs"] = np.array(maxabs, dtype=np.float64)
        store[f"{variant}_grad_heads"] = np.stack(heads)
    store["meta"] = np.array([n, 256, seeds[0], seeds[1]], dtype=np.int64)
    np.savez_compressed(os.path.join(mg.OUT, "udeb4_train_n8.npz"), **store)
    print("wrote udeb4_train_n8.npz  (%d grads x 2 variants, seeds %s)" % (len(names), seeds))


if __name__ == "__main__":
    main()
