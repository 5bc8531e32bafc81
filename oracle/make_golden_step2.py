"""TEST INFRASTRUCTURE — more step-level goldens of the REFERENCE's AbstractEngine.train_unidefense_model
(engine/abstract_engine.py:207-381, imported from /root/reference, this container only):

  tests/golden/udeb4_step_pert_n4.npz   UDEB4, N=4, 256x256, cur_step 1, pass 2 perturbed by the STYLE branch of
        model/unidefense.py:177-191 (permuted batch, CORAL colour transfer, then) 'freq': FrequencyStyleTransfer
        (model/modules.py:35-55) / 'efdm': SpatialStyleTransfer (:58-76)
  tests/golden/udr18_step_n8.npz        UDR18 (BASELINE configs[0]: ResNet18, 128x128, bs 8), pass 2 perturbed by
        `downscale`, cur_step 1 ('early') and 50 ('kl': KL mask-alignment losses)

Every random draw is pinned: Bernoulli masks injected at the F.dropout / drop_connect sites, torch.rand(1) (branch),
torch.randint (choice), torch.randperm (the engine's permutation lists) and the transfer's lmda = torch.rand((B,1,1[,1]))
replaced by seeded stand-ins — oracle/pins.py holds the SAME stand-ins for the GPU test, which patches them around the
HIP engine (its perturbation code draws in the reference's order, tests/test_c_perturb.py).
Stored: the returned loss scalars + pass-1 cls_out, per-parameter update norms and heads, each also from a float64 run.

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_step2
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from oracle import ref_import, param_fill, eb4, pins                     # noqa: E402
from oracle.make_golden import make_rng, LAMBDAS                            # noqa: E402
from oracle.make_golden_r18 import make_rng_r18                             # noqa: E402
from oracle.make_golden_step import param_groups_weight_decay, OPT, WD      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
NUM_STEPS = 100
CASES = {"UDEB4": dict(n=4, size=256, in_seed=38, mask_seed=138, ctor=dict(extractor="efficientnet-b4")),
         "UDR18": dict(n=8, size=128, in_seed=42, mask_seed=142, ctor=dict(extractor="resnet18"))}


def run(model_name, pert, cur_step, dtype=torch.float32):
    case = CASES[model_name]
    n, drop_rate = case["n"], 0.5
    ref_model, ref_loss = ref_import.import_reference()
    engmod = ref_import.import_abstract_engine()
    import model.efficientnet.model as effmod
    import model.unidefense as udmod
    torch.manual_seed(0)
    m = ref_model.load_model(model_name)(num_classes=2, drop_rate=drop_rate, **case["ctor"])
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dtype)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    x = param_fill.make_input(n, case["size"], seed=case["in_seed"]).to(dtype)
    tgt = param_fill.make_labels(n)
    mk = make_rng if model_name == "UDEB4" else make_rng_r18
    rngs = [mk(n, case["mask_seed"], drop_rate), mk(n, case["mask_seed"] + 1, drop_rate)]

    eng = engmod.AbstractEngine.__new__(engmod.AbstractEngine)
    eng.model = m.train()
    eng.device = torch.device("cpu")
    eng.config = {"config": dict(LAMBDAS)}
    eng.num_steps, eng.warmup_step = NUM_STEPS, 0
    eng.optimizer = torch.optim.AdamW(param_groups_weight_decay(m.named_parameters(), WD), **OPT)
    eng.scheduler = torch.optim.lr_scheduler.StepLR(eng.optimizer, step_size=22500, gamma=0.5)
    eng.loss_criterion = {"softmax": ref_loss.LOSSES["cross_entropy"], "triplet": ref_loss.LOSSES["aw_triplet"],
                          "kl_div": ref_loss.LOSSES["kl_div"], "fac": ref_loss.LOSSES["factorization"]}

    F = torch.nn.functional
    orig = dict(dropout=F.dropout, dc=effmod.drop_connect, pert=udmod.PERT_FUNCS)
    state = {"pass": 0, "queue": [], "dc": 0}
    dc_order = [i for i, b in enumerate(eb4.eb4_arch()["blocks"]) if b["skip"] and i > 0]

    def fake_dropout(inp, p=0.5, training=True, inplace=False):
        if not state["queue"]:      # a new forward pass begins
            state["queue"] = [("dec_keep", 0.2), ("emb_keep", drop_rate), ("feat_keep", drop_rate)]
        name, pp = state["queue"].pop(0)
        assert abs(pp - p) < 1e-12 and training
        rng = rngs[state["pass"]]
        if not state["queue"]:
            state["pass"] += 1
        scale = rng[name].to(inp.dtype) / (1.0 - p)
        return inp.mul_(scale) if inplace else inp * scale

    def fake_dc(inputs, p, training):
        idx = dc_order[state["dc"] % len(dc_order)]      # call order within a forward = block order
        state["dc"] += 1
        assert abs(p - 0.2 * idx / 32) < 1e-12 and training
        return inputs / (1 - p) * rngs[state["pass"]]["drop_connect"][idx].reshape(-1, 1, 1, 1)

    F.dropout, effmod.drop_connect = fake_dropout, fake_dc
    udmod.PERT_FUNCS = [udmod.pert_ds, udmod.pert_ds, udmod.pert_ds]
    try:
        with pins.pinned_draws(pert):
            scaler = torch.cuda.amp.GradScaler(2 ** 10, enabled=False)
            eng.optimizer.zero_grad()
            ret = eng.train_unidefense_model(x, tgt, cur_step, scaler, n // 2, n // 2)
    finally:
        F.dropout, effmod.drop_connect = orig["dropout"], orig["dc"]
        udmod.PERT_FUNCS = orig["pert"]
    store = {}
    for k, v in ret.items():
        store[("out_" if k == "cls_out" else "loss_") + k] = v.detach().numpy()
    names, dn, heads = [], [], []
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        names.append(k)
        dn.append((p.detach() - before[k]).double().norm().item())
        h = torch.zeros(8, dtype=dtype)
        f = (p.detach() - before[k]).flatten()[:8]
        h[: f.numel()] = f
        heads.append(h.numpy())
    store["names"] = np.array(names)
    store["delta_norms"] = np.array(dn)
    store["delta_heads"] = np.stack(heads)
    return store


def record(out, tag, model_name, pert, step):
    st = run(model_name, pert, step)
    for k, v in st.items():
        out[f"{tag}_{k}"] = v
    print(model_name, tag, {k: float(v) for k, v in st.items() if k.startswith("loss_")})
    st64 = run(model_name, pert, step, torch.float64)          # the conditioning yardstick (see make_golden_step.py)
    for k, v in st64.items():
        if k.startswith(("loss_", "out_", "delta_")):
            out[f"{tag}64_{k}"] = v


def main():
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=0, world_size=1)     # the step calls dist.barrier()
    c = CASES["UDEB4"]
    out = {}
    for tag in ("freq", "efdm"):
        record(out, tag, "UDEB4", tag, 1)
    out["meta"] = np.array([c["n"], c["size"], c["in_seed"], c["mask_seed"], NUM_STEPS], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "udeb4_step_pert_n4.npz"), **out)
    print("wrote udeb4_step_pert_n4.npz")
    c = CASES["UDR18"]
    out = {}
    for tag, step in (("early", 1), ("kl", 50)):
        record(out, tag, "UDR18", "downscale", step)
    out["meta"] = np.array([c["n"], c["size"], c["in_seed"], c["mask_seed"], NUM_STEPS], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "udr18_step_n8.npz"), **out)
    print("wrote udr18_step_n8.npz")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
