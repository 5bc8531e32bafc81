"""TEST INFRASTRUCTURE — step-level golden: the REFERENCE's AbstractEngine.train_unidefense_model
(engine/abstract_engine.py:207-381, imported from /root/reference, this container only) run for one full
two-pass train step of UDEB4 (N=4, 256x256, fp32 CPU, AdamW(amsgrad) with the UniAttack hyper-parameters of
config_template/uniatt/Prot1/model_udeb4.yml), with every random draw pinned:
  * Bernoulli masks of both passes injected (pass 1: mask seed, pass 2: mask seed + 1) like make_golden.py;
  * the perturbation of pass 2 forced to `downscale` (model/modules.py:19-21), the one deterministic member
    of PERT_FUNCS (the branch draw torch.rand(1) and the index draw torch.randint are patched);
  * cur_step = 1 of num_steps = 100 (mask losses are means) and cur_step = 50 (KL mask-alignment losses).

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_step
Writes tests/golden/udeb4_step_n4.npz: the returned loss scalars, pass-1 cls_out, and per-parameter
update norms |p_after - p_before| with the first 8 elements of the update; plus the same quantities of a
float64 run of the reference ('<tag>64_*'), the yardstick for how ill-conditioned each quantity is.
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from oracle import ref_import, param_fill, eb4            # noqa: E402
from oracle.make_golden import make_rng, LAMBDAS          # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N, IN_SEED, MASK_SEED, NUM_STEPS = 4, 38, 138, 100
OPT = dict(lr=1e-4, betas=(0.9, 0.999), amsgrad=True)
WD = 5e-6


def param_groups_weight_decay(named_params, weight_decay):
    """timm.optim.optim_factory.param_groups_weight_decay as the reference's engines use it
    (engine/forgery_engine.py:15,152): no decay for 1-D / scalar tensors and '.bias'."""
    decay, no_decay = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith(".bias")) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def run(cur_step, dtype=torch.float32):
    ref_model, ref_loss = ref_import.import_reference()
    engmod = ref_import.import_abstract_engine()
    import model.efficientnet.model as effmod
    import model.unidefense as udmod
    torch.manual_seed(0)
    drop_rate = 0.5
    m = ref_model.load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=drop_rate)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dtype)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    x = param_fill.make_input(N, 256, seed=IN_SEED).to(dtype)
    tgt = param_fill.make_labels(N)
    rngs = [make_rng(N, MASK_SEED, drop_rate), make_rng(N, MASK_SEED + 1, drop_rate)]

    eng = engmod.AbstractEngine.__new__(engmod.AbstractEngine)
    eng.model = m.train()
    eng.device = torch.device("cpu")
    eng.config = {"config": dict(LAMBDAS)}
    eng.num_steps, eng.warmup_step = NUM_STEPS, 0
    eng.optimizer = torch.optim.AdamW(param_groups_weight_decay(m.named_parameters(), WD), **OPT)
    eng.scheduler = torch.optim.lr_scheduler.StepLR(eng.optimizer, step_size=22500, gamma=0.5)
    eng.loss_criterion = {"softmax": ref_loss.LOSSES["cross_entropy"], "triplet": ref_loss.LOSSES["aw_triplet"],
                          "kl_div": ref_loss.LOSSES["kl_div"], "fac": ref_loss.LOSSES["factorization"]}

    # ---- pin the randomness -------------------------------------------------------------------------
    F = torch.nn.functional
    orig = dict(dropout=F.dropout, dc=effmod.drop_connect, rand=torch.rand, randint=torch.randint,
                pert=udmod.PERT_FUNCS)
    state = {"pass": 0, "queue": [], "dc": 0}
    arch = eb4.eb4_arch()
    dc_order = [i for i, b in enumerate(arch["blocks"]) if b["skip"] and i > 0]

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

    def fake_rand(*size, **kw):
        if size == (1,):
            return torch.zeros(1)            # `torch.rand(1) > 0.5` False -> PERT_FUNCS branch (unidefense.py:178,195)
        return orig["rand"](*size, **kw)

    def fake_randint(low, high=None, size=None, **kw):
        return torch.zeros(size if size is not None else (1,), dtype=torch.int64)

    F.dropout, effmod.drop_connect = fake_dropout, fake_dc
    torch.rand, torch.randint = fake_rand, fake_randint
    udmod.PERT_FUNCS = [udmod.pert_ds, udmod.pert_ds, udmod.pert_ds]
    try:
        scaler = torch.cuda.amp.GradScaler(2 ** 10, enabled=False)
        eng.optimizer.zero_grad()
        ret = eng.train_unidefense_model(x, tgt, cur_step, scaler, N // 2, N // 2)
    finally:
        F.dropout, effmod.drop_connect = orig["dropout"], orig["dc"]
        torch.rand, torch.randint = orig["rand"], orig["randint"]
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


def main():
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)     # the step calls dist.barrier()
    out = {}
    for tag, step in (("early", 1), ("kl", 50)):
        st = run(step)
        for k, v in st.items():
            out[f"{tag}_{k}"] = v
        print(tag, {k: float(v) for k, v in st.items() if k.startswith("loss_")})
        # the same step in float64: |fp32 - fp64| of the reference itself is the conditioning yardstick the
        # parity test scales its tolerances with (the first Adam steps are sign-like, see tests/test_d_engine_gpu.py)
        st64 = run(step, torch.float64)
        for k, v in st64.items():
            if k.startswith(("loss_", "out_", "delta_")):
                out[f"{tag}64_{k}"] = v
        print(tag + "64", {k: float(v) for k, v in st64.items() if k.startswith("loss_")})
    out["meta"] = np.array([N, 256, IN_SEED, MASK_SEED, NUM_STEPS], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "udeb4_step_n4.npz"), **out)
    print("wrote udeb4_step_n4.npz")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
