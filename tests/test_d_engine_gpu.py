"""GPU parity, step level: AbstractEngine.train_unidefense_model (two passes, two AdamW(amsgrad) steps) on the
HIP model against the vectors recorded from the REFERENCE's engine (tests/golden/udeb4_step_n4.npz,
oracle/make_golden_step.py): same seeded inputs / parameters / Bernoulli masks, pass-2 perturbation forced
to `downscale`.

Compared: the ten returned loss scalars and the pass-1 logits (1e-3 relative; the reference's own fp32 and fp64
runs of this step agree to ~1e-6 on the mask means and ~6e-5 on the KL terms), and the parameter updates.
The first Adam steps are sign-like (update ~ lr * g/|g|), so a parameter whose gradient is rounding noise
(e.g. the bias of a BN that only feeds other batch-stat BNs: true gradient 0) moves by +-lr with a random
sign in ANY implementation; the update check (tests/test_step_cpu.py:check_updates) therefore asks for the
agreement fraction the reference's own fp32-vs-fp64 runs reach (~97 % of tensors / ~96 % of elements) - 2 %.
"""
import os

import numpy as np
import pytest
import torch

from oracle import param_fill
from tests import oracle_util as ou
from tests.margins import within

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,cur_step", [("early", 1), ("kl", 50)])
def test_two_pass_step_vs_reference_golden(golden_dir, tag, cur_step, run_mode):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    from unidefense_amd.engine import AbstractEngine
    from unidefense_amd.engine.optim import build_optimizer
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model, perturb

    g = np.load(os.path.join(golden_dir, "udeb4_step_n4.npz"))
    n, size, in_seed, mask_seed, num_steps = [int(v) for v in g["meta"]]
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dev).train()
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    x = param_fill.make_input(n, size, in_seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    m.rng_queue = [ou.make_rng(n, mask_seed, 0.5), ou.make_rng(n, mask_seed + 1, 0.5)]

    eng = AbstractEngine({"config": dict(ou.LAMBDAS)})
    eng.model, eng.device = m, dev
    eng.num_steps, eng.warmup_step = num_steps, 0
    eng.optimizer = build_optimizer(m, dict(name="adamw", lr=1e-4, betas=[0.9, 0.999], weight_decay=5e-6, amsgrad=True))
    eng.scheduler = torch.optim.lr_scheduler.StepLR(eng.optimizer, step_size=22500, gamma=0.5)
    eng.loss_criterion = {"softmax": LOSSES["cross_entropy"], "triplet": LOSSES["aw_triplet"],
                          "kl_div": LOSSES["kl_div"], "fac": LOSSES["factorization"]}
    orig = perturb.perturb_input
    perturb.perturb_input = lambda x_, a, b, c: perturb.downscale(x_)      # forced choice, like the golden
    try:
        scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10, enabled=False)
        eng.optimizer.zero_grad()
        ret = eng.train_unidefense_model(x, tgt, cur_step, scaler, n // 2, n // 2)
    finally:
        perturb.perturb_input = orig
    bad = []
    for k, v in ret.items():
        key = ("out_" if k == "cls_out" else "loss_") + k
        ref = np.asarray(g[f"{tag}_{key}"], dtype=np.float64)
        got = v.detach().double().cpu().numpy()
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
        print(f"  {k}: rel err {err:.3e}")
        if not within("returned " + k, err, 1e-3):
            bad.append((k, err))
    assert not bad, bad
    from tests.test_step_cpu import check_updates
    check_updates(g, tag, {k: (p.detach() - before[k]).cpu() for k, p in m.named_parameters()})


def _make_engine(dev, use_graphs):
    from unidefense_amd.engine import AbstractEngine
    from unidefense_amd.engine.optim import build_optimizer
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.0, drop_connect_rate=0.0)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dev).train()
    m._dec_dropout = False                    # no random draw left in the forward: both engines see the same function
    eng = AbstractEngine({"config": dict(ou.LAMBDAS)})
    eng.model, eng.device, eng.num_steps, eng.warmup_step = m, dev, 100, 0
    eng.use_graphs = use_graphs
    eng.optimizer = build_optimizer(m, dict(name="adamw", lr=1e-4, betas=[0.9, 0.999], weight_decay=5e-6, amsgrad=True))
    eng.scheduler = torch.optim.lr_scheduler.StepLR(eng.optimizer, step_size=22500, gamma=0.5)
    eng.loss_criterion = {"softmax": LOSSES["cross_entropy"], "triplet": LOSSES["aw_triplet"],
                          "kl_div": LOSSES["kl_div"], "fac": LOSSES["factorization"]}
    return eng


@pytest.mark.parametrize("pert", ["downscale", "freq", "efdm"])
def test_graph_captured_step_equals_eager_step(pert):
    """engine.use_graphs: the two passes replayed from hipGraphs (perturbation / optimizer / scaler outside) must
    reproduce the eager step over several steps with changing inputs — same losses, same parameters.  Dropout is
    switched off so that both executions evaluate the same function.  pert = downscale: the perturbation forced by replacing
    perturb_input (the graphed step then perturbs between the replays, as in round 4); freq / efdm: perturb_input untouched and
    the host draws pinned to the style branch (oracle/pins.py): permuted batch -> CORAL (host LAPACK) -> frequency amplitude transfer /
    exact feature-distribution matching between the two replays.  (Round 5 measured a variant that plans the perturbation before
    pass 1 — host draws and CORAL's statistics on a side stream, no host wait between the replays: 57.9 - 58.3 ms per step against
    56.6 - 57.1 for this order, tools/ab_train_step.py history in profiles/r05/train_step_ab.txt — the round-4 order has no gap
    left to remove and stays.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import contextlib
    from oracle import pins
    dev = torch.device("cuda:0")
    from unidefense_amd.model import perturb
    n = 4
    tgt = param_fill.make_labels(n).to(dev)
    xs = [param_fill.make_input(n, 256, 50 + i).to(dev) for i in range(4)]
    orig = perturb.perturb_input
    if pert == "downscale":
        perturb.perturb_input = lambda x_, a, b, c: perturb.downscale(x_)
    try:
        results = {}
        for use_graphs in (False, True):
          with (pins.pinned_draws(pert) if pert != "downscale" else contextlib.nullcontext()):
              eng = _make_engine(dev, use_graphs)
              scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10, enabled=True)     # the reference's GradScaler(2**10)
              rets = []
              for i, x in enumerate(xs):
                  eng.optimizer.zero_grad()
                  rets.append({k: v.detach().float().cpu() for k, v in
                               eng.train_unidefense_model(x, tgt, 50 + i, scaler, n // 2, n // 2).items()})
              results[use_graphs] = (rets, {k: v.detach().cpu().clone() for k, v in eng.model.named_parameters()})
              if use_graphs:
                  assert any("g2" in st for st in eng._graphs.values()), "the graphed path did not capture"
    finally:
        perturb.perturb_input = orig
    (r0, p0), (r1, p1) = results[False], results[True]
    for i, (a, b) in enumerate(zip(r0, r1)):
        for k in a:
            err = (a[k] - b[k]).abs().max().item() / max(a[k].abs().max().item(), 1e-30)
            # the KL mask terms (~1e-3, a second-order difference of two nearly equal masks) amplify the run-to-run
            # rounding of the split-K atomics: two EAGER runs of step 0 already differ by 1.6e-4 there
            assert within(f"[{pert}] step {i} {k}: graph vs eager", err, 2e-2 if k in ("freq_mask_loss", "spat_mask_loss") else 1e-3), (i, k, err)
    # parameters after 8 Adam steps: sign-like first steps move a parameter whose gradient is rounding noise by
    # +-lr per step in ANY two runs (tests/test_step_cpu.py), so compare the distribution, not the worst tensor
    devs = sorted(((p0[k] - p1[k]).abs().max() / (p0[k].abs().max() + 1e-12)).item() for k in p0)
    med, p90, worst = devs[len(devs) // 2], devs[int(0.9 * len(devs))], devs[-1]
    print(f"  4 steps, relative parameter deviation eager vs graphed: median {med:.2e}  90% {p90:.2e}  worst {worst:.2e}")
    ok = [within(f"[{pert}] parameter deviation graph vs eager: median", med, 1e-4), within(f"[{pert}] ... 90 %", p90, 5e-3),
          within(f"[{pert}] ... worst", worst, 0.5)]
    assert all(ok)      # observed (atomics mode): 3e-6 / 8e-4 / 5e-2


def test_second_backward_accumulates_in_one_launch_like_autograd():
    """The train step's second backward adds onto the first's gradients (one zero_grad per step, engine/forgery_engine.py:241;
    backward at abstract_engine.py:281 and :374).  The engine lets the model do that with ONE multi-tensor launch
    (model/unidefense.py:_accumulate_in_place, csrc/optim.hip:ud_multi_add) instead of autograd's AccumulateGrad launch per
    parameter: same gradients bit for bit (fp32 a + b either way), hence the same parameters after the step; and the launch
    count of the eager step falls by the ~500 adds."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    from unidefense_amd.config import override
    from unidefense_amd.model import perturb
    from unidefense_amd.model import unidefense as U
    n = 4
    tgt = param_fill.make_labels(n).to(dev)
    x = param_fill.make_input(n, 256, 61).to(dev)
    orig, orig_min = perturb.perturb_input, U._MULTI_ADD_MIN
    perturb.perturb_input = lambda x_, a, b, c: perturb.downscale(x_)
    res = {}
    try:
        with override(deterministic=True, gemm_tune=False):
            for mode in ("autograd", "multi"):
                U._MULTI_ADD_MIN = 10 ** 9 if mode == "autograd" else orig_min
                eng = _make_engine(dev, False)
                scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10, enabled=True)
                calls = []
                real = U.K.multi_add
                U.K.multi_add = lambda d, s_: (calls.append(len(d)), real(d, s_))[1]
                try:
                    eng.optimizer.zero_grad()
                    eng.train_unidefense_model(x, tgt, 50, scaler, n // 2, n // 2)
                finally:
                    U.K.multi_add = real
                res[mode] = ({k: p.grad.detach().clone() for k, p in eng.model.named_parameters() if p.grad is not None},
                             {k: p.detach().clone() for k, p in eng.model.named_parameters()}, calls)
    finally:
        perturb.perturb_input, U._MULTI_ADD_MIN = orig, orig_min
    assert res["autograd"][2] == [] and len(res["multi"][2]) == 1 and res["multi"][2][0] >= 490, (res["autograd"][2], res["multi"][2])
    for k, g in res["autograd"][0].items():
        assert torch.equal(g, res["multi"][0][k]), k
    for k, p in res["autograd"][1].items():
        assert torch.equal(p, res["multi"][1][k]), k


def _engine_for(m, dev, num_steps):
    from unidefense_amd.engine import AbstractEngine
    from unidefense_amd.engine.optim import build_optimizer
    from unidefense_amd.loss import LOSSES
    eng = AbstractEngine({"config": dict(ou.LAMBDAS)})
    eng.model, eng.device = m, dev
    eng.num_steps, eng.warmup_step = num_steps, 0
    eng.optimizer = build_optimizer(m, dict(name="adamw", lr=1e-4, betas=[0.9, 0.999], weight_decay=5e-6, amsgrad=True))
    eng.scheduler = torch.optim.lr_scheduler.StepLR(eng.optimizer, step_size=22500, gamma=0.5)
    eng.loss_criterion = {"softmax": LOSSES["cross_entropy"], "triplet": LOSSES["aw_triplet"],
                          "kl_div": LOSSES["kl_div"], "fac": LOSSES["factorization"]}
    return eng


def _check_step(g, tag, ret, m, before, loss_tol=1e-3, slack=0.02):
    bad = []
    for k, v in ret.items():
        key = ("out_" if k == "cls_out" else "loss_") + k
        ref = np.asarray(g[f"{tag}_{key}"], dtype=np.float64)
        got = v.detach().double().cpu().numpy()
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
        print(f"  {k}: rel err {err:.3e}")
        # the KL mask terms are second-order differences of two nearly equal masks (values ~1e-3): they amplify fp32
        # rounding — two runs of THIS path with different split-K atomic orders already differ by up to ~1.5e-3 there
        tol = max(loss_tol, 5e-3) if k in ("freq_mask_loss", "spat_mask_loss") else loss_tol
        if not within("returned " + k, err, tol):
            bad.append((k, err))
    assert not bad, bad
    from tests.test_step_cpu import check_updates
    check_updates(g, tag, {k: (p.detach() - before[k]).cpu() for k, p in m.named_parameters()}, slack=slack)


@pytest.mark.parametrize("tag", ["freq", "efdm"])
def test_two_pass_step_style_perturbation_vs_reference_golden(golden_dir, tag, run_mode):
    """The reference ENGINE's step with pass 2 perturbed by the style branch (permuted batch -> CORAL -> frequency
    amplitude transfer / exact feature-distribution matching; model/unidefense.py:177-191, model/modules.py:35-76): the
    ten returned scalars + pass-1 logits within 1e-3, parameter updates as in test_two_pass_step_vs_reference_golden.
    oracle/pins.py replaces the global-RNG draws on both sides by the same seeded stand-ins."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pins
    from unidefense_amd.model import load_model
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "udeb4_step_pert_n4.npz"))
    n, size, in_seed, mask_seed, num_steps = [int(v) for v in g["meta"]]
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dev).train()
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    x = param_fill.make_input(n, size, in_seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    m.rng_queue = [ou.make_rng(n, mask_seed, 0.5), ou.make_rng(n, mask_seed + 1, 0.5)]
    eng = _engine_for(m, dev, num_steps)
    scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10, enabled=False)
    eng.optimizer.zero_grad()
    with pins.pinned_draws(tag) as calls:
        ret = eng.train_unidefense_model(x, tgt, 1, scaler, n // 2, n // 2)
    assert calls == {"perm": 2, "lmda": 1}, calls          # both permutation lists and the transfer's lmda were drawn
    _check_step(g, tag, ret, m, before)


@pytest.mark.parametrize("tag,cur_step", [("early", 1), ("kl", 50)])
def test_udr18_two_pass_step_vs_reference_golden(golden_dir, tag, cur_step, run_mode):
    """BASELINE configs[0] (ResNet18, 128x128, bs 8): the reference engine's two-pass step, pass 2 perturbed by `downscale`."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pins
    from tests.test_c_r18 import make_rng_r18
    from unidefense_amd.model import load_model, perturb
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "udr18_step_n8.npz"))
    n, size, in_seed, mask_seed, num_steps = [int(v) for v in g["meta"]]
    m = load_model("UDR18")(extractor="resnet18", num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dev).train()
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    x = param_fill.make_input(n, size, in_seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    m.rng_queue = [make_rng_r18(n, mask_seed), make_rng_r18(n, mask_seed + 1)]
    eng = _engine_for(m, dev, num_steps)
    scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10, enabled=False)
    eng.optimizer.zero_grad()
    orig = perturb.PERT_FUNCS
    perturb.PERT_FUNCS = [perturb.downscale] * 3           # as the generator did with the reference's list
    try:
        with pins.pinned_draws("downscale"):
            ret = eng.train_unidefense_model(x, tgt, cur_step, scaler, n // 2, n // 2)
    finally:
        perturb.PERT_FUNCS = orig
    # A full engine step cannot pin the ReLU on/off patterns (tests/test_c_r18.py does, for the gradients): ~10 of the 2.4e7
    # units sit within fp32 rounding of 0, each flip moves a few weight gradients by ~1e-2, and the first (sign-like) Adam
    # step turns that into a visibly different update norm for a handful of the 107 tensors (observed 101 / 107 agree).
    _check_step(g, tag, ret, m, before, slack=0.08)
