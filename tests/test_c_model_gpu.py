"""GPU parity tests, model level: UniDefenseModelEb4 on the HIP kernels against
  (a) the golden vectors recorded from the REFERENCE (tests/golden/*.npz, oracle/make_golden.py), and
  (b) the oracle run on the CPU on the same seeded inputs (float64 = "exact", float32 = the reference's own
      arithmetic).

Tolerances (BASELINE.json north_star: "within 1e-3 rel fp32"):
  * forward outputs, losses: 1e-3 relative to the tensor's max magnitude (observed ~1e-6..3e-5).
  * gradients: per tensor  max|d| <= 1e-3 * max|ref| + 2e-5  —  OR  <= 5x the error the oracle's OWN float32
    run makes against its float64 run on that tensor.  The second clause exists because this step is not
    conditioned to 1e-3 in fp32 everywhere: with batch statistics over a batch of 2..4 the reference's fp32
    CPU path itself is off by up to ~3e-3 on the scalar sf_coef gradients (global sums with heavy
    cancellation).  "As accurate as the reference's fp32 arithmetic" is the meaningful bar there.
  * two discontinuous gradients are kept out of the comparison by construction, not by tolerance:
    - torch.max over channels in the dynamic filters (arg-max flips at near-ties): fixtures use batches whose
      smallest top-2 gap is > 1e-3 (oracle "_max_gap");
    - sign() of the two L1 reconstruction terms: the 'smooth' variant sets lambda_recons = lambda_freq = 0; the
      'full' variant (the reference's pass-1 loss) is held to 2e-2 because a single element of rec - x or of
      its spectrum (~4e5 each) within rounding of zero flips its sign under any change of summation order and
      moves the upstream gradient by ~2/sqrt(4e5) ~ 3e-3 of its norm.
"""
import functools
import os

import numpy as np
import pytest
import torch

from oracle import eb4, param_fill
from tests import oracle_util as ou
from tests.margins import within

pytestmark = pytest.mark.gpu
RTOL = 1e-3
GRAD_RTOL, GRAD_ATOL = 1e-3, 2e-5
FULL_RTOL = 2e-2
# BN2 biases whose gradient is analytically zero (the block's output reaches nothing but BatchNorm-ed paths)
# the N = 8 fixture's gate gradients on which the REFERENCE's own fp32 record is 3e-5 off the float64 value (5.5e-3): held to 1e-2
# of the reference there and to the plain 1e-3 against float64 (test_train_grads_vs_oracle_elementwise_n8_plain_bound)
# ... and blocks.23's gate gradient (1.085e-2): the reference's fp32 record sits 9.1e-6 (8.4e-4 of the value) from the float64 result —
# measured twice against both: this path with the one-pass project forward 1.25e-5 from the record and 3.4e-6 from float64, without
# it 6.9e-6 and 2.2e-6 (profiles/r06/pj_bwd_fused.txt) — so "1e-3 of the record" leaves 1.7e-6 around the true value.  It meets the
# floored bound of this test (4e-4) and the plain 1e-3 against float64 (3.1e-4); the name only admits it to the purely relative check.
N8_REF_FP32_NOISE = ("backbone._blocks.9._depthwise_conv.sf_coef", "backbone._blocks.23._depthwise_conv.sf_coef")
STRUCT_ZERO_GRADS = {f"backbone._blocks.{i}._bn2.bias" for i in list(range(16)) + [30, 31]}
# ... and the same biases in the stage whose output x_b4 also reaches the decoder / the triplet feature: still almost
# entirely cancelled (reference norm 1e-4 against 1e-2 .. 1e+1 elsewhere)
NEAR_ZERO_GRADS = {f"backbone._blocks.{i}._bn2.bias" for i in range(16, 22)}
# ... and the affine parameters of the decoder's InstanceNorms (dec_block1 / dec_block2, layers 1 / 4 / 7): in the smooth
# loss variant their gradients have norms 7e-5 .. 1e-3 (the decoder convs: 4e-3 .. 2e-2) as sums over up to 65k pixels that
# cancel to ~1 % — an absolute error of 5e-7, the accumulation-order noise of those sums (it depends on which GEMM plan
# the tuner picked for the surrounding convs), exceeds 1e-3 of what is left.  They still have to meet the 2e-5 floor and
# the < 1e-3 reference-norm condition below.
NEAR_ZERO_GRADS |= {f"dec_block{b}.{l}.{w}" for b in (1, 2) for l in (1, 4, 7) for w in ("weight", "bias")}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _model(dev, sf, fuse):
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=sf, fuse_coef=fuse)
    return m.to(dev)


def _close(a, b, name):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    print(f"  {name}: rel err {err:.3e}")
    return name, err


def _check_outputs(out, g):
    ld = out["loss_dict"]
    res = [_close(out["cls_out"], g["cls_out"], "cls_out"),
           _close(torch.nn.functional.adaptive_avg_pool2d(out["rec"].detach().cpu(), 8), g["rec_pool8"], "rec"),
           _close(ld["factorization"][:, :64], g["factorization"], "factorization")]
    for k in ("freq_mask", "spat_mask", "spatial", "freq"):
        res.append(_close(ld[k], g[k], k))
    for i in range(3):
        res.append(_close(ld["triplet"][i], g[f"triplet{i}"], f"triplet{i}"))
    bad = [(n, e) for n, e in res if not within("output " + n, e, RTOL)]
    assert not bad, bad


def _pass1_loss(out, tgt, lam):
    from unidefense_amd.loss import LOSSES
    ld = out["loss_dict"]
    n_real = tgt.numel() // 2
    trip = sum(LOSSES["aw_triplet"](f, tgt) for f in ld["triplet"])
    cls = LOSSES["cross_entropy"](out["cls_out"], tgt)
    real_rec = ld["spatial"].narrow(0, 0, n_real).mean()
    real_freq = ld["freq"].narrow(0, 0, n_real).mean()
    total = cls + lam["lambda_mask"] * ld["freq_mask"].mean() + lam["lambda_mask"] * ld["spat_mask"].mean() + \
        lam["lambda_triplet"] * trip + lam["lambda_recons"] * real_rec + lam["lambda_freq"] * real_freq
    return {"total_loss": total, "cls_loss": cls, "triplet_loss": trip, "real_rec_loss": real_rec,
            "real_freq_loss": real_freq}


@pytest.mark.parametrize("fname,sf,fuse", [("udeb4_eval_n2.npz", 0.0, 0.3), ("udeb4_eval_n2_init.npz", -10.0, 0.0)])
def test_eval_vs_reference_golden(golden_dir, fname, sf, fuse):
    dev = _dev()
    g = np.load(os.path.join(golden_dir, fname))
    n, size, seed = [int(v) for v in g["meta"]]
    m = _model(dev, sf, fuse).eval()
    x = param_fill.make_input(n, size, seed).to(dev)
    with torch.no_grad():
        out = m(x)
    _check_outputs(out, g)


def test_eval_intermediates_vs_oracle():
    """Stage-by-stage comparison with the oracle (localises a failing kernel)."""
    dev = _dev()
    m = _model(dev, 0.0, 0.3).eval()
    x = param_fill.make_input(2, 256, 5)
    sd = ou.oracle_state(0.0, 0.3)
    with torch.no_grad():
        ref = eb4.forward_eb4(sd, x, training=False)
        got = m._run(x.to(dev), None, None)
    bad = []
    for k in ("x_b0", "x_b1", "x_b2", "x_b3", "x_b4", "x_b5", "att_out", "x_b6", "dec1", "dec2"):
        n_, e = _close(got["_feats"][k].permute(0, 3, 1, 2), ref["_feats"][k], k)
        if not within("stage " + k, e, RTOL):
            bad.append((n_, e))
    n_, e = _close(got["_feats"]["dec3"], ref["_feats"]["dec3"], "dec3")
    if not within("stage dec3", e, RTOL):
        bad.append((n_, e))
    assert not bad, bad


@pytest.mark.parametrize("variant,fname", [("smooth", "udeb4_train_n4.npz"), ("full", "udeb4_train_n4.npz"),
                                           ("smooth", "udeb4_train_n8.npz"), ("full", "udeb4_train_n8.npz")])
def test_train_fwd_bwd_vs_reference_golden(golden_dir, variant, fname, run_mode):
    """(udeb4_train_n8.npz, round 5: batch statistics over EIGHT samples, oracle/make_golden_n8.py.)  Outputs, losses and all 504 parameter gradients (norm + first 8 elements) of the train-mode step vs the
    vectors recorded from the reference (fp32 CPU): EVERY tensor within 1e-3 of its norm (+ the 2e-5 floor), both loss
    variants (observed: 3.5e-4 smooth / 7.4e-4 full at worst).  The floor matters for exactly one family: BN2's bias in
    the stages whose output only ever feeds another BatchNorm (blocks 0-15, 30, 31) has a gradient that is
    analytically ZERO (1e-15 in the float64 oracle: a per-channel shift is removed by the next normalisation); the
    reference's fp32 run and this one both hold ~5e-6 of rounding noise there."""
    dev = _dev()
    g = np.load(os.path.join(golden_dir, fname))
    n, size, seed, mseed = [int(v) for v in g["meta"]]
    lam = ou.LAMBDAS if variant == "full" else ou.SMOOTH_LAMBDAS
    rtol = GRAD_RTOL
    m = _model(dev, 0.0, 0.3).train()
    x = param_fill.make_input(n, size, seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    rng = ou.make_rng(n, mseed, 0.5)
    out = m(x, rng=rng)
    _check_outputs(out, g)
    ls = _pass1_loss(out, tgt, lam)
    for k, v in ls.items():
        _, e = _close(v, g[f"{variant}_loss_" + k], k)
        assert within("loss " + k, e, RTOL), (k, e)
    ls["total_loss"].backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(m.named_parameters())
    rows = []
    for i, k in enumerate(names):
        gr = params[k].grad
        assert gr is not None, k
        ref_norm = float(g[f"{variant}_grad_norms"][i])
        err = abs(gr.double().norm().item() - ref_norm)
        head = gr.flatten()[:8].cpu().numpy()
        herr = float(np.abs(head - g[f"{variant}_grad_heads"][i][: head.size]).max())
        rows.append((max(err, herr) / (ref_norm + GRAD_ATOL / rtol), k, err, herr, ref_norm))
    rows.sort(reverse=True)
    within_1e3 = sum(1 for r in rows if r[0] <= 1e-3)
    print(f"  {within_1e3}/{len(rows)} gradient tensors within 1e-3 of the reference; worst:")
    for r in rows[:12]:
        print("  rel %.3e  %-58s norm err %.3e head err %.3e ref norm %.3e" % r)
    # The scalar sf_coef gradients are single global sums with heavy cancellation (header): on the N = 8 fixture the REFERENCE's
    # own fp32 value of one of them (blocks.9, 5.5e-3) sits 3e-5 from the float64 result, which this path reproduces to the plain
    # 1e-3 bound (test_train_grads_vs_oracle_elementwise_n8_plain_bound: 504 of 504).  Against the reference's fp32 record such a
    # scalar is held to 1e-2 of its value; at most two may need it, every other tensor meets the bound above.
    # The exemption is pinned to the NAMED tensor (N8_REF_FP32_NOISE): any other scalar gate leaving the bound fails.
    scal = [r for r in rows if not r[0] <= rtol and n >= 8 and r[1] in N8_REF_FP32_NOISE and max(r[2], r[3]) <= 1e-2 * r[4]]
    rows_strict = [r for r in rows if r not in scal]
    within(f"worst of 504 gradient tensors (N = {n}): max(norm err, head err) / (ref norm + floor)", rows_strict[0][0], rtol)
    bad = [r for r in rows_strict if not r[0] <= rtol]
    assert not bad, bad[:10]
    assert within(f"N = {n} {variant}: scalar sf_coef gradients held to 1e-2 of the reference's fp32 value instead", len(scal), 1 if n >= 8 else 0)
    assert len(rows) == 504 and within_1e3 >= len(rows) - len(scal)
    # Who needs the absolute floor at all?  Only gradients that are zero up to rounding on BOTH sides: BatchNorm biases
    # of the STRUCT_ZERO_GRADS family, reference norm < 1e-4 (against 1e-2 .. 1e+1 for every other tensor), and their
    # nearly cancelled siblings NEAR_ZERO_GRADS.
    floor_users = [r for r in rows if max(r[2], r[3]) > 1e-3 * r[4]]
    print("  tensors outside a purely relative 1e-3:", [(r[1], "%.1e" % r[4]) for r in floor_users])
    odd = [(r[1], r[4]) for r in floor_users
           if not ((r[1] in STRUCT_ZERO_GRADS and r[4] < 1e-4) or (r[1] in NEAR_ZERO_GRADS and r[4] < 1e-3) or
                   (n >= 8 and r[1] in N8_REF_FP32_NOISE and max(r[2], r[3]) <= 1e-2 * r[4]))]
    assert not odd, odd


# (a third case, ("smooth", 2), ran until round 3 and was dropped for its 80 - 100 s of CPU oracle time; since the oracle runs on
# the job's CPU quota (tests/oracle_util.py:fit_cpu_threads) a case costs ~8 s: ("full", 4) on other seeds holds the L1 tails'
# backward on batch statistics over four samples too)
@functools.lru_cache(maxsize=None)
def _oracle_grads(variant, n, seeds):
    """the float64 and float32 CPU oracle runs of a case (~8 s on 16 threads): shared by the run modes"""
    lam = ou.SMOOTH_LAMBDAS if variant == "smooth" else ou.LAMBDAS
    x = param_fill.make_input(n, 256, seeds[0])
    tgt = param_fill.make_labels(n)
    rng = ou.make_rng(n, seeds[1], 0.5)
    sd = ou.oracle_state(0.0, 0.3, dtype=torch.float64, requires_grad=True)
    o64, _ = ou.oracle_train_pass1(sd, x.double(), tgt, rng, 0.5, lam)
    assert o64["_max_gap"].item() > 1e-3
    sd32 = ou.oracle_state(0.0, 0.3, requires_grad=True)
    ou.oracle_train_pass1(sd32, x, tgt, rng, 0.5, lam)
    return sd, sd32


@pytest.mark.parametrize("variant,n,seeds", [("smooth", 4, (38, 138)), ("full", 2, (38, 138)), ("full", 4, (64, 164))])
def test_train_grads_vs_oracle_elementwise(variant, n, seeds, run_mode):
    """Every parameter gradient, element by element, against the oracle in FLOAT64 on the CPU (same seeded
    inputs, parameters and masks), with the oracle's own float32 run as the conditioning yardstick."""
    dev = _dev()
    lam = ou.SMOOTH_LAMBDAS if variant == "smooth" else ou.LAMBDAS
    rtol = GRAD_RTOL if variant == "smooth" else FULL_RTOL
    if n == 2:
        rtol = max(rtol, 2e-3)      # batch statistics over TWO samples: the worst-conditioned case (see header)
    x = param_fill.make_input(n, 256, seeds[0])
    tgt = param_fill.make_labels(n)
    rng = ou.make_rng(n, seeds[1], 0.5)
    sd, sd32 = _oracle_grads(variant, n, seeds)
    m = _model(dev, 0.0, 0.3).train()
    out = m(x.to(dev), rng=rng)
    _pass1_loss(out, tgt.to(dev), lam)["total_loss"].backward()
    rows = []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        ref = sd[k].grad
        d = (p.grad.detach().double().cpu() - ref).abs().max().item()
        s = ref.abs().max().item()
        d32 = (sd32[k].grad.double() - ref).abs().max().item()
        bound = max(rtol * s + GRAD_ATOL, 5.0 * d32)
        rows.append((d / bound, k, d, s, d32))
    rows.sort(reverse=True)
    n_strict = sum(1 for r in rows if r[2] <= GRAD_RTOL * r[3] + GRAD_ATOL)
    print(f"  {n_strict}/{len(rows)} tensors within the plain 1e-3 bound; worst (err/bound):")
    for r in rows[:15]:
        print("  %.3f  %-58s maxerr %.3e  maxref %.3e  cpu-fp32-err %.3e" % r)
    within("worst of the gradient tensors: max err / max(rtol * scale + floor, 5 x oracle fp32-vs-fp64 err)", rows[0][0], 1.0)
    bad = [r for r in rows if not r[0] < 1.0]
    assert not bad, bad[:10]
    # Which tensors need more than the plain 1e-3 bound?  Only the 32 scalar sf_coef gradients: each is ONE global sum
    # over a whole feature map of sigma'(alpha) * dy * (freq - spat) with heavy cancellation, worst over a batch of TWO
    # (the reference's own fp32 run misses the same ones by as much).  Observed: none at N = 4, two or three at N = 2,
    # a different handful as the summation order of the kernels changes — so the KIND is asserted, the count recorded.
    loose = [r[1] for r in rows if not r[2] <= GRAD_RTOL * r[3] + GRAD_ATOL]
    assert within("tensors outside the plain 1e-3 bound (all of them scalar sf_coef gradients), of 32", len(loose), 32)
    assert all(k.endswith("sf_coef") for k in loose), loose


@pytest.mark.parametrize("variant", ["smooth", "full"])
def test_train_grads_vs_oracle_elementwise_n8_plain_bound(variant, run_mode):
    """The same element-wise comparison at N = 8 (seeds of tests/golden/udeb4_train_n8.npz) held to the PLAIN bound
    max|d| <= rtol * max|ref| + 2e-5 for every one of the 504 tensors — no `5 x the oracle's own fp32 error` clause: with
    batch statistics over eight samples the step is conditioned well enough that the escape is not needed (rtol 1e-3 smooth;
    the L1 tails' sign() gradients keep 2e-2 for the full loss, see the header).  The count of tensors outside the bound is
    recorded (profiles/r05/margins.md) and must be zero."""
    dev = _dev()
    n, seeds = 8, (130, 230)
    lam = ou.SMOOTH_LAMBDAS if variant == "smooth" else ou.LAMBDAS
    rtol = GRAD_RTOL if variant == "smooth" else FULL_RTOL
    x = param_fill.make_input(n, 256, seeds[0])
    tgt = param_fill.make_labels(n)
    rng = ou.make_rng(n, seeds[1], 0.5)
    sd, sd32 = _oracle_grads(variant, n, seeds)
    m = _model(dev, 0.0, 0.3).train()
    out = m(x.to(dev), rng=rng)
    _pass1_loss(out, tgt.to(dev), lam)["total_loss"].backward()
    rows = []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        ref = sd[k].grad
        d = (p.grad.detach().double().cpu() - ref).abs().max().item()
        s = ref.abs().max().item()
        d32 = (sd32[k].grad.double() - ref).abs().max().item()
        rows.append((d / (rtol * s + GRAD_ATOL), k, d, s, d32))
    rows.sort(reverse=True)
    for r in rows[:10]:
        print("  %.3f  %-58s maxerr %.3e  maxref %.3e  cpu-fp32-err %.3e" % r)
    loose = [r for r in rows if not r[0] <= 1.0]
    oracle_loose = sum(1 for r in rows if r[4] > rtol * r[3] + GRAD_ATOL)
    print(f"  outside the plain bound: {len(loose)} of {len(rows)} (the oracle's own fp32 run: {oracle_loose})")
    assert len(rows) == 504
    within(f"N = 8 {variant}: worst gradient tensor, max err / (rtol * scale + floor), no fp32-yardstick clause", rows[0][0], 1.0)
    assert within(f"N = 8 {variant}: tensors outside the plain bound, of 504", len(loose), 0), loose[:8]


def test_weight_planes_follow_raw_pointer_weight_updates():
    """The planes of all conv weights are re-made by two launches at the start of EVERY forward — train and eval — and handed out
    only inside that forward (kernels._WeightPlaneBatch).  The HIP optimizer writes weights through raw pointers, which no
    Tensor._version sees: after such an update an eval forward must see the NEW weights.  Simulated with `.data` writes on the
    spectral / expand / project weights; the outputs must equal, bit for bit, those of the same forward with the batch off."""
    from unidefense_amd import kernels as K
    from unidefense_amd.config import override
    dev = _dev()
    torch.manual_seed(3)
    with override(spectral_p2="on", deterministic=True, gemm_tune=False):
        m = _model(dev, 0.0, 0.3)
        x = param_fill.make_input(2, 256, seed=5).to(dev)
        tgt = param_fill.make_labels(2).to(dev)
        m.train()
        for _ in range(2):                       # first forward registers the weights, the second runs the batch
            out = m(x, rng=ou.make_rng(2, 12, 0.5))
            (out["cls_out"].sum() + out["loss_dict"]["spatial"].mean()).backward()
        batch = m.__dict__["_ud_weight_planes"]
        assert len(batch.entries) >= 60 and not batch.active
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if n_.endswith(("freq_conv.weight", "_expand_conv.weight", "_project_conv.weight")):
                    p.data.mul_(1.25)            # like ud_adamw_multi: the version counter does not move
        m.eval()
        with torch.no_grad():
            got = m(x)
            with override(weight_plane_batch=False):
                want = m(x)
        torch.cuda.synchronize()
        assert torch.equal(got["cls_out"], want["cls_out"]) and torch.equal(got["rec"], want["rec"])
        assert not batch.active


def test_eval_mode_backward_vs_oracle():
    """Backward through an EVAL-mode forward (BatchNorm on its running statistics, no dropout / drop-connect; the reference is
    plain autograd, so fine-tuning on frozen statistics works there): parameter gradients of a smooth scalar of the outputs
    against the oracle's autograd in float64, held to the same per-tensor bar as the training gradients."""
    dev = _dev()
    n = 1          # nothing in an eval-mode forward couples the samples: one image halves the float64 oracle's CPU time
    m = _model(dev, 0.0, 0.3).eval()
    x = param_fill.make_input(n, 256, 7)
    sd = ou.oracle_state(0.0, 0.3, dtype=torch.float64, requires_grad=True)

    def scalar(o):
        ld = o["loss_dict"]
        return (o["cls_out"] * o["cls_out"]).sum() + 10.0 * (o["rec"] * o["rec"]).mean() + ld["freq_mask"].mean() \
            + ld["spat_mask"].mean() + sum((f * f).mean() for f in ld["triplet"])
    ref = eb4.forward_eb4(sd, x.double(), training=False)
    scalar(ref).backward()
    out = m(x.to(dev))
    assert out["cls_out"].requires_grad
    scalar(out).backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    worst, bad, seen = 0.0, [], 0
    for k, v in sd.items():
        if not v.requires_grad or v.grad is None or k not in params:
            continue
        g, r = params[k].grad, v.grad
        if g is None:
            bad.append((k, "no gradient"))
            continue
        seen += 1
        rn = float(r.abs().max())
        e = float((g.double().cpu() - r).abs().max()) / (rn + 1e-30)
        if rn > 1e-6:
            worst = max(worst, e)
            if e > GRAD_RTOL and float((g.double().cpu() - r).abs().max()) > GRAD_ATOL:
                bad.append((k, e))
    assert seen >= 480 and not bad, bad[:10]
    assert within("eval-mode backward: worst gradient tensor, max|d| / max|ref|", worst, 1e-2)


def _variant_model(dev, bias, affine):
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5, bias=bias, affine=affine)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    return m.to(dev)


@pytest.mark.parametrize("tag,bias,affine", [("bias_noaffine", True, False), ("bias", True, True), ("noaffine", False, False)])
def test_constructor_variants_eval_vs_reference_golden(golden_dir, tag, bias, affine):
    """The constructor variants no shipped YAML uses (model/unidefense.py:36-38: bias=True on the decoder / filter convs,
    affine=False on their InstanceNorms / BatchNorms) against vectors recorded from the REFERENCE built the same way
    (oracle/make_golden_variants.py, which also pins the oracle's handling of the two flags); state-dict keys as the reference's."""
    dev = _dev()
    g = np.load(os.path.join(golden_dir, f"udeb4_eval_n2_{tag}.npz"))
    n, size, seed = [int(v) for v in g["meta"]]
    m = _variant_model(dev, bias, affine).eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == eb4.eb4_state_shapes(2, bias=bias, affine=affine)
    x = param_fill.make_input(n, size, seed).to(dev)
    with torch.no_grad():
        out = m(x)
    _check_outputs(out, g)


def test_constructor_variant_train_vs_reference_golden(golden_dir):
    """bias=True + affine=False, train-mode forward + smooth pass-1 loss + backward at N = 2: outputs, losses and every
    parameter gradient (norm + first 8 elements; the conv biases among them) against the reference's — the bar of the default
    model's golden test."""
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "udeb4_train_n2_bias_noaffine.npz"))
    n, size, seed, mseed = [int(v) for v in g["meta"]]
    m = _variant_model(dev, True, False).train()
    x = param_fill.make_input(n, size, seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    out = m(x, rng=ou.make_rng(n, mseed, 0.5))
    _check_outputs(out, g)
    ls = _pass1_loss(out, tgt, ou.SMOOTH_LAMBDAS)
    for k, v in ls.items():
        _, e = _close(v, g["smooth_loss_" + k], k)
        assert within("loss " + k, e, RTOL), (k, e)
    ls["total_loss"].backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(m.named_parameters())
    assert sum(1 for k in names if k.endswith((".0.bias", ".3.bias", ".6.bias", ".9.bias"))) >= 13          # the conv biases
    rows = []
    for i, k in enumerate(names):
        gr = params[k].grad
        assert gr is not None, k
        ref_norm = float(g["smooth_grad_norms"][i])
        err = abs(gr.double().norm().item() - ref_norm)
        head = gr.flatten()[:8].cpu().numpy()
        herr = float(np.abs(head - g["smooth_grad_heads"][i][: head.size]).max())
        rows.append((max(err, herr) / (ref_norm + GRAD_ATOL / GRAD_RTOL), k, err, herr, ref_norm))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("  rel %.3e  %-58s norm err %.3e head err %.3e ref norm %.3e" % r)
    # Batch statistics over a batch of 2: the reference's own fp32 run is off by up to ~3e-3 on the scalar sf_coef gradients
    # (global sums with heavy cancellation; see the header) and a golden cannot supply the oracle's fp32-vs-fp64 yardstick —
    # so the sf_coef scalars are held to 1e-2 here and every other tensor (the conv biases among them) to 3e-3.
    scal = [r for r in rows if r[1].endswith("sf_coef")]
    rest = [r for r in rows if not r[1].endswith("sf_coef")]
    assert within("variant model, worst sf_coef gradient", scal[0][0], 1e-2)
    assert within("variant model, worst other gradient tensor: max(norm err, head err) / (ref norm + floor)", rest[0][0], 3e-3)
    assert len(rows) == len(names) >= 490


def _rng_sized(n, seed, drop_rate, size, nblk=32, dc_rate=0.2):
    """oracle/make_golden_variants.py:make_rng_sized (keep-masks that follow the feature maps of any input size)"""
    g = torch.Generator().manual_seed(seed)

    def bern(shape, keep):
        return (torch.rand(shape, generator=g) < keep).float()
    s4, s5 = -(-size // 16), -(-size // 32)
    rng = {"dec_keep": bern((n, 160, s4, s4), 0.8), "emb_keep": bern((n, 272, s5, s5), 1.0 - drop_rate),
           "feat_keep": bern((n, 1792), 1.0 - drop_rate), "drop_connect": {}}
    for idx in range(1, nblk):
        rng["drop_connect"][idx] = bern((n,), 1.0 - dc_rate * idx / nblk)
    return rng


def test_reference_batch_of_20_at_380(golden_dir):
    """The batch every shipped Eb4 YAML trains with: 10 real + 10 fake per GPU at 380 x 380 (config_template/uniatt/Prot1/
    data_ffpp.yml:71-72).  (1) Eval mode couples no samples: the two images of the reference golden udeb4_eval_n2_s380.npz, embedded
    in a batch of 20, must come out as the reference computed them (the bs-20 GEMM plans, the 95 x 95 block's DFT-matrix path and
    csrc/pool.hip's 95 -> 48 adaptive pool on real shapes).  (2) A train-mode forward + pass-1 loss + backward at 10 + 10: finite,
    every trainable parameter receives a gradient, and the backward is linear in the loss (x 4 -> gradients x 4)."""
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "udeb4_eval_n2_s380.npz"))
    n, size, seed = [int(v) for v in g["meta"]]
    x2 = param_fill.make_input(n, size, seed)
    x = torch.cat([x2, param_fill.make_input(18, size, seed + 1)], 0).to(dev)
    m = _model(dev, 0.0, 0.3).eval()
    with torch.no_grad():
        out = m(x)
    ld = out["loss_dict"]
    first = {"cls_out": out["cls_out"][:n], "rec": out["rec"][:n],
             "loss_dict": {"factorization": ld["factorization"][:n], "freq_mask": ld["freq_mask"][:n], "spat_mask": ld["spat_mask"][:n],
                           "spatial": ld["spatial"][:n], "freq": ld["freq"][:n], "triplet": [t[:n] for t in ld["triplet"]]}}
    _check_outputs(first, g)
    m.train()
    tgt = param_fill.make_labels(20).to(dev)
    rng = _rng_sized(20, 77, 0.5, size)
    grads = []
    for scale in (1.0, 4.0):
        for p_ in m.parameters():
            p_.grad = None
        o = m(x, rng=rng)
        ls = _pass1_loss(o, tgt, ou.SMOOTH_LAMBDAS)
        assert torch.isfinite(ls["total_loss"]).item()
        (ls["total_loss"] * scale).backward()
        grads.append({k: p_.grad.detach().double() for k, p_ in m.named_parameters() if p_.grad is not None})
    assert len(grads[0]) == 504 and all(torch.isfinite(v).all().item() for v in grads[0].values())
    gmax = max(float(v.abs().max()) for v in grads[0].values())
    worst = max(float((grads[1][k] - 4.0 * grads[0][k]).abs().max()) / (4.0 * float(grads[0][k].abs().max()) + 1e-3 * gmax)
                for k in grads[0])
    assert within("380 x 380, bs 10 + 10: backward linear in the loss (x 4), worst tensor", worst, 1e-3)


def test_native_resolution_380_vs_reference_golden(golden_dir):
    """UniDefenseModelEb4 at the reference's own resolution (config_template/uniatt/Prot1/data_ffpp.yml:71-72 and every other
    shipped Eb4 YAML: 380 x 380).  Feature maps 190 / 95 / 48 / 24 / 12: the 3 * 2^k sides run on the in-register mixed-radix
    FFT kernels inside the fused MBConv node, the one SF block on the 95 x 95 map (stride 2, pooled 95 -> 48 by overlapping
    windows) on the operator path with DFT matrices on the GEMM kernels.  Eval outputs, and a train step at N = 2 — outputs,
    losses and all 504 parameter gradients — against vectors recorded from the reference (oracle/make_golden_variants.py 380)."""
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "udeb4_eval_n2_s380.npz"))
    n, size, seed = [int(v) for v in g["meta"]]
    assert size == 380
    m = _model(dev, 0.0, 0.3).eval()
    with torch.no_grad():
        _check_outputs(m(param_fill.make_input(n, size, seed).to(dev)), g)
    g = np.load(os.path.join(golden_dir, "udeb4_train_n2_s380.npz"))
    n, size, seed, mseed = [int(v) for v in g["meta"]]
    m.train()
    x = param_fill.make_input(n, size, seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    out = m(x, rng=_rng_sized(n, mseed, 0.5, size))
    _check_outputs(out, g)
    ls = _pass1_loss(out, tgt, ou.SMOOTH_LAMBDAS)
    for k, v in ls.items():
        _, e = _close(v, g["smooth_loss_" + k], k)
        assert within("380: loss " + k, e, RTOL), (k, e)
    ls["total_loss"].backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(m.named_parameters())
    rows = []
    for i, k in enumerate(names):
        gr = params[k].grad
        assert gr is not None, k
        ref_norm = float(g["smooth_grad_norms"][i])
        err = abs(gr.double().norm().item() - ref_norm)
        head = gr.flatten()[:8].cpu().numpy()
        herr = float(np.abs(head - g["smooth_grad_heads"][i][: head.size]).max())
        rows.append((max(err, herr) / (ref_norm + GRAD_ATOL / GRAD_RTOL), k, err, herr, ref_norm))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("  rel %.3e  %-58s norm err %.3e head err %.3e ref norm %.3e" % r)
    # N = 2: as in test_constructor_variant_train_vs_reference_golden, the sf_coef scalars are held to 1e-2, the rest to 3e-3
    # ... and the BN2 biases whose gradient is analytically zero (STRUCT_ZERO_GRADS: rounding noise on both sides,
    # reference norms 1e-5 .. 1e-4 here) to an absolute 2e-4
    zero = STRUCT_ZERO_GRADS
    scal = [r for r in rows if r[1].endswith("sf_coef")]
    noise = [r for r in rows if r[1] in zero and r[4] < 1e-3]          # (blocks 0 and 31 of that list carry real gradients)
    rest = [r for r in rows if not r[1].endswith("sf_coef") and r not in noise]
    assert within("380: worst sf_coef gradient", scal[0][0], 1e-2)
    assert within("380: analytically-zero BN2 bias gradients, absolute error", max(max(r[2], r[3]) for r in noise), 2e-4)
    assert within("380: worst other gradient tensor: max(norm err, head err) / (ref norm + floor)", rest[0][0], 3e-3)
    assert len(rows) == 504
