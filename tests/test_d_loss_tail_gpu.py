"""The scalar tail of a pass's loss as two HIP launches (loss/pass_tail.py, csrc/loss.hip `ud_loss_tail_run`) against the torch
formulation of the same terms in float64 — the reference's own ops (engine/abstract_engine.py:232-270: CrossEntropyLoss, mask
means, AsymmetricalWeightedTripletLoss per feature, real / fake means of the per-sample terms, the weighted sum): every
returned scalar and every input gradient, with and without the optional terms, ragged feature widths, unequal real / fake
counts, and a scaled incoming gradient (the GradScaler's)."""
import pytest
import torch

from tests.margins import within

pytestmark = pytest.mark.gpu


def _torch_tail(cls_out, tgt, feats, fm, sm, spatial, freq, R, F, w):
    import torch.nn as nn
    from unidefense_amd.loss.triplet_loss import AsymmetricalWeightedTripletLoss
    ce = nn.CrossEntropyLoss()(cls_out, tgt)
    crit = AsymmetricalWeightedTripletLoss()          # (double tensors: the torch formulation, not the HIP one)
    trip = sum(crit(f, tgt) for f in feats) if feats else cls_out.new_zeros(())
    z = cls_out.new_zeros(())
    fmm = fm.mean() if fm is not None else z
    smm = sm.mean() if sm is not None else z
    rr = spatial[:R].mean() if spatial is not None else z
    fr = spatial[R:R + F].mean() if spatial is not None else z
    rq = freq[:R].mean() if freq is not None else z
    fq = freq[R:R + F].mean() if freq is not None else z
    total = w["cls"] * ce + w["mask"] * (fmm + smm) + w["triplet"] * trip + w["rec"] * rr + w["freq"] * rq
    return dict(total=total, cls=ce, triplet=trip, real_rec=rr, fake_rec=fr, real_freq=rq, fake_freq=fq, freq_mask=fmm,
                spat_mask=smm)


@pytest.mark.parametrize("N,R,C,dims,masks,persample", [
    (32, 16, 2, (160, 80, 40), True, True),          # the bench's pass 1
    (20, 10, 2, (160, 80, 40), True, True),          # the reference YAMLs' batch
    (8, 4, 2, (448, 128), True, True),               # UDR18: two features
    (12, 5, 3, (33,), False, True),                  # unequal counts, ragged width, no masks
    (6, 2, 2, (), True, False),                      # no triplet / per-sample terms
])
def test_pass_tail_equals_the_torch_formulation(N, R, C, dims, masks, persample):
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.loss.pass_tail import pass_tail
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N * 100 + R)
    F = N - R
    mk = lambda *s: torch.randn(*s, generator=g)
    cls_out = (2 * mk(N, C)).to(dev).requires_grad_()
    tgt = torch.tensor([0] * R + [1] * F, device=dev)
    feats = [(0.5 * mk(N, d)).to(dev).requires_grad_() for d in dims]
    fm = torch.rand(N, 1, 8, 5, generator=g).to(dev).requires_grad_() if masks else None
    sm = torch.rand(N, 1, 8, 8, generator=g).to(dev).requires_grad_() if masks else None
    spatial = torch.rand(N, generator=g).to(dev).requires_grad_() if persample else None
    freq = torch.rand(N, generator=g).to(dev).requires_grad_() if persample else None
    w = dict(cls=1.0, mask=0.1, triplet=0.1, rec=0.1, freq=1.0)
    out = {"cls_out": cls_out, "loss_dict": {"triplet": feats if feats else None, "freq_mask": fm, "spat_mask": sm,
                                             "spatial": spatial, "freq": freq}}
    f = pass_tail(out, tgt, R, F, {"softmax": LOSSES["cross_entropy"], "triplet": LOSSES["aw_triplet"]}, w)
    assert f is not None
    scale = 1024.0                                     # an incoming gradient other than 1 (GradScaler(2**10))
    (f["total"] * scale).backward()
    leaves = [t for t in [cls_out, fm, sm, spatial, freq] + feats if t is not None]
    got = [t.grad.double().cpu() for t in leaves]

    d = lambda t: None if t is None else t.detach().double().cpu().requires_grad_()
    ref_in = [d(cls_out), d(fm), d(sm), d(spatial), d(freq)] + [d(x) for x in feats]
    ref = _torch_tail(ref_in[0], tgt.cpu(), ref_in[5:], ref_in[1], ref_in[2], ref_in[3], ref_in[4], R, F, w)
    (ref["total"] * scale).backward()
    want = [t.grad for t in ref_in if t is not None]
    for k, v in ref.items():
        fk, fv = float(f[k].detach()), float(v.detach())
        err = abs(fk - fv) / max(abs(fv), 1e-3)
        assert within(f"tail {k}", err, 2e-5), (k, fk, fv)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape
        err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
        assert within(f"tail grad {i}", err, 5e-5), (i, err)
    for k in f:
        if k != "total":
            assert not f[k].requires_grad


def test_pass_tail_declines_what_it_does_not_cover():
    """other criteria / layouts take the torch road (None), they are never approximated"""
    import torch.nn as nn
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.loss.pass_tail import pass_tail
    dev = torch.device("cuda:0")
    cls_out = torch.randn(8, 2, device=dev)
    tgt = torch.tensor([0] * 4 + [1] * 4, device=dev)
    out = {"cls_out": cls_out, "loss_dict": {"triplet": None, "freq_mask": None, "spat_mask": None, "spatial": None, "freq": None}}
    w = dict(cls=1.0, mask=0.1, triplet=0.1, rec=0.1, freq=1.0)
    crit = {"softmax": LOSSES["cross_entropy"], "triplet": LOSSES["aw_triplet"]}
    assert pass_tail(out, tgt, 4, 4, crit, w) is not None
    assert pass_tail(out, tgt, 4, 4, dict(crit, softmax=nn.CrossEntropyLoss(label_smoothing=0.1)), w) is None
    assert pass_tail(out, tgt, 4, 4, dict(crit, softmax=nn.BCEWithLogitsLoss()), w) is None
    assert pass_tail(out, tgt, None, None, crit, w) is None
    assert pass_tail({"cls_out": cls_out.cpu(), "loss_dict": out["loss_dict"]}, tgt.cpu(), 4, 4, crit, w) is None
    assert pass_tail({"cls_out": cls_out[:, :1], "loss_dict": out["loss_dict"]}, tgt, 4, 4, crit, w) is None
