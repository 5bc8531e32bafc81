"""The scalar tail of a pass's loss as ONE autograd node over two HIP launches (csrc/loss.hip, `ud_loss_tail_run`).

Reference: `AbstractEngine.train_unidefense_model`, engine/abstract_engine.py:232-270 (pass 1) and :294-371 (pass 2): the
`softmax` criterion on `cls_out`, the means of the two attention masks, the asymmetrical weighted triplet terms of up to three
features, the means of the per-sample reconstruction / frequency terms over the real (and, for logging, the fake) samples, and
their weighted sum.  As torch ops that is ~45 launches of 4-5 microseconds between the forward and the backward of every pass
(0.2 ms of a 24 ms step); here: two launches forward, one multiply backward.

`pass_tail(...)` returns None when the fused form does not apply (CPU tensors, a criterion other than the plain
CrossEntropyLoss / this package's triplet loss, one-logit heads, unknown batch layout) — the caller then takes the torch road,
which stays the definition.
"""
import torch
import torch.nn as nn

from .triplet_loss import AsymmetricalWeightedTripletLoss

_KEYS = ("total", "cls", "triplet", "real_rec", "fake_rec", "real_freq", "fake_freq", "freq_mask", "spat_mask")


class _PassTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n_real, n_fake, weights, tgt, nfeat, cls_out, fm, sm, spatial, freq, *feats):
        from .. import kernels as K

        def c(t):
            return None if t is None else t.detach().contiguous()
        vals, grads = K.loss_tail(c(cls_out), tgt.contiguous(), n_real, n_fake, [c(f) for f in feats], c(fm), c(sm), c(spatial),
                                  c(freq), weights)
        ctx.grads = grads
        ctx.shapes = [None if t is None else t.shape for t in (cls_out, fm, sm, spatial, freq)]
        outs = vals.unbind(0)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g, *_):
        grads = ctx.grads
        flat = grads["_flat"] * g                      # every gradient of the tail scaled by the incoming one: one launch
        base = grads["_flat"].data_ptr()

        def view(k, shape=None):
            t = grads.get(k)
            if t is None:
                return None
            off = (t.data_ptr() - base) // 4
            return flat[off:off + t.numel()].view(t.shape if shape is None else shape)
        sh = ctx.shapes
        nfeat = sum(1 for k in grads if k.startswith("feat"))
        return (None, None, None, None, None, view("cls", sh[0]), view("fm", sh[1]), view("sm", sh[2]), view("spatial", sh[3]),
                view("freq", sh[4])) + tuple(view(f"feat{i}") for i in range(nfeat))


def pass_tail(out_dict, tgt, n_real, n_fake, criteria, weights, masks=True):
    """weights: dict(cls, mask, triplet, rec, freq) — the factors of the pass (pass 1: 1, lambda_mask, lambda_triplet,
    lambda_recons, lambda_freq).  masks = False leaves the two mask terms out (pass 2's KL branch adds them itself).
    Returns {name: 0-dim tensor} with `total` differentiable, or None when the fused form does not apply."""
    cls_out = out_dict["cls_out"]
    ld = out_dict.get("loss_dict", {})
    feats = ld.get("triplet")
    fm, sm = (ld.get("freq_mask"), ld.get("spat_mask")) if masks else (None, None)
    spatial, freq = ld.get("spatial"), ld.get("freq")
    ce, trip = criteria.get("softmax"), criteria.get("triplet")
    N = cls_out.shape[0]
    ok = (cls_out.is_cuda and cls_out.dtype == torch.float32 and cls_out.dim() == 2 and 2 <= cls_out.shape[-1] <= 64
          and type(ce) is nn.CrossEntropyLoss and ce.weight is None and ce.ignore_index == -100 and ce.reduction == "mean"
          and ce.label_smoothing == 0.0 and tgt.dtype == torch.int64 and tgt.shape == (N,)
          and n_real is not None and n_fake is not None and 0 < n_real < N and n_real + n_fake <= N
          and (feats is None or (type(trip) is AsymmetricalWeightedTripletLoss and len(feats) <= 3
                                 and all(f.dim() == 2 and f.shape[0] == N and f.dtype == torch.float32 for f in feats)))
          and all(t is None or (t.dtype == torch.float32 and t.is_cuda) for t in (fm, sm, spatial, freq))
          and all(t is None or t.shape == (N,) for t in (spatial, freq)))
    if not ok:
        return None
    feats = list(feats) if feats is not None else []
    w = (weights["cls"], weights["mask"], weights["mask"], weights["triplet"], weights["rec"], weights["freq"])
    outs = _PassTail.apply(int(n_real), int(n_fake), w, tgt, len(feats), cls_out, fm, sm, spatial, freq, *feats)
    return dict(zip(_KEYS, outs))
