"""Evaluation metrics and the cross-rank score gather of the test / validate stage.

The reference scores every frame with p(real) = softmax(cls_out)[:, 0] (engine/forgery_engine.py:349,437), gathers the
per-rank dictionaries with ``dist.all_gather_object`` (pickling through the CPU, :374-375) and hands the flat lists to
``utils/statistic.py:cal_metrics`` (sklearn ``roc_curve`` with the REAL class as the positive one, scipy ``brentq`` for the
equal-error rate).  Here: ``gather_scores`` exchanges two padded device tensors with one ``all_gather_into_tensor`` each
(RCCL on the GPU), and ``cal_metrics`` is plain numpy written from the definitions (tests/test_metrics_cpu.py checks it
against sklearn / scipy, which this image happens to have)."""
import numpy as np
import torch
import torch.distributed as dist


def gather_scores(scores, labels, group=None):
    """scores [n] float, labels [n] int on any device -> (all scores, all labels) of every rank, in rank order.
    Ranks may hold different counts: counts are exchanged first, the payload is padded to the largest."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return scores, labels
    world = dist.get_world_size(group)
    n = torch.tensor([scores.numel()], device=scores.device, dtype=torch.int64)
    counts = torch.empty(world, device=scores.device, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, n, group=group)
    cap = int(counts.max().item())
    pay = torch.zeros(2, cap, device=scores.device, dtype=torch.float32)
    pay[0, :scores.numel()] = scores.float()
    pay[1, :labels.numel()] = labels.float()
    out = torch.empty(world, 2, cap, device=scores.device, dtype=torch.float32)
    dist.all_gather_into_tensor(out.view(-1), pay.view(-1), group=group)
    keep = torch.arange(cap, device=scores.device)[None, :] < counts[:, None]
    return out[:, 0][keep], out[:, 1][keep].long()


def roc_points(y_true, score, pos_label=0):
    """(fpr, tpr, thresholds) at every distinct score, thresholds decreasing, with the (0, 0) start point
    (threshold = inf): the curve sklearn.metrics.roc_curve(drop_intermediate=False) describes."""
    y = (np.asarray(y_true) == pos_label).astype(np.float64)
    s = np.asarray(score, dtype=np.float64)
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    last = np.r_[np.nonzero(np.diff(s))[0], s.size - 1]          # last index of every run of equal scores
    tps = np.cumsum(y)[last]
    fps = (last + 1) - tps
    tpr = np.r_[0.0, tps / max(tps[-1], 1.0)]
    fpr = np.r_[0.0, fps / max(fps[-1], 1.0)]
    return fpr, tpr, np.r_[np.inf, s[last]]


def _interp_root(fpr, tpr):
    """x in [0, 1] with 1 - x = tpr(x) on the piecewise-linear ROC curve (the equal-error rate)."""
    g = 1.0 - fpr - tpr                                          # decreasing from 1 to -1 along the curve
    i = int(np.argmax(g <= 0.0))
    if i == 0:
        return float(fpr[0])
    x0, x1, g0, g1 = fpr[i - 1], fpr[i], g[i - 1], g[i]
    if x1 == x0:                                                 # vertical segment: the crossing is at this fpr
        return float(x0)
    # on the segment tpr is linear in x: g(x) = g0 + (g1 - g0) (x - x0) / (x1 - x0)
    return float(x0 + (x1 - x0) * g0 / (g0 - g1))


def _tpr_at_fpr(tpr, fpr, value):
    """TPR at the operating point whose FPR is nearest `value` (the last one among equals), as the reference picks it."""
    target = fpr[int(np.argmin(np.abs(fpr - value)))]
    return float(tpr[int(np.max(np.nonzero(fpr == target)[0]))])


def cal_metrics(y_trues, y_preds, threshold=0.5):
    """Keys of the reference's cal_metrics (utils/statistic.py:33-74): AUC, EER, Thre, ACC, TP_Ratio, NumP, TN_Ratio,
    NumN, APCER, BPCER, ACER, TPR1%, TPR5%.  Labels: 0 = real (the positive class), 1 = fake; y_preds = p(real).
    threshold: a number, or 'auto' (the EER threshold)."""
    y = np.asarray(y_trues).astype(np.int64)
    p = np.asarray(y_preds, dtype=np.float64)
    fpr, tpr, thr = roc_points(y, p, pos_label=0)
    m = {"AUC": float(np.trapezoid(tpr, fpr))}
    m["EER"] = _interp_root(fpr, tpr)
    finite = thr.copy()
    finite[0] = thr[1] + 1.0 if thr.size > 1 else 1.0            # sklearn's first threshold is max(score) + 1 / inf
    m["Thre"] = float(np.interp(m["EER"], fpr, finite))
    if threshold == "auto":
        threshold = m["Thre"]
    else:
        m["Thre"] = float(threshold)
    pred = 1 - (p > threshold).astype(np.int64)                  # p(real) above the threshold -> predicted real (0)
    tp = int(np.sum((y == 0) & (pred == 0)))
    fn = int(np.sum((y == 0) & (pred == 1)))
    fp = int(np.sum((y == 1) & (pred == 0)))
    tn = int(np.sum((y == 1) & (pred == 1)))
    m["ACC"] = (tp + tn) / max(y.size, 1)
    m["TP_Ratio"] = tp / max(tp + fn, 1)
    m["NumP"] = tp + fn
    m["TN_Ratio"] = tn / max(tn + fp, 1)
    m["NumN"] = tn + fp
    m["APCER"] = fp / max(tn + fp, 1)
    m["BPCER"] = fn / max(fn + tp, 1)
    m["ACER"] = 0.5 * (m["APCER"] + m["BPCER"])
    m["TPR1%"] = _tpr_at_fpr(tpr, fpr, 0.01)
    m["TPR5%"] = _tpr_at_fpr(tpr, fpr, 0.05)
    return m
