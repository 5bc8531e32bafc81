"""CPU: engine/metrics.py against the libraries the reference's cal_metrics calls (utils/statistic.py:33-74: sklearn
roc_curve / auc / confusion_matrix, scipy brentq + interp1d), and the cross-rank score gather over gloo (world size 2,
unequal counts)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unidefense_amd.engine.metrics import cal_metrics, gather_scores, roc_points


def _library_metrics(y, p, threshold):
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import auc, confusion_matrix, roc_curve
    fpr, tpr, thr = roc_curve(y, p, pos_label=0, drop_intermediate=False)
    out = {"AUC": auc(fpr, tpr), "EER": brentq(lambda x: 1. - x - interp1d(fpr, tpr)(x), 0., 1.)}
    pred = 1 - (np.array(p) > threshold).astype(int)
    (tp, fn), (fp, tn) = confusion_matrix(y, pred)
    out.update(ACC=(tp + tn) / len(y), APCER=fp / (tn + fp), BPCER=fn / (fn + tp), NumP=tp + fn, NumN=tn + fp)
    return out, (fpr, tpr)


@pytest.mark.parametrize("seed,n,sep", [(0, 400, 1.0), (1, 37, 0.3), (2, 2000, 2.5), (3, 64, 0.0)])
def test_cal_metrics_matches_sklearn_scipy(seed, n, sep):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, 2, n)
    y[:2] = (0, 1)
    logits = rng.normal(size=n) + sep * (y == 0)
    p = 1.0 / (1.0 + np.exp(-logits))
    p = np.round(p, 2) if seed == 1 else p                       # ties
    got = cal_metrics(y, p, threshold=0.5)
    ref, (fpr, tpr) = _library_metrics(y, p, 0.5)
    f2, t2, _ = roc_points(y, p, pos_label=0)
    assert np.allclose(f2, fpr) and np.allclose(t2, tpr)
    for k in ("AUC", "ACC", "APCER", "BPCER", "NumP", "NumN"):
        assert got[k] == pytest.approx(ref[k], abs=1e-12), k
    assert got["EER"] == pytest.approx(ref["EER"], abs=1e-9)
    assert got["ACER"] == pytest.approx(0.5 * (ref["APCER"] + ref["BPCER"]), abs=1e-12)
    assert 0.0 <= got["TPR1%"] <= got["TPR5%"] <= 1.0


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 5 + 3 * rank
    s = torch.arange(n, dtype=torch.float32) + 100 * rank
    l = torch.full((n,), rank, dtype=torch.int64)
    gs, gl = gather_scores(s, l)
    q.put((rank, gs.tolist(), gl.tolist()))
    dist.destroy_process_group()


def test_gather_scores_world2_unequal_counts():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    want_s = list(range(5)) + [100 + i for i in range(8)]
    want_l = [0] * 5 + [1] * 8
    for _, gs, gl in res:
        assert gs == want_s and gl == want_l
