"""GPU, TWO ranks on ONE device: the data-parallel semantics of the full HIP model at world size 2.

The 1-GPU box cannot run RCCL with two ranks (NCCL refuses two ranks on one device), but the gloo backend moves CUDA
tensors through the host — slow, and exactly the same torch.distributed calls the RCCL path issues (broadcast,
all_gather_into_tensor, all_reduce incl. async_op).  So two processes, both on cuda:0, each with HALF of a batch,
wrapped in HipDataParallel (SyncBN + streamed gradient buckets), must reproduce what ONE process computes on the
whole batch with plain BatchNorm:
    * SyncBN statistics over both shards == full-batch statistics  -> same forward outputs per sample;
    * mean of the ranks' gradients of their local mean-losses == gradient of the full-batch mean loss.
The loss leaves the AW-triplet terms out (pairwise over the batch: not decomposable over ranks — the reference's DDP run
has the same property) and orders each shard [real; fake] like the reference's two samplers (forgery_engine.py:67-86).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N_RANK, SIZE = 4, 256


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


MODELS = {"UDEB4": (256, dict(extractor="efficientnet-b4", drop_connect_rate=0.0)),
          # BASELINE configs[3]: UDR50 at 320 x 320 — every BatchNorm on the operator path (engine/ocim_engine.py:130-133)
          "UDR50": (320, dict(extractor="resnet50"))}


def _build(dev, name="UDEB4"):
    from oracle import param_fill
    from unidefense_amd.model import load_model
    m = load_model(name)(num_classes=2, drop_rate=0.0, **MODELS[name][1])
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dev).train()
    m._dec_dropout = False
    return m


def _loss(out, tgt):
    from tests import oracle_util as ou
    lam = dict(ou.LAMBDAS)
    ld = out["loss_dict"]
    n_real = tgt.numel() // 2
    cls = torch.nn.functional.cross_entropy(out["cls_out"], tgt)
    return cls + lam["lambda_mask"] * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) + \
        lam["lambda_recons"] * ld["spatial"].narrow(0, 0, n_real).mean() + \
        lam["lambda_freq"] * ld["freq"].narrow(0, 0, n_real).mean()


def _digest(g):
    """What travels between processes for a gradient: its first 4096 entries, its L2 norm and its max."""
    g = g.detach()
    return g.reshape(-1)[:4096].cpu().numpy(), g.double().norm().item(), g.abs().max().item()     # numpy: pickled by value


def _shards(name="UDEB4"):
    """Two [real, real, fake, fake] shards and the equivalent full batch [4 real; 4 fake]."""
    from oracle import param_fill
    x = param_fill.make_input(2 * N_RANK, MODELS[name][0], seed=123)                 # rows 0-3 real, 4-7 fake
    h = N_RANK // 2
    idx = [list(range(r * h, (r + 1) * h)) + list(range(N_RANK + r * h, N_RANK + (r + 1) * h)) for r in range(2)]
    return x, idx


def _worker(rank, port, q, exchange=True, name="UDEB4"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UD_SYNCBN_EXCHANGE="1" if exchange else "0")
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        from unidefense_amd.config import cfg
        from unidefense_amd.engine.parallel import HipDataParallel
        cfg.syncbn_exchange = bool(exchange)
        m = _build(dev, name)
        dp = HipDataParallel(m, bucket_bytes=32 << 20)
        x, idx = _shards(name)
        xs = x[idx[rank]].contiguous().to(dev)
        tgt = torch.tensor([0] * (N_RANK // 2) + [1] * (N_RANK // 2), device=dev)
        res = None
        for _ in range(2):                     # 1st backward learns the use counts (reduce at the end), 2nd streams
            for p in m.parameters():
                p.grad = None
            out = dp(xs)
            _loss(out, tgt).backward()
            res = ({k: _digest(p.grad) for k, p in m.named_parameters() if p.grad is not None},
                   out["cls_out"].detach().cpu().numpy(), out["rec"].detach().cpu().numpy())
        xok = dp.bn_exchange.ok
        if xok:
            dp.bn_exchange.check()
        q.put((rank, idx[rank], res + (xok,), None))
    except Exception as e:                      # noqa: BLE001
        import traceback
        q.put((rank, None, None, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("name,exchange", [("UDEB4", True), ("UDEB4", False), ("UDR50", True)],
                         ids=["UDEB4-peer-exchange", "UDEB4-all_reduce", "UDR50-320-peer-exchange"])
def test_two_ranks_equal_one_process_full_batch(name, exchange):
    """exchange: the SyncBN sums travel through BnExchange's peer-mapped mailboxes (csrc/xchg.hip: HIP IPC between the two
    processes, one kernel per sum) — it must have passed its self-test and been used; False: dist.all_reduce.
    UDEB4: the fused MBConv path + the operator-path BatchNorms of attention / head; UDR50 at 320 x 320 (BASELINE configs[3]):
    every BatchNorm on the operator path (tape._syncbn_act), mailbox rows sized for its 2048-channel norms."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.margins import within
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, exchange, name)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(120)
    for rank, _, _, err in got:
        assert err is None, f"rank {rank}:\n{err}"
    got.sort(key=lambda t: t[0])
    assert [g[2][3] for g in got] == [exchange, exchange], "BnExchange state"
    # single process, whole batch, plain BatchNorm
    dev = torch.device("cuda:0")
    m = _build(dev, name)
    x, idx = _shards(name)
    tgt = torch.tensor([0] * N_RANK + [1] * N_RANK, device=dev)
    out = m(x.to(dev))
    _loss(out, tgt).backward()
    ref_g = {k: _digest(p.grad) for k, p in m.named_parameters() if p.grad is not None}
    # forward: every sample's outputs agree (SyncBN == full-batch BN; the batch statistics are summed in two halves, so the
    # head's logits move by fp32 rounding through the whole trunk: 1e-5 (UDEB4) ... 6e-5 (UDR50 at 320) of their scale)
    for rank, ids, (g, cls, rec, _xok), _ in got:
        e1 = abs(cls - out["cls_out"].detach().cpu().numpy()[ids]).max() / out["cls_out"].abs().max().item()
        e2 = abs(rec - out["rec"].detach().cpu().numpy()[ids]).max() / out["rec"].abs().max().item()
        print(f"  rank {rank}: cls_out {e1:.2e}  rec {e2:.2e}")
        assert within(f"rank {rank} cls_out vs full batch", e1, 3e-4) and within(f"rank {rank} rec vs full batch", e2, 1e-3), (rank, e1, e2)
    # both ranks hold the same averaged gradients, equal to the full-batch gradients
    g0, g1 = got[0][2][0], got[1][2][0]
    assert set(g0) == set(ref_g) == set(g1)
    gmax = max(v[2] for v in ref_g.values())
    import numpy as np
    # the scalar gates (sf_coef, fuse_coef: one global sum each, cancelling to a layer-dependent degree) against the common
    # scale of those sums, like tests/test_c_r50.py
    gate_scale = max([abs(float(h[0])) for h, _, _ in ref_g.values() if h.size == 1] + [1e-30])
    rows, worst_rr = [], 0.0
    for k, (head, norm, mx) in ref_g.items():
        scale = mx + 3e-3 * gmax                     # zero-true-gradient tensors hold rounding noise (test_y_fullsize_gpu.py)
        worst_rr = max(worst_rr, float(abs(g0[k][0] - g1[k][0]).max()) / scale)
        if head.size == 1:
            rows.append(("gate", abs(float(g0[k][0][0] - head[0])) / max(abs(float(head[0])), 0.2 * gate_scale), k))
            continue
        rows.append(("entry", float(abs(g0[k][0] - head).max()) / scale, k))
        rows.append(("norm", abs(g0[k][1] - norm) / (norm + 3e-3 * gmax * head.size ** 0.5), k))
        rows.append(("l2", float(np.linalg.norm(g0[k][0] - head)) / (float(np.linalg.norm(head)) + 3e-3 * gmax * head.size ** 0.5), k))
    worst = {kind: max((r for r in rows if r[0] == kind), key=lambda r: r[1]) for kind in ("gate", "entry", "norm", "l2")}
    for kind in worst:
        top = sorted((r for r in rows if r[0] == kind), key=lambda r: -r[1])[:4]
        print(f"  {kind:5s} worst:", ", ".join(f"{v:.2e} {k}" for _, v, k in top))
    print(f"  {len(ref_g)} gradients; rank 0 vs rank 1 {worst_rr:.2e}")
    ok = [within("rank 0 vs rank 1 gradients", worst_rr, 1e-6)]
    if name == "UDEB4":
        ok += [within("averaged gradient heads vs full batch", worst["entry"][1], 2e-3),
               within("averaged gradient norms vs full batch", worst["norm"][1], 2e-3),
               within("scalar gate gradients vs full batch / max(own, 0.2 x largest gate gradient)", worst["gate"][1], 2e-3)]
    else:
        # A ReLU network: the two shards run other GEMM plans than the full batch (other M), i.e. other rounding, and a
        # handful of the 1e8 ReLU units within that of zero land on the other side — single weight-gradient ENTRIES (and
        # the scalar gates, which are single entries) move by percents (tests/test_y_fullsize_gpu.py), each tensor as a
        # whole does not: relative L2 / norms per tensor, the gates against their common scale.
        ok += [within("averaged gradient heads vs full batch, relative L2 per tensor", worst["l2"][1], 5e-2),
               within("averaged gradient norms vs full batch", worst["norm"][1], 2e-2),
               within("scalar gate gradients vs full batch / max(own, 0.2 x largest gate gradient)", worst["gate"][1], 1e-1)]
    assert all(ok)


def _xchg_worker(rank, port, q, world=2, skew=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        import time
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from unidefense_amd.engine.parallel import BnExchange
        ex = BnExchange(dist.group.WORLD, dev)
        assert ex.ok, "self-test failed"
        g = torch.Generator().manual_seed(7)
        vecs = [torch.randn(world, n, generator=g, dtype=torch.float64) for n in (8, 96, 1632 * 2, 3264 * 2, 5, 8192)]
        worst = 0.0
        for rep in range(3):
            for i, v in enumerate(vecs):
                if skew and (i + rep + rank) % world == 0:
                    # one rank (a different one each time) arrives late: the others spin in the kernel on its tagged row while
                    # their own later exchanges must not overwrite a slot it has not read yet (SLOTS = 4 in flight)
                    torch.cuda.synchronize()
                    time.sleep(0.05)
                a = (v[rank] * (rep + 1)).to(dev)
                if (i + rep) % 2:
                    own = torch.full_like(a, float("nan"))
                    ex.allreduce(a, own)                                      # + this rank's own values, written by the same launch
                    assert torch.equal(own.cpu(), v[rank] * (rep + 1)), "local_out"
                else:
                    ex.allreduce(a)
                want = v[0] * (rep + 1)
                for r in range(1, world):                                     # rank order 0 + 1 + ...: exactly this fp64 sum
                    want = want + v[r] * (rep + 1)
                worst = max(worst, float((a - want.to(dev)).abs().max()))
        # a captured sequence replays with the device-side sequence counter still advancing
        a = torch.zeros(4096, dtype=torch.float64, device=dev)
        src = torch.full((4096,), float(rank + 1), dtype=torch.float64, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                a.copy_(src)
                ex.allreduce(a)
                a.mul_(0.5)
                ex.allreduce(a)
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        ex.check()
        tot = world * (world + 1) / 2.0                 # world 2: (1 + 2) -> 1.5 on both -> 3.0
        replay_ok = bool((a == tot * 0.5 * world).all())
        ex.close()
        q.put((rank, worst, replay_ok, None))
    except Exception:                      # noqa: BLE001
        import traceback
        q.put((rank, None, None, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_bn_exchange_two_processes():
    """BnExchange alone: two processes on the one GPU, mailboxes mapped into each other through HIP IPC; sums of several
    lengths are exact (rank-ordered fp64 adds), eager and from a replayed hipGraph."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
    for rank, worst, replay_ok, err in got:
        assert err is None, f"rank {rank}:\n{err}"
        assert worst == 0.0 and replay_ok, (rank, worst, replay_ok)


def test_bn_exchange_four_processes_with_skewed_arrivals():
    """Four mailboxes on the one GPU (what a 4-GPU node maps, BASELINE configs[3]), ranks arriving up to 50 ms apart in turn:
    every sum exact in rank order, slots reused 18 times each without a reader losing its row, a captured pair replayed."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, port, q, 4, True)) for r in range(4)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(4)]
    for p in procs:
        p.join(120)
    for rank, worst, replay_ok, err in got:
        assert err is None, f"rank {rank}:\n{err}"
        assert worst == 0.0 and replay_ok, (rank, worst, replay_ok)
