"""CPU, gloo, world_size 2: the N>1 logic of the data-parallel path (engine/parallel.py, tape.sync_batch_stats)
— gradient bucketing/averaging, parameter broadcast, and the SyncBatchNorm statistics combine."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unidefense_amd.engine.parallel import HipDataParallel
        from unidefense_amd.tape import sync_batch_stats
        torch.manual_seed(rank)                       # different initial weights per rank
        net = nn.Sequential(nn.Linear(7, 5), nn.BatchNorm1d(5), nn.Linear(5, 3))
        dp = HipDataParallel(net, bucket_bytes=64)     # tiny buckets -> several flushes
        # 1. broadcast: every rank now holds rank 0's parameters
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, 0)
        ok_bcast = torch.equal(flat, ref)
        # 2. gradient averaging over ranks, None entries preserved
        params = list(net.parameters())
        grads = [torch.full_like(p, float(rank + 1)) * (i + 1) for i, p in enumerate(params)]
        grads[1] = None
        out = dp.sync_grads(grads)
        ok_avg = out[1] is None and all(
            torch.allclose(g, torch.full_like(g, (i + 1) * (1 + world) / 2.0)) for i, g in enumerate(out) if g is not None)
        # 3. SyncBN statistics: global mean/var of the concatenated shards
        g = torch.Generator().manual_seed(123)
        full = torch.randn(world * 6, 4, generator=g, dtype=torch.float64) * 3 + 1
        shard = full[rank * 6:(rank + 1) * 6]
        mean, var, invstd = sync_batch_stats(shard.mean(0, keepdim=True), shard.var(0, unbiased=False, keepdim=True),
                                             1e-5, None)
        ok_bn = torch.allclose(mean.view(-1), full.mean(0)) and torch.allclose(var.view(-1), full.var(0, unbiased=False)) \
            and torch.allclose(invstd.view(-1), torch.rsqrt(full.var(0, unbiased=False) + 1e-5))
        # 4. the streaming protocol of _NetFunction.backward: the tape hands each gradient to the reducer when its
        #    use count is reached (here 1 each), buckets flush mid-way (async all_reduce), finish() returns views;
        #    the caller pre-scales by 1/world so the SUM is the mean
        from unidefense_amd import tape as T
        tp = T.Tape()
        red = dp.module._grad_reducer
        red.begin()
        tp.param_uses = {id(p): 1 for p in params}
        tp.param_ready = red.ready
        locals_ = {id(p): torch.full_like(p, float(rank + 1) * (i + 1)) / world for i, p in enumerate(params)}
        for p in reversed(params):
            tp.add_param_grad(p, locals_[id(p)])
        out2 = red.finish()
        ok_stream = set(out2) == {id(p) for p in params} and tp.param_seen == tp.param_uses and all(
            torch.allclose(out2[id(p)], torch.full_like(p, (i + 1) * (1 + world) / 2.0)) for i, p in enumerate(params))
        ok_avg = ok_avg and ok_stream and dp.module._grad_prescale == 1.0 / world
        # 5. the fused path's SyncBN context (tape.DataParallelCtx): fp64 accumulators summed in place, the local copy kept
        #    for the affine gradients; sums that fit the mailbox go through the exchange object, larger ones through
        #    dist.all_reduce (here a stand-in exchange that counts its calls and reduces over gloo)
        dpc = T.DataParallelCtx(dist.group.WORLD)
        acc = torch.full((6,), float(rank + 1), dtype=torch.float64)
        loc = dpc.reduce(acc, keep_local=True)
        tri = world * (world + 1) / 2.0
        ok_ctx = dpc.world == world and dpc.synced and torch.equal(acc, torch.full((6,), tri, dtype=torch.float64)) \
            and torch.equal(loc, torch.full((6,), float(rank + 1), dtype=torch.float64))

        class _Exchange:
            MAX_DOUBLES, calls = 4, 0

            def allreduce(self, a, local_out=None):          # (BnExchange's contract: the rank's own values out of the same call)
                self.calls += 1
                if local_out is not None:
                    local_out.copy_(a)
                dist.all_reduce(a)
        ex = _Exchange()
        dpx = T.DataParallelCtx(dist.group.WORLD, ex)
        small, big = torch.ones(4, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
        dpx.reduce(small)
        dpx.reduce(big)
        mine = torch.full((4,), float(rank + 1), dtype=torch.float64)
        loc_x = dpx.reduce(mine, keep_local=True)
        ok_ctx = ok_ctx and ex.calls == 2 and bool((small == world).all()) and bool((big == world).all()) \
            and torch.equal(loc_x, torch.full((4,), float(rank + 1), dtype=torch.float64)) and bool((mine == tri).all())
        ok_bn = ok_bn and ok_ctx
        q.put((rank, ok_bcast, ok_avg, ok_bn, hasattr(net, "_sync_bn_group")))
    except Exception as e:          # surface the failure instead of letting the parent time out
        q.put((rank, False, False, False, repr(e)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_bcast, ok_avg, ok_bn, has_group in res:
        assert ok_bcast, f"rank {rank}: parameters not broadcast"
        assert ok_avg, f"rank {rank}: gradients not averaged"
        assert ok_bn, f"rank {rank}: SyncBN statistics wrong"
        assert has_group
