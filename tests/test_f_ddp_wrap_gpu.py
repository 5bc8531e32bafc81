"""GPU, one rank over RCCL: the reference's own wrapping (engine/forgery_engine.py:142-146) applied to the HIP model —
torch.nn.SyncBatchNorm.convert_sync_batchnorm + torch DistributedDataParallel — must run and reproduce the plain
step (second recipe).  With one rank every collective is the identity."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_reference_style_syncbn_plus_ddp_wrapping_runs():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import param_fill
    from tests.test_f_dp2_gpu import _build, _loss
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    x = param_fill.make_input(4, 256, seed=7).to(dev)
    tgt = param_fill.make_labels(4).to(dev)
    plain = _build(dev)
    _loss(plain(x), tgt).backward()
    with torch.no_grad():
        ref_loss = float(_loss(plain(x), tgt))
    ref = {k: p.grad.detach().clone() for k, p in plain.named_parameters() if p.grad is not None}
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(_build(dev)).to(dev)
        assert any(isinstance(mod, torch.nn.SyncBatchNorm) for mod in m.modules())
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=False)
        loss = _loss(ddp(x), tgt)
        loss.backward()
        assert abs(float(loss.detach()) - ref_loss) <= 1e-4 * abs(ref_loss)
        gmax = max(v.abs().max().item() for v in ref.values())
        worst = 0.0
        for k, p in ddp.module.named_parameters():
            if k in ref:
                assert p.grad is not None, k
                worst = max(worst, (p.grad - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 3e-3 * gmax))
        print(f"  DDP + SyncBN (1 rank) vs plain: worst gradient difference {worst:.2e}")
        from tests.margins import within
        assert within("SyncBN + DDP wrap vs plain step: worst gradient deviation", worst, 2e-3)
    finally:
        if created:
            dist.destroy_process_group()
