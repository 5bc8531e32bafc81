"""engine.get_engine(name)(config, stage).train() / .test() — the orchestration surface of main.py:55-59 of the
reference, driven by a config dict laid out like its YAML files (config_template/forgery/model_udeb4.yml)."""
import pytest
import torch

CONFIG = {
    "model": {"name": "UDEB4", "num_classes": 2, "drop_rate": 0.2, "extractor": "efficientnet-b4"},
    "config": {"warmup_step": 2, "lambda_triplet": 0.1, "lambda_recons": 0.1, "lambda_freq": 1.0, "lambda_mask": 0.1,
               "lambda_fac": 0.1, "num_steps": 4, "log_steps": 2, "local_rank": 0,
               "optimizer": {"name": "adamw", "lr": 1e-4, "betas": [0.9, 0.999], "weight_decay": 5e-6, "amsgrad": True},
               "scheduler": {"name": "StepLR", "step_size": 22500, "gamma": 0.5}},
    "data": {"train_batch_size": 2, "size": 256},
}


def test_get_engine_names_and_cpu_refusal():
    from unidefense_amd.engine import get_engine, AbstractEngine
    for name in ("FE", "OCIM", "UE"):
        assert issubclass(get_engine(name), AbstractEngine)
    with pytest.raises(KeyError):
        get_engine("nope")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="GPU only"):
            get_engine("FE")(CONFIG, "Train")


def test_scheduler_factory():
    from unidefense_amd.engine.optim import build_scheduler
    p = torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.SGD([p], lr=0.1)
    s = build_scheduler(opt, {"name": "StepLR", "step_size": 2, "gamma": 0.5})
    for _ in range(2):
        opt.step(); s.step()
    assert abs(opt.param_groups[0]["lr"] - 0.05) < 1e-12
    opt2 = torch.optim.SGD([p], lr=0.3)
    c = build_scheduler(opt2, None)
    opt2.step()
    c.step()
    assert c.get_last_lr() == [0.3]
    with pytest.raises(KeyError):
        build_scheduler(opt, {"name": "NoSuchLR"})


def test_timm_schedulers_follow_the_published_formulas():
    """TimmStepLR / TimmCosineLR of the reference's table (scheduler/__init__.py:24,31): closed-form values of
    timm.scheduler.StepLRScheduler / CosineLRScheduler (timm is absent here: parity unpinned), driven the way the reference's
    engines drive every scheduler — step() with no argument, get_last_lr()."""
    import math
    from unidefense_amd.engine.optim import SCHEDULERS, build_scheduler
    assert set(SCHEDULERS) == {"ConstantLR", "StepLR", "TimmStepLR", "MultiStepLR", "CosineAnnealingLR",
                               "CosineAnnealingWarmRestarts", "ExponentialLR", "ReduceLROnPlateau", "TimmCosineLR"}
    p = torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.SGD([{"params": [p], "lr": 0.2}, {"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 0.1}], lr=0.2)
    s = build_scheduler(opt, {"name": "TimmStepLR", "decay_t": 3, "decay_rate": 0.5, "warmup_t": 2, "warmup_lr_init": 0.0})
    assert s.get_last_lr() == [0.0, 0.0]                                    # the warm-up's start value is set at construction
    seen = []
    for _ in range(9):
        s.step()
        seen.append(s.get_last_lr()[0])
    # t = 1: warm-up 0.1; from t = 2 the decay clock runs on t - 2 (warmup_prefix): 0.2 x 0.5 ** ((t - 2) // 3)
    want = [0.1] + [0.2 * 0.5 ** ((t - 2) // 3) for t in range(2, 10)]
    assert seen == pytest.approx(want, abs=1e-15)
    assert s.get_last_lr()[1] == pytest.approx(want[-1] / 2)                # every group from its own initial_lr
    opt = torch.optim.SGD([p], lr=1.0)
    c = build_scheduler(opt, {"name": "TimmCosineLR", "t_initial": 4, "lr_min": 0.1, "cycle_decay": 0.5, "cycle_limit": 2})
    got = []
    for _ in range(9):
        c.step()
        got.append(c.get_last_lr()[0])
    want = []
    for t in range(1, 10):
        i, tc = divmod(t, 4)
        want.append(0.1 if i >= 2 else 0.1 + 0.5 * (0.5 ** i - 0.1) * (1 + math.cos(math.pi * tc / 4)))
    assert got == pytest.approx(want, abs=1e-15)
    c.step(epoch=2)                                                         # timm's own calling convention still works
    assert c.get_last_lr()[0] == pytest.approx(0.1 + 0.45 * (1 + math.cos(math.pi / 2)))
    m = build_scheduler(torch.optim.SGD([p], lr=1.0), {"name": "TimmCosineLR", "t_initial": 2, "cycle_mul": 2.0, "cycle_limit": 3})
    vals = []
    for _ in range(7):
        m.step()
        vals.append(m.get_last_lr()[0])
    # cycles of length 2, 4, 8 starting at t = 0, 2, 6
    assert vals == pytest.approx([0.5, 1.0, 0.5 * (1 + math.cos(math.pi / 4)), 0.5, 0.5 * (1 + math.cos(3 * math.pi / 4)), 1.0,
                                  0.5 * (1 + math.cos(math.pi / 8))], abs=1e-12)
    with pytest.raises(TypeError):
        build_scheduler(opt, {"name": "TimmStepLR", "decay_t": 1, "noise_range_t": 3})


@pytest.mark.gpu
def test_engine_trains_and_tests_on_synthetic_batches():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import copy
    from unidefense_amd.engine import get_engine
    torch.manual_seed(0)
    eng = get_engine("FE")(copy.deepcopy(CONFIG), "Train")
    before = {k: v.detach().clone() for k, v in eng.model.named_parameters()}
    log = eng.train()
    assert log["step"] == 4 and all(torch.isfinite(torch.tensor(v)) for v in log.values())
    for k in ("total_loss", "cls_loss", "triplet_loss", "real_rec_loss", "real_freq_loss"):
        assert k in log, sorted(log)
    moved = sum(int(not torch.equal(before[k], v.detach())) for k, v in eng.model.named_parameters())
    assert moved >= 500, moved                         # 504 trainable tensors, two AdamW steps per train step
    assert abs(log["lr"] - 1e-4) < 1e-12               # warm-up (2 steps) finished, StepLR not yet decayed
    res = eng.test(batches=2)
    assert res["scores"].shape == (8,) and 0.0 <= res["acc"] <= 1.0
    assert torch.isfinite(res["scores"]).all()
    # the test stage's metrics (forgery_engine.py:446-452) and the checkpoint round trip ON the GPU model:
    for k in ("AUC", "EER", "ACER", "TPR5%", "ACC", "NumP", "NumN"):
        assert k in res, sorted(res)
    assert res["NumP"] == 4 and res["NumN"] == 4 and 0.0 <= res["AUC"] <= 1.0
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        eng.config["config"]["dir"] = d
        val = eng.validate(step=4, batches=2)
        assert eng.best_step == 4 and eng.best_auc == val["AUC"] and eng.best_acc == val["ACC"]
        import os
        assert sorted(os.listdir(d)) == ["best_model.bin", "latest_model.bin"]
        ck = torch.load(os.path.join(d, "best_model.bin"), map_location="cpu")
        assert ck["step"] == 4 and round(ck["best_auc"], 4) == round(val["AUC"], 4)        # the reference's reader idiom
        cfg2 = copy.deepcopy(CONFIG)
        cfg2["config"].update(dir=d, resume=True)
        eng2 = get_engine("FE")(cfg2, "Test")
        for (k1, a), (_, b) in zip(eng.model.state_dict().items(), eng2.model.state_dict().items()):
            assert torch.equal(a, b), k1
        res2 = eng2.test(batches=2)
        assert torch.equal(res2["scores"], res["scores"])          # same weights, same synthetic test batches


@pytest.mark.gpu
def test_engine_with_prefetched_host_batches():
    """config['data']['iterator'] = RealFakePrefetcher over two host-side sources: pinned copies on a side stream one
    step ahead; the engine consumes them (graph-captured from the 2nd step on) and the batches arrive intact."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import copy
    from unidefense_amd.engine import get_engine
    from unidefense_amd.engine.data import RealFakePrefetcher
    g = torch.Generator().manual_seed(3)
    real = [(torch.rand(2, 3, 256, 256, generator=g) * 2 - 1, torch.zeros(2, dtype=torch.long)) for _ in range(2)]
    fake = [(torch.rand(2, 3, 256, 256, generator=g) * 2 - 1, torch.ones(2, dtype=torch.long)) for _ in range(3)]
    pf = RealFakePrefetcher(real, fake)
    xr, yr, xf, yf = pf(1, 2, 256, "cuda:0")
    assert torch.equal(xr.cpu(), real[0][0]) and torch.equal(xf.cpu(), fake[0][0]) and yr.eq(0).all() and yf.eq(1).all()
    cfg = copy.deepcopy(CONFIG)
    cfg["config"]["num_steps"], cfg["config"]["log_steps"] = 3, 3
    cfg["data"]["iterator"] = pf
    log = get_engine("UE")(cfg, "Train").train()
    assert log["step"] == 3 and all(torch.isfinite(torch.tensor(v)) for v in log.values())


@pytest.mark.gpu
def test_engine_fp16_precision_mode():
    """config.precision = "fp16" (BASELINE configs[4]): fp16-MFMA GEMMs + half storage of the MBConv trunk, under the
    engine's GradScaler, with the passes graph-captured from the second step on: finite losses, parameters move.  The GEMM
    path is process-wide: building an engine leaves it alone, each engine selects its own on entry to train() / test()."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import copy
    from unidefense_amd import lib
    from unidefense_amd.engine import get_engine
    torch.manual_seed(0)
    cfg = copy.deepcopy(CONFIG)
    cfg["config"]["precision"] = "fp16"
    try:
        eng = get_engine("FE")(cfg, "Train")
        assert eng._gemm_path == 3 and eng.model_without_ddp.half_storage is True
        before = {k: v.detach().clone() for k, v in eng.model.named_parameters()}
        log = eng.train()
        assert log["step"] == 4 and all(torch.isfinite(torch.tensor(v)) for v in log.values()), log
        moved = sum(int(not torch.equal(before[k], v.detach())) for k, v in eng.model.named_parameters())
        assert moved >= 500, moved
        bad = copy.deepcopy(CONFIG)
        bad["config"]["precision"] = "bf16"
        with pytest.raises(ValueError):
            get_engine("FE")(bad, "Train")
        e32 = get_engine("FE")(copy.deepcopy(CONFIG), "Train")        # building an fp32 engine does not touch the path
        assert lib.call("ud_gemm_get_path") == 3
        e32.test(batches=1)
        assert lib.call("ud_gemm_get_path") == 0                      # ... running it selects its own
        eng.test(batches=1)
        assert lib.call("ud_gemm_get_path") == 3
    finally:
        lib.call("ud_gemm_set_path", 0)
        assert lib.call("ud_gemm_get_path") == 0
