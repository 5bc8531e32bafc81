"""GPU: HipAdamW (csrc/optim.hip, one multi-tensor launch) against torch.optim.AdamW on the same tensors: two weight-decay
groups with their own lr (timm's grouping, engine/forgery_engine.py:149-156), amsgrad on / off, GradScaler's grad_scale and
a skipped (found_inf) step, ragged tensor sizes incl. scalars and lengths that are not multiples of 4 / of the chunk.
Bar: parameters and optimizer state agree to 2e-6 of each tensor's largest entry after 6 steps (fp32 rounding of one
fused expression vs ATen's sequence of foreach ops)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(), (1,), (3,), (7, 5), (1000,), (65536,), (65537,), (300, 331), (3264, 96), (48, 3, 3, 3), (131075,)]


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in SHAPES]


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))


@pytest.mark.parametrize("amsgrad", [True, False])
def test_hip_adamw_matches_torch(amsgrad):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.engine.optim import HipAdamW
    dev = torch.device("cuda:0")
    pa, pb = _params(dev, 1), _params(dev, 1)

    def groups(ps):
        return [{"params": [p for p in ps if p.ndim <= 1], "weight_decay": 0.0, "lr": 2e-3},
                {"params": [p for p in ps if p.ndim > 1], "weight_decay": 5e-2}]
    ref = torch.optim.AdamW(groups(pa), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, amsgrad=amsgrad, foreach=True)
    hip = HipAdamW(groups(pb), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, amsgrad=amsgrad)
    scale = torch.tensor(1024.0, device=dev)
    gen = torch.Generator().manual_seed(7)
    for it in range(7):
        skip = it == 3
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, generator=gen).to(dev) * (10.0 ** (it % 3 - 2))
            a.grad = g.clone()                      # torch: already unscaled gradients
            b.grad = g * scale                      # ours: scaled gradients + grad_scale on the device
        if not skip:
            ref.step()
        hip.grad_scale, hip.found_inf = scale, torch.tensor(1.0 if skip else 0.0, device=dev)
        hip.step()
        # a scheduler moves the lr between steps
        for o in (ref, hip):
            for gr in o.param_groups:
                gr["lr"] *= 0.9
    torch.cuda.synchronize()
    assert hip.step_count() == 6
    bad = []
    for i, (a, b) in enumerate(zip(pa, pb)):
        e = _rel(b, a)
        if e > 2e-6:
            bad.append(("param", SHAPES[i], e))
        for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ()):
            e = _rel(hip.state[b][k], ref.state[a][k])
            if e > 2e-6:
                bad.append((k, SHAPES[i], e))
    assert not bad, bad


def test_grad_scaler_drives_hip_adamw():
    """torch.amp.GradScaler.step hands grad_scale / found_inf to the optimizer (no unscale pass); an inf gradient
    skips the update and halves the scale."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.engine.optim import HipAdamW
    dev = torch.device("cuda:0")
    p = torch.ones(1000, device=dev, requires_grad=True)
    opt = HipAdamW([p], lr=0.1, weight_decay=0.0, amsgrad=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10)
    loss = (p * 3.0).sum()
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    assert torch.allclose(p, torch.full_like(p, 0.9), atol=1e-5)          # first Adam step moves by lr * sign(g)
    p.grad = torch.full_like(p, float("inf"))
    before = p.detach().clone()
    scaler.step(opt)
    scaler.update()
    assert torch.equal(p, before) and opt.step_count() == 1 and scaler.get_scale() == 2 ** 9


@pytest.mark.parametrize("amsgrad", [True, False])
def test_state_dict_round_trip_and_interchange_with_torch_adamw(amsgrad):
    """Checkpoint / resume: 3 steps, state_dict() -> a freshly built HipAdamW on copies of the parameters ->
    load_state_dict() -> 3 more steps must equal 6 uninterrupted steps of torch.optim.AdamW (moments kept, bias corrections
    continued from step 3); and the two optimizers read each other's state dicts (same keys incl. the per-parameter `step`)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.engine.optim import HipAdamW
    dev = torch.device("cuda:0")
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=amsgrad)
    pa, pb = _params(dev, 2), _params(dev, 2)
    ref = torch.optim.AdamW(pa, foreach=True, **kw)
    hip = HipAdamW(pb, **kw)
    gen = torch.Generator().manual_seed(9)
    grads = [[torch.randn(p.shape, generator=gen).to(dev) for p in pa] for _ in range(6)]

    def run(opt, ps, its):
        for it in its:
            for p, g in zip(ps, grads[it]):
                p.grad = g.clone()
            opt.step()
    run(ref, pa, range(6))
    run(hip, pb, range(3))
    sd = hip.state_dict()
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
    keys = {"step", "exp_avg", "exp_avg_sq"} | ({"max_exp_avg_sq"} if amsgrad else set())
    assert all(set(st) == keys for st in sd["state"].values())
    pc = [p.detach().clone().requires_grad_(True) for p in pb]
    hip2 = HipAdamW(pc, **kw)
    hip2.load_state_dict(sd)
    run(hip2, pc, range(3, 6))
    assert hip2.step_count() == 6
    for i, (a, c) in enumerate(zip(pa, pc)):
        assert _rel(c, a) <= 2e-6, ("param", SHAPES[i], _rel(c, a))
        for k in keys - {"step"}:
            assert _rel(hip2.state[c][k], ref.state[a][k]) <= 2e-6, (k, SHAPES[i])
    # torch's optimizer resumes from HipAdamW's state, and HipAdamW from torch's
    pd = [p.detach().clone().requires_grad_(True) for p in pb]
    ref2 = torch.optim.AdamW(pd, foreach=True, **kw)
    ref2.load_state_dict(sd)
    run(ref2, pd, range(3, 6))
    pe = [p.detach().clone().requires_grad_(True) for p in pa]          # parameters after 6 torch steps ...
    hip3 = HipAdamW(pe, **kw)
    hip3.load_state_dict(ref.state_dict())                               # ... and torch's state after 6 steps
    assert all(_rel(d, a) <= 2e-6 for d, a in zip(pd, pa))
    for p, g in zip(pa, grads[0]):
        p.grad = g.clone()
    for p, g in zip(pe, grads[0]):
        p.grad = g.clone()
    ref.step()
    hip3.step()
    assert hip3.step_count() == 7 and all(_rel(e, a) <= 2e-6 for e, a in zip(pe, pa))


def test_multi_add_equals_per_tensor_add():
    """ud_multi_add (csrc/optim.hip: the gradient accumulation of the train step's second backward, engine/abstract_engine.py:281
    and :374 under one zero_grad) against torch's `dst += src` per tensor: bit-identical, for 300 tensors (three launches of up to
    120 items) of ragged sizes — scalars, lengths that are not multiples of 4 or of the 16384-element chunk, one of 11 M elements,
    empty ones, and views whose base address is only 4-byte aligned; captured into a hipGraph and replayed too."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd import kernels as K
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    sizes = [1, 3, 4, 5, 24, 1000, 16383, 16384, 16385, 65537, 0, 3264 * 3264, 131075, 7, 48 * 27]
    sizes = (sizes * 20)[:300]
    sizes[40] = 0
    dst, src = [], []
    for i, n in enumerate(sizes):
        n = n if (n < 10 ** 6 or i < 15) else 1000           # the 11 M tensor once
        d, s = torch.randn(n + 3, generator=g).to(dev), torch.randn(n + 3, generator=g).to(dev)
        o = i % 3                                             # offsets 0 / 4 / 8 bytes: aligned and unaligned bases mixed
        dst.append(d[o:o + n])
        src.append(s[(o + 1) % 3:(o + 1) % 3 + n])
    want = [d + s for d, s in zip(dst, src)]
    keep = [d.clone() for d in dst]
    K.multi_add(dst, src)
    torch.cuda.synchronize()
    for i, (d, w) in enumerate(zip(dst, want)):
        assert torch.equal(d, w), (i, sizes[i])
    # replayed from a graph: the (dst, src, numel) triples travel in the kernel arguments, nothing is staged on the device
    for d, k in zip(dst, keep):
        d.copy_(k)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            K.multi_add(dst, src)
        gr.replay()                                           # the capture itself does not execute
    torch.cuda.synchronize()
    for i, (d, w) in enumerate(zip(dst, want)):
        assert torch.equal(d, w), ("graph", i, sizes[i])
    with pytest.raises(ValueError):
        K.multi_add([dst[0]], [src[0].double()])
