"""TEST INFRASTRUCTURE — golden vectors for the constructor variants of UniDefenseModelEb4 the YAMLs do not use
(model/unidefense.py:36-38: bias=True on the decoder / filter convs, affine=False on their norms), recorded by running the
REFERENCE (imported from /root/reference, this container only) like oracle/make_golden.py does for the default model.

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_variants            (from the repo root)
Also checks, on the spot, that the oracle's state-dict keys for the variant are the reference's and that the oracle reproduces
the reference's eval outputs (the pin of oracle/eb4.py's bias / affine handling).
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import ref_import, param_fill, eb4  # noqa: E402
from oracle.make_golden import OUT, LAMBDAS, make_rng, pack_outputs, run_reference_train  # noqa: E402


def main():
    ref_model, ref_loss = ref_import.import_reference()
    torch.manual_seed(0)
    drop_rate = 0.5
    for tag, bias, affine in (("bias_noaffine", True, False), ("bias", True, True), ("noaffine", False, False)):
        m = ref_model.load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=drop_rate, bias=bias,
                                          affine=affine)
        want = eb4.eb4_state_shapes(2, bias=bias, affine=affine)
        have = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert want == have, (sorted(set(want) ^ set(have))[:10], [k for k in want if k in have and want[k] != have[k]][:10])
        param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
        # eval, N = 2
        x = param_fill.make_input(2, 256, seed=1)
        m.eval()
        store = {}
        with torch.no_grad():
            out = m(x)
            pack_outputs(out, "", store)
            sd = param_fill.fill_state_dict(want, 0.0, 0.3)
            ora = eb4.forward_eb4(sd, x, training=False)
        for k in ("cls_out", "rec"):
            e = ((ora[k] - out[k]).abs().max() / out[k].abs().max()).item()
            assert e < 1e-4, (tag, k, e)
        store["meta"] = np.array([2, 256, 1], dtype=np.int64)
        np.savez_compressed(os.path.join(OUT, f"udeb4_eval_n2_{tag}.npz"), **store)
        print(f"wrote udeb4_eval_n2_{tag}.npz (oracle == reference on the eval outputs)")
        if tag != "bias_noaffine":
            continue
        # train fwd + pass-1 loss (smooth variant) + bwd, N = 2; seeds (44, 144): the smallest relative gap between the two largest
        # channels entering torch.max in the dynamic filters is 3.3e-3 for this parameter set (searched 40..44), so the arg-max —
        # whose gradient is discontinuous — cannot flip under fp32 rounding differences
        n, in_seed, mask_seed = 2, 44, 144
        x = param_fill.make_input(n, 256, seed=in_seed)
        tgt = param_fill.make_labels(n)
        rng = make_rng(n, seed=mask_seed, drop_rate=drop_rate)
        with torch.no_grad():
            gap = eb4.forward_eb4(sd, x, training=True, drop_rate=drop_rate, rng=rng)["_max_gap"].item()
        print("smallest top-2 gap entering torch.max: %.2e" % gap)
        assert gap > 5e-4, gap
        lam = dict(LAMBDAS, lambda_recons=0.0, lambda_freq=0.0)
        out, losses = run_reference_train(m, ref_loss, x, tgt, rng, drop_rate, lam)
        store = {}
        pack_outputs(out, "", store)
        for k, v in losses.items():
            store["smooth_loss_" + k] = np.array(v.item(), dtype=np.float64)
        names, norms, heads, maxabs = [], [], [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            norms.append(p.grad.double().norm().item())
            maxabs.append(p.grad.abs().max().item())
            h = torch.zeros(8)
            f = p.grad.flatten()[:8]
            h[: f.numel()] = f
            heads.append(h.numpy())
        store["grad_names"] = np.array(names)
        store["smooth_grad_norms"] = np.array(norms, dtype=np.float64)
        store["smooth_grad_maxabs"] = np.array(maxabs, dtype=np.float64)
        store["smooth_grad_heads"] = np.stack(heads)
        store["meta"] = np.array([n, 256, in_seed, mask_seed], dtype=np.int64)
        np.savez_compressed(os.path.join(OUT, f"udeb4_train_n2_{tag}.npz"), **store)
        print(f"wrote udeb4_train_n2_{tag}.npz ({len(names)} grads)")


def main_res():
    """eval goldens of the two ResNet models built with bias=True, affine=False (model/unidefense.py:268-270, 448-450)"""
    from oracle import r18, r50
    from oracle.make_golden_r18 import pack
    ref_model, _ = ref_import.import_reference()
    torch.manual_seed(0)
    for name, ctor, shapes, fwd, size in (("udr18", dict(extractor="resnet18"), r18.r18_state_shapes, r18.forward_r18, 128),
                                          ("udr50", dict(extractor="resnet50"), r50.r50_state_shapes, r50.forward_r50, 256)):
        m = ref_model.load_model("UDR18" if name == "udr18" else "UDR50")(num_classes=2, drop_rate=0.5, bias=True, affine=False,
                                                                           **ctor)
        want = shapes(2, bias=True, affine=False)
        have = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert want == have, (sorted(set(want) ^ set(have))[:10], [k for k in want if k in have and want[k] != have[k]][:10])
        param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
        x = param_fill.make_input(2, size, seed=3)
        m.eval()
        store = {}
        with torch.no_grad():
            out = m(x)
            pack(out, store, "eval_")
            ora = fwd(param_fill.fill_state_dict(want, 0.0, 0.3), x, training=False)
        for k in ("cls_out", "rec"):
            e = ((ora[k] - out[k]).abs().max() / out[k].abs().max()).item()
            assert e < 1e-4, (name, k, e)
        store["meta"] = np.array([2, size, 3], dtype=np.int64)
        np.savez_compressed(os.path.join(OUT, f"{name}_eval_n2_bias_noaffine.npz"), **store)
        print(f"wrote {name}_eval_n2_bias_noaffine.npz (oracle == reference on the eval outputs)")


def make_rng_sized(n, seed, drop_rate, size, nblk=32, dc_rate=0.2):
    """make_golden.make_rng for any input size: the keep-masks follow the maps (x_b4: size / 16 rounded up, x_b5: size / 32)"""
    g = torch.Generator().manual_seed(seed)

    def bern(shape, keep):
        return (torch.rand(shape, generator=g) < keep).float()
    s4, s5 = -(-size // 16), -(-size // 32)
    rng = {"dec_keep": bern((n, 160, s4, s4), 0.8), "emb_keep": bern((n, 272, s5, s5), 1.0 - drop_rate),
           "feat_keep": bern((n, 1792), 1.0 - drop_rate), "drop_connect": {}}
    for idx in range(1, nblk):
        rng["drop_connect"][idx] = bern((n,), 1.0 - dc_rate * idx / nblk)
    return rng


def main_380():
    """UDEB4 at the reference's native resolution (config_template/uniatt/Prot1/data_ffpp.yml:71-72: 380 x 380; feature maps
    190 / 95 / 48 / 24 / 12): eval outputs and a train step (smooth pass-1 loss, all gradients) at N = 2"""
    ref_model, ref_loss = ref_import.import_reference()
    torch.manual_seed(0)
    drop_rate, size, n = 0.5, 380, 2
    m = ref_model.load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=drop_rate)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    sd = param_fill.fill_state_dict(eb4.eb4_state_shapes(2), 0.0, 0.3)
    x = param_fill.make_input(n, size, seed=1)
    m.eval()
    store = {}
    with torch.no_grad():
        out = m(x)
        pack_outputs(out, "", store)
        ora = eb4.forward_eb4(sd, x, training=False)
    for k in ("cls_out", "rec"):
        e = ((ora[k] - out[k]).abs().max() / out[k].abs().max()).item()
        assert e < 1e-4, (k, e)
    store["meta"] = np.array([n, size, 1], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "udeb4_eval_n2_s380.npz"), **store)
    print("wrote udeb4_eval_n2_s380.npz (oracle == reference on the eval outputs)")
    # train step: seeds searched for a safe arg-max gap in the dynamic filters
    for in_seed in range(60, 80):
        x = param_fill.make_input(n, size, seed=in_seed)
        rng = make_rng_sized(n, 100 + in_seed, drop_rate, size)
        with torch.no_grad():
            gap = eb4.forward_eb4(sd, x, training=True, drop_rate=drop_rate, rng=rng)["_max_gap"].item()
        print("seed", in_seed, "top-2 gap %.2e" % gap, flush=True)
        if gap > 1.5e-3:
            break
    else:
        raise SystemExit("no seed with a safe gap")
    tgt = param_fill.make_labels(n)
    lam = dict(LAMBDAS, lambda_recons=0.0, lambda_freq=0.0)
    out, losses = run_reference_train(m, ref_loss, x, tgt, rng, drop_rate, lam)
    store = {}
    pack_outputs(out, "", store)
    for k, v in losses.items():
        store["smooth_loss_" + k] = np.array(v.item(), dtype=np.float64)
    names, norms, heads = [], [], []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(p.grad.double().norm().item())
        h = torch.zeros(8)
        f = p.grad.flatten()[:8]
        h[: f.numel()] = f
        heads.append(h.numpy())
    store["grad_names"] = np.array(names)
    store["smooth_grad_norms"] = np.array(norms, dtype=np.float64)
    store["smooth_grad_heads"] = np.stack(heads)
    store["meta"] = np.array([n, size, in_seed, 100 + in_seed], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "udeb4_train_n2_s380.npz"), **store)
    print(f"wrote udeb4_train_n2_s380.npz ({len(names)} grads)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "380":
        main_380()
    elif len(sys.argv) > 1 and sys.argv[1] == "res":
        main_res()
    else:
        main()
