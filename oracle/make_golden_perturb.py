"""TEST INFRASTRUCTURE — golden vectors for the pass-2 input perturbations, recorded from
the REFERENCE's own code imported from /root/reference (this container only; nothing of it is copied):

  function level (N=4, 3x32x32, fp32 CPU, lmda handed in by patching torch.rand):
    FrequencyStyleTransfer / SpatialStyleTransfer (model/modules.py:35-76), coral (utils/operation.py:20-45),
    downscale (model/modules.py:19-21; also the source-index vectors it induces at 128 / 256 / 320)
  branch level: UniDefenseModelRes18.forward's `need augmentation` block (model/unidefense.py:366-389, the same code
    as Eb4's :177-198) run with torch.manual_seed(s) for seeds hitting every branch; the perturbed batch is captured
    at the encoder's first conv (forward pre-hook) — this pins the ORDER and KIND of the global-RNG draws.
    The blur branch is skipped (torchvision absent: unpinned); the noise branch draws its field from the same CPU
    generator and is recorded too.

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_perturb      -> tests/golden/perturb_n4.npz
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import ref_import, param_fill          # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "perturb_n4.npz")
N, SIZE, IN_SEED = 4, 32, 61
PERT_REAL, PERT_FAKE = [1, 0], [1, 0]


class _Captured(Exception):
    pass


def main():
    ref_model, _ = ref_import.import_reference()
    import model.modules as refmod
    import model.unidefense as udmod
    from utils.operation import coral as ref_coral

    x = param_fill.make_input(N, SIZE, seed=IN_SEED)
    style = torch.cat([x[:2][PERT_REAL], x[2:][PERT_FAKE]], 0)
    rec = {"meta": np.array([N, SIZE, IN_SEED]), "pert_real": np.array(PERT_REAL), "pert_fake": np.array(PERT_FAKE)}

    lm = torch.tensor([0.55, 0.7, 0.85, 0.95])
    real_rand = torch.rand
    try:
        torch.rand = lambda shape, *a, **k: ((lm - 0.5) * 2.0).reshape(shape)     # lmda = rand/2 + 0.5
        rec["lmda"] = lm.numpy()
        rec["freq_transfer"] = refmod.FrequencyStyleTransfer()(x, style).numpy()
        rec["spat_transfer"] = refmod.SpatialStyleTransfer()(x, style).numpy()
    finally:
        torch.rand = real_rand
    rec["coral"] = torch.stack([ref_coral(s, c) for c, s in zip(x, style)], 0).numpy()
    rec["downscale"] = refmod.downscale(x).numpy()
    for s in (128, 256, 320):        # the composed source index along one axis at the configs' input sizes
        ramp = torch.arange(s, dtype=torch.float32).reshape(1, 1, 1, s).expand(1, 1, 2, s)
        rec[f"downscale_index_{s}"] = refmod.downscale(ramp)[0, 0, 0].numpy().astype(np.int64)

    m = ref_model.load_model("UDR18")(extractor="resnet18", num_classes=2, drop_rate=0.0).train()
    first = m.extractor.conv1 if hasattr(m.extractor, "conv1") else next(m.extractor.children())

    def hook(mod, args):
        raise _Captured(args[0].detach().clone())
    h = first.register_forward_pre_hook(hook)
    want = {}
    for seed in range(200):
        for color in (False, True):
            torch.manual_seed(seed)                           # replay of the branch draws, to label the case
            style_branch = bool(torch.rand(1) > 0.5)
            which = int(torch.randint(0, 2 if style_branch else 3, size=(1,)))
            name = (("freq" if which == 0 else "spat") + ("_coral" if color else "")) if style_branch \
                else ["noise", "blur", "down"][which]
            if name in want or name == "blur" or (not style_branch and color):
                continue
            torch.manual_seed(seed)
            try:
                m(x, torch.tensor(PERT_REAL), torch.tensor(PERT_FAKE), color)
            except _Captured as e:
                want[name] = (seed, color, e.args[0].numpy())
        if len(want) == 6:
            break
    h.remove()
    assert len(want) == 6, sorted(want)
    for name, (seed, color, arr) in want.items():
        rec[f"branch_{name}"] = arr
        rec[f"branch_{name}_seed"] = np.array([seed, int(color)])
        print(f"  branch {name:11s} seed {seed:3d} preserve_color {color}  |noise_x - x| max {np.abs(arr - x.numpy()).max():.3e}")
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
