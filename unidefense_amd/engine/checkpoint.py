"""Checkpoint I/O in the reference's format.

The reference's engines write ``{"step", "best_step", "best_auc", "best_acc", "model": state_dict}`` to
``latest_model.bin`` / ``best_model.bin`` with torch.save (engine/forgery_engine.py:215-223) and leave `_load_ckpt`
unimplemented (engine/abstract_engine.py:67-68).  The models here keep the reference's state-dict key names
(tests/test_abi_cpu.py), so a file written by either side loads on the other; a `module.` prefix (a state dict taken from
a DistributedDataParallel wrapper) is tolerated on load.
"""
import torch


def save_checkpoint(model, path, step, best_step=1, best_auc=0., best_acc=0.):
    """best_* default to the reference's initial values (forgery_engine.py:159-161: 1, 0., 0.): its readers call
    round(ckpt['best_auc'], 4) on the file (forgery_engine.py:205-207), which a None would break."""
    module = getattr(model, "module", model)
    torch.save({"step": int(step), "best_step": int(best_step), "best_auc": float(best_auc), "best_acc": float(best_acc),
                "model": {k: v.detach().cpu() for k, v in module.state_dict().items()}}, path)


def load_checkpoint(model, path, strict=True):
    """Returns the checkpoint's metadata (everything but the weights)."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    module = getattr(model, "module", model)
    ret = module.load_state_dict(sd, strict=strict)
    if not strict and (ret.missing_keys or ret.unexpected_keys):
        print(f"load_checkpoint: missing {len(ret.missing_keys)}, unexpected {len(ret.unexpected_keys)} keys")
    return {k: v for k, v in ckpt.items() if k != "model"} if isinstance(ckpt, dict) and "model" in ckpt else {}
