"""TEST INFRASTRUCTURE — import the reference (read-only, /root/reference) in THIS container.

Only used by ``oracle/make_golden.py``; never on the GPU box (the reference does not travel).
Stubs the three third-party packages the reference imports but this image lacks
(SURVEY.md §8c): torchvision (gaussian_blur only), timm (layers/helpers used by the ResNet
fork at import time), wandb.  Never writes under /root/reference (no bytecode).
"""
import sys
import types
import importlib.util

sys.dont_write_bytecode = True
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch.nn as nn

    if "torchvision" not in sys.modules:
        def gaussian_blur(*a, **k):
            raise NotImplementedError("torchvision is absent; blur perturbation is unpinned")
        tv = _stub("torchvision")
        tr = _stub("torchvision.transforms")
        fn = _stub("torchvision.transforms.functional", gaussian_blur=gaussian_blur)
        ds = _stub("torchvision.datasets", VisionDataset=object)
        tv.transforms, tr.functional, tv.datasets = tr, fn, ds
    if "timm" not in sys.modules:
        class _Unused(nn.Module):
            def __init__(self, *a, **k):
                raise NotImplementedError("timm stub: not exercised by the reference's configs")

        def create_attn(attn_layer, *a, **k):
            assert attn_layer is None
            return None

        def get_attn(x):
            return x

        def create_classifier(num_features, num_classes, pool_type="avg", **k):
            return nn.AdaptiveAvgPool2d(1), nn.Linear(num_features, num_classes)

        def build_model_with_cfg(cls, variant, pretrained, **kw):
            assert not pretrained
            return cls(**kw)

        def checkpoint_seq(*a, **k):
            raise NotImplementedError

        timm = _stub("timm")
        models = _stub("timm.models")
        layers = _stub("timm.models.layers", DropBlock2d=_Unused, DropPath=_Unused,
                       AvgPool2dSame=_Unused, BlurPool2d=_Unused, GroupNorm=_Unused,
                       create_attn=create_attn, get_attn=get_attn,
                       create_classifier=create_classifier)
        helpers = _stub("timm.models.helpers", build_model_with_cfg=build_model_with_cfg,
                        checkpoint_seq=checkpoint_seq)
        timm.models, models.layers, models.helpers = models, layers, helpers
    if "wandb" not in sys.modules:
        _stub("wandb")


def import_reference():
    """Returns the reference's ``model`` package and ``loss`` package."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import model as ref_model          # noqa: E402  (reference package)
    import loss as ref_loss            # noqa: E402
    import model.efficientnet.model as effmod
    effmod.load_pretrained_weights = lambda *a, **k: None   # no network (model.py:395)
    return ref_model, ref_loss


def import_abstract_engine():
    """engine/abstract_engine.py loaded directly (engine/__init__ pulls cv2/lmdb/albumentations)."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_abstract_engine", REF + "/engine/abstract_engine.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
