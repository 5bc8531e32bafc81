"""CPU: checkpoint round trip in the reference's file format and the pretrained-backbone import that tolerates the
Spatial-Frequency keys a plain EfficientNet checkpoint lacks (model/efficientnet/utils.py:589-634)."""
import pytest
import torch


def _model():
    from oracle import param_fill
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.2)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    return m


def test_checkpoint_round_trip_reference_format(tmp_path):
    from unidefense_amd.engine.checkpoint import load_checkpoint, save_checkpoint
    a, b = _model(), _model()
    with torch.no_grad():
        for p in b.parameters():
            p.add_(1.0)
    path = str(tmp_path / "latest_model.bin")
    save_checkpoint(a, path, step=123, best_acc=0.9)
    raw = torch.load(path, map_location="cpu")
    assert set(raw) == {"step", "best_step", "best_auc", "best_acc", "model"} and len(raw["model"]) == 802
    meta = load_checkpoint(b, path)
    assert meta["step"] == 123 and meta["best_acc"] == 0.9
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    # a state dict taken from a DDP wrapper ('module.' prefix) loads too
    torch.save({"step": 1, "model": {"module." + k: v for k, v in sa.items()}}, path)
    with torch.no_grad():
        for p in b.parameters():
            p.zero_()
    load_checkpoint(b, path)
    assert all(torch.equal(sa[k], b.state_dict()[k]) for k in sa)


def test_pretrained_backbone_import_tolerates_only_sf_keys(tmp_path):
    m = _model()
    full = {k: v.clone() for k, v in m.backbone.state_dict().items()}
    plain = {k: v for k, v in full.items() if "sf_coef" not in k and "freq_conv" not in k}     # a stock EfficientNet-b4
    assert len(plain) < len(full)
    plain["_fc.weight"], plain["_fc.bias"] = torch.zeros(1000, 1792), torch.zeros(1000)       # dropped on load
    path = str(tmp_path / "adv-efficientnet-b4.pth")
    torch.save(plain, path)
    fresh = _model()
    with torch.no_grad():
        for p in fresh.backbone.parameters():
            p.zero_()
    fresh.load_backbone_weights(path)
    got = fresh.backbone.state_dict()
    assert all(torch.equal(got[k], full[k]) for k in plain if not k.startswith("_fc."))
    bad = dict(plain)
    bad.pop(next(k for k in plain if k.endswith("_bn0.weight")))
    torch.save(bad, path)
    with pytest.raises(RuntimeError, match="pretrained weights mismatch"):
        _model().load_backbone_weights(path)
    extra = dict(plain, **{"_blocks.0.not_a_key": torch.zeros(1)})
    torch.save(extra, path)
    with pytest.raises(RuntimeError, match="pretrained weights mismatch"):
        _model().load_backbone_weights(path)
