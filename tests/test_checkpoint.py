"""Checkpoint ingest: `.pth.tar` files with a 'net' entry, loaded as the reference tracker does (tracker:24)."""
import numpy as np
import pytest
import torch

from uvltrack_amd import weightgen as wg
from uvltrack_amd.checkpoint import load_checkpoint, read_checkpoint
from uvltrack_amd.spec import spec_tiny


def _model(spec):
    from uvltrack_amd.model import ModalityAdaptiveBoxHead, ModalityUnifiedFeatureExtractor, UVLTrack
    return UVLTrack(ModalityUnifiedFeatureExtractor(spec), ModalityAdaptiveBoxHead(spec))


def test_roundtrip_and_gates(tmp_path):
    spec = spec_tiny()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in wg.make_state_dict(spec, 7, include_unused=True).items()}
    path = str(tmp_path / "UVLTrack_tiny_ep0300.pth.tar")
    torch.save({"epoch": 300, "net": sd, "optimizer": {"lr": 1e-4}}, path)
    assert set(read_checkpoint(path)) == set(sd)
    m = _model(spec)
    res = load_checkpoint(m, path)
    assert not res.missing_keys and not res.unexpected_keys
    got = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k].cpu(), v), k
    # strict=False tolerates extra / missing keys like the reference call, but not a foreign file
    sd2 = dict(sd)
    sd2["box_head.extra.weight"] = torch.zeros(3)
    del sd2["backbone.vit.norm.weight"]
    torch.save({"net": sd2}, path)
    res = load_checkpoint(_model(spec), path)
    assert res.unexpected_keys == ["box_head.extra.weight"] and res.missing_keys == ["backbone.vit.norm.weight"]
    torch.save({"net": {"foo": torch.zeros(1)}}, path)
    with pytest.raises(RuntimeError):
        load_checkpoint(_model(spec), path)
    torch.save({"model": sd}, path)
    with pytest.raises(KeyError):
        load_checkpoint(_model(spec), path)


class _Evil:
    def __reduce__(self):
        import os
        return (os.getenv, ("HOME",))


def test_rejected_pickle_is_not_reloaded_unsafely(tmp_path):
    """A checkpoint the restricted loader rejects must not be re-read with the full unpickler behind the caller's back."""
    import pickle
    path = str(tmp_path / "tampered.pth.tar")
    torch.save({"net": {"w": torch.zeros(1)}, "extra": _Evil()}, path)
    with pytest.raises(pickle.UnpicklingError, match="allow_unsafe_pickle"):
        read_checkpoint(path)
    assert set(read_checkpoint(path, allow_unsafe_pickle=True)) == {"w"}      # explicit opt-in for a trusted file


def test_submodule_weight_changes_invalidate_the_packed_copy():
    """The reference reads parameters live; here the HIP library holds a packed copy, so every nn.Module route to the weights --
    on the model or on a sub-module -- must tick the model's weight clock, and in-place parameter writes its version sum."""
    spec = spec_tiny()
    m = _model(spec)
    v0, t0 = m._weights_version, m._tensor_versions()
    m.backbone.load_state_dict(m.backbone.state_dict())
    assert m._weights_version > v0
    v1 = m._weights_version
    m.box_head.conv_cls.float()                                # _apply on a grand-child
    assert m._weights_version > v1
    with torch.no_grad():
        next(m.box_head.parameters()).mul_(1.0)
    assert m._tensor_versions() > t0
    v2 = m._weights_version
    m.mark_weights_dirty()
    assert m._weights_version > v2
