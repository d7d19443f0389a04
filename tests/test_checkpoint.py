"""Checkpoint ingest: `.pth.tar` files with a 'net' entry, loaded as the reference tracker does (tracker:24)."""
import collections

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


def test_real_reference_layout_loads_without_the_training_modules(tmp_path):
    """The reference trainer's files carry a `Settings` object of lib.train.admin.settings, optimizer state and statistics
    (lib/train/trainers/base_trainer.py:130-140).  That module does not exist here: the restricted unpickler must still return
    'net' -- with no opt-in flag -- and must not need (or import) the missing module."""
    import sys
    import types
    spec = spec_tiny()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in wg.make_state_dict(spec, 7, include_unused=True).items()}
    mod = types.ModuleType("lib.train.admin.settings_for_test")

    class Settings:
        def __init__(self):
            self.env = types.SimpleNamespace(workspace_dir="/x", tensorboard_dir="/y")
            self.script_name, self.config_name, self.local_rank = "uvltrack", "baseline_base", -1

    class StatValue:
        def __init__(self):
            self.history, self.val = [0.5, 0.25], 0.25
    Settings.__module__ = StatValue.__module__ = mod.__name__
    Settings.__qualname__, StatValue.__qualname__ = "Settings", "StatValue"
    mod.Settings, mod.StatValue = Settings, StatValue
    sys.modules[mod.__name__] = mod
    path = str(tmp_path / "UVLTrack_ep0300.pth.tar")
    try:
        torch.save({"epoch": 300, "actor_type": "UVLTrackActor", "net_type": "UVLTrack", "net": sd, "net_info": None, "constructor": None,
                    "optimizer": {"state": {0: {"exp_avg": torch.zeros(3)}}, "param_groups": [{"lr": 1e-4, "params": [0]}]},
                    "stats": {"train": collections.OrderedDict(loss=StatValue())}, "settings": Settings()}, path)
    finally:
        del sys.modules[mod.__name__]
    got = read_checkpoint(path)
    assert mod.__name__ not in sys.modules                      # nothing was imported on the file's behalf
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    res = load_checkpoint(_model(spec), path)
    assert not res.missing_keys and not res.unexpected_keys


class _Evil:
    def __init__(self, target):
        self.target = target

    def __reduce__(self):
        import os
        return (os.mkdir, (self.target,))


def test_code_in_a_checkpoint_is_never_executed(tmp_path):
    """A tampered file that asks the unpickler to call a function: the call must not happen, the weights still load."""
    path = str(tmp_path / "tampered.pth.tar")
    target = str(tmp_path / "created_by_the_pickle")
    torch.save({"net": {"w": torch.arange(3.0)}, "extra": _Evil(target)}, path)
    import os
    got = read_checkpoint(path)
    assert not os.path.exists(target)
    assert set(got) == {"w"} and torch.equal(got["w"], torch.arange(3.0))
    torch.save({"net": {"w": _Evil(target)}}, path)              # a non-tensor where a weight should be
    with pytest.raises(TypeError):
        read_checkpoint(path)
    assert not os.path.exists(target)


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
    # a tensor OBJECT replaced by attribute assignment / register_parameter is not in the cached flat list: the clock must tick, and
    # in-place writes to the NEW tensor must then be seen by the version sum
    m._tensor_versions()                                       # fills the cache
    v3 = m._weights_version
    node = m.box_head.conv_cls
    leaf_owner = next(mod for mod in node.modules() if any(True for _ in mod.parameters(recurse=False)))
    pname, old = next(iter(leaf_owner.named_parameters(recurse=False)))
    setattr(leaf_owner, pname, torch.nn.Parameter(old.detach().clone(), requires_grad=False))
    assert m._weights_version > v3
    t3 = m._tensor_versions()
    with torch.no_grad():
        getattr(leaf_owner, pname).add_(1.0)
    assert m._tensor_versions() > t3
    v4 = m._weights_version
    leaf_owner.register_buffer("extra_buf", torch.zeros(1))
    assert m._weights_version > v4
