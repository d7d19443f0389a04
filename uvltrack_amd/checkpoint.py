"""Checkpoint ingest: the released UVLTrack checkpoints are `torch.save({'net': state_dict, ...})` files
(`*.pth.tar`); the reference tracker loads them with
    network.load_state_dict(torch.load(checkpoint, map_location='cpu')['net'], strict=False)      (lib/test/tracker/uvltrack.py:24)
`load_checkpoint` does exactly that for this repository's model object and reports what `strict=False` let through, so a
wrong file fails loudly instead of silently running on random weights.

What is in such a file.  The reference trainer saves {'epoch', 'actor_type', 'net_type', 'net', 'net_info', 'constructor', 'optimizer',
'stats', 'settings'} (lib/train/trainers/base_trainer.py:130-140): besides tensors it holds a `Settings` object of
`lib.train.admin.settings`, a module this repository does not have, and whatever the optimizer / statistics classes pickle.  Torch's
`weights_only=True` loader rejects those files outright, and the full unpickler would (a) fail on the missing module and (b) execute
whatever a downloaded file asks it to.  `read_checkpoint` therefore reads the file with a RESTRICTED unpickler: the tensor / storage /
container rebuilders torch itself needs are resolved normally, every other global the pickle names is replaced by an inert placeholder
class -- nothing from the file is ever imported or called -- and only `ckpt['net']` is returned."""
from __future__ import annotations

import collections
import pickle
import types

import torch


class _Inert:
    """Stand-in for any class or function a checkpoint names that is not a tensor rebuilder: constructing, calling or restoring it
    does nothing and keeps nothing alive that could run code."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        pass

    def __reduce__(self):
        return (_Inert, ())

    # list / dict subclasses are restored with append / extend / __setitem__ calls
    def append(self, *a):
        pass

    def extend(self, *a):
        pass

    def __setitem__(self, *a):
        pass


_ALLOWED_EXACT = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("builtins", "set"): set, ("builtins", "frozenset"): frozenset, ("builtins", "slice"): slice, ("builtins", "complex"): complex,
    ("builtins", "bytearray"): bytearray,
}
_TORCH_REBUILDERS = {
    "torch._utils": ("_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_parameter_with_state", "_rebuild_qtensor"),
    "torch._tensor": ("_rebuild_from_type_v2",),
    "torch.nn.parameter": ("Parameter",),
    "torch": ("Size", "Tensor", "device", "dtype"),
    "torch.serialization": ("_get_layout",),
}


def _resolve(module: str, name: str):
    """The object a pickle global may resolve to, or the inert placeholder."""
    if (module, name) in _ALLOWED_EXACT:
        return _ALLOWED_EXACT[(module, name)]
    if module in _TORCH_REBUILDERS and name in _TORCH_REBUILDERS[module]:
        obj = __import__(module, fromlist=[name])
        return getattr(obj, name)
    if module in ("torch", "torch.storage") and name.endswith("Storage") and name.replace("_", "").isalnum():
        return getattr(__import__(module, fromlist=[name]), name)      # FloatStorage, HalfStorage, UntypedStorage, ... (data holders)
    if module == "torch" and name in ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool"):
        return getattr(torch, name)
    return _Inert


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        return _resolve(module, name)


def _restricted_pickle_module():
    m = types.ModuleType("uvltrack_amd._restricted_pickle")
    m.Unpickler = _RestrictedUnpickler
    m.load = lambda f, **kw: _RestrictedUnpickler(f, **kw).load()
    m.loads = lambda b, **kw: _RestrictedUnpickler(__import__("io").BytesIO(b), **kw).load()
    m.UnpicklingError = pickle.UnpicklingError
    m.HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    return m


def read_checkpoint(path: str):
    """Returns the 'net' state_dict of a reference checkpoint (CPU tensors).  Safe on untrusted files and independent of the
    reference's training modules (see the module docstring): no flag, no fallback to the full unpickler."""
    ckpt = torch.load(path, map_location="cpu", pickle_module=_restricted_pickle_module(), weights_only=False)
    if not isinstance(ckpt, dict) or "net" not in ckpt:
        raise KeyError("%s is not a UVLTrack checkpoint: expected a dict with key 'net' (got %s)" %
                       (path, sorted(map(str, ckpt))[:8] if isinstance(ckpt, dict) else type(ckpt).__name__))
    net = ckpt["net"]
    if not isinstance(net, dict) or not all(isinstance(k, str) and torch.is_tensor(v) for k, v in net.items()):
        bad = [k for k, v in net.items() if not torch.is_tensor(v)][:5] if isinstance(net, dict) else type(net).__name__
        raise TypeError("%s: 'net' must map names to tensors (offending entries: %s)" % (path, bad))
    return net


def load_checkpoint(model, path: str, strict: bool = False, min_match: float = 0.9):
    """model.load_state_dict(torch.load(path)['net'], strict=strict) + a sanity gate: at least `min_match` of the model's own
    tensors must be present in the file with the right shape.  Returns the `load_state_dict` result (missing / unexpected keys)."""
    sd = read_checkpoint(path)
    own = model.state_dict()
    good = sum(1 for k, v in own.items() if k in sd and tuple(sd[k].shape) == tuple(v.shape))
    if good < min_match * len(own):
        raise RuntimeError("checkpoint %s matches only %d of the model's %d tensors -- wrong architecture (B/L) or not a UVLTrack file"
                           % (path, good, len(own)))
    return model.load_state_dict(sd, strict=strict)
