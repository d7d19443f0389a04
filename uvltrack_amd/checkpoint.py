"""Checkpoint ingest: the released UVLTrack checkpoints are `torch.save({'net': state_dict, ...})` files
(`*.pth.tar`); the reference tracker loads them with
    network.load_state_dict(torch.load(checkpoint, map_location='cpu')['net'], strict=False)      (lib/test/tracker/uvltrack.py:24)
`load_checkpoint` does exactly that for this repository's model object and reports what `strict=False` let through, so a
wrong file fails loudly instead of silently running on random weights."""
from __future__ import annotations

import torch


def read_checkpoint(path: str, allow_unsafe_pickle: bool = False):
    """Returns the 'net' state_dict of a reference checkpoint (CPU tensors).

    The file is read with torch's restricted unpickler (`weights_only=True`: tensors and plain containers only).  Checkpoints are
    downloaded from public links, so a file the restricted loader rejects is NOT silently re-read with the full unpickler (that
    would execute whatever code the file carries): the error names what was rejected, and the caller opts in explicitly with
    `allow_unsafe_pickle=True` for a file they trust."""
    import pickle
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not allow_unsafe_pickle:
            raise pickle.UnpicklingError(
                "%s holds more than tensors and plain containers (%s).  Loading it would run arbitrary code from the file; "
                "pass allow_unsafe_pickle=True only for a checkpoint you trust." % (path, e)) from e
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(ckpt, dict) or "net" not in ckpt:
        raise KeyError("%s is not a UVLTrack checkpoint: expected a dict with key 'net' (got %s)" %
                       (path, sorted(ckpt)[:8] if isinstance(ckpt, dict) else type(ckpt).__name__))
    return ckpt["net"]


def load_checkpoint(model, path: str, strict: bool = False, min_match: float = 0.9, allow_unsafe_pickle: bool = False):
    """model.load_state_dict(torch.load(path)['net'], strict=strict) + a sanity gate: at least `min_match` of the model's own
    tensors must be present in the file with the right shape.  Returns the `load_state_dict` result (missing / unexpected keys)."""
    sd = read_checkpoint(path, allow_unsafe_pickle=allow_unsafe_pickle)
    own = model.state_dict()
    good = sum(1 for k, v in own.items() if k in sd and tuple(sd[k].shape) == tuple(v.shape))
    if good < min_match * len(own):
        raise RuntimeError("checkpoint %s matches only %d of the model's %d tensors -- wrong architecture (B/L) or not a UVLTrack file"
                           % (path, good, len(own)))
    return model.load_state_dict(sd, strict=strict)
