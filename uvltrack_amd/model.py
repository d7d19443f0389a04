"""Drop-in module objects behind the reference's registry surface.

`UVLTrack` mirrors reference lib/models/uvltrack/uvltrack.py:8-45 for the inference path:
`.backbone`, `.box_head`, `.to()`, `.eval()`, `.state_dict()`, `.load_state_dict(strict=False)` and
`.forward_test(template, search, text, prompt, flag) -> dict`.  The modules are plain parameter
containers with the reference's state_dict key schema; all arithmetic happens in the
HIP library through uvltrack_amd.engine.HipEngine.  There is no CPU path: forward_test on CPU tensors,
or without the built library, raises NativeLibraryError.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import weightgen
from ._native import NativeLibraryError
from .spec import ModelSpec, spec_from_cfg, state_dict_schema

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked", "coodinate")


class _WeightClock:
    """One counter per model, shared by every container below it: anything that can replace or rewrite parameters through the
    nn.Module API (load_state_dict / _apply = .to() / .half() / ... on the model OR on any sub-module) ticks it, and the packed
    bf16 copy inside the HIP library is rebuilt when the tick seen at packing time is stale."""

    def __init__(self):
        self.ticks = 0


class _Node(nn.Module):
    """A parameter container; children are created on demand from dotted state_dict names."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_clock", _WeightClock())

    def _adopt_clock(self, clock):
        for m in self.modules():
            if isinstance(m, _Node):
                object.__setattr__(m, "_clock", clock)

    def _apply(self, fn, *a, **k):
        self._clock.ticks += 1
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._clock.ticks += 1
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # a tensor OBJECT replaced through the nn.Module API (`node.weight = nn.Parameter(...)`, register_parameter / register_buffer) is not
    # in the cached flat list of UVLTrack._tensor_versions: tick the clock so that the list is rebuilt and the packed copy refreshed
    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor) and "_clock" in self.__dict__:
            self._clock.ticks += 1
        super().__setattr__(name, value)

    def register_parameter(self, name, param):
        if "_clock" in self.__dict__:
            self._clock.ticks += 1
        return super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent: bool = True):
        if "_clock" in self.__dict__:
            self._clock.ticks += 1
        return super().register_buffer(name, tensor, persistent=persistent)

    def _attach(self, dotted: str, tensor: torch.Tensor):
        parts = dotted.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        leaf = parts[-1]
        if leaf in _BUFFER_LEAVES:
            node.register_buffer(leaf, tensor)
        else:
            node.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))


def _init_tensor(name, shape, spec):
    """Cheap deterministic init (the reference initialises randomly and then loads a checkpoint)."""
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.long)
    if name.endswith(("pos_embed_z", "pos_embed_x", "coodinate", "logit_scale")):
        return torch.from_numpy(np.asarray(weightgen.make_tensor(0, name, shape, spec)))
    if name.endswith("running_var") or (name.endswith("weight") and len(shape) == 1):
        return torch.ones(shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        g = torch.Generator().manual_seed(weightgen._fnv1a64(name) & 0x7FFFFFFF)
        return (torch.rand(shape, generator=g) * 2 - 1) * float(np.sqrt(3.0 / fan_in))
    return torch.zeros(shape)


class ModalityUnifiedFeatureExtractor(_Node):
    """Parameter container for reference extractor.py:11-41 (ViT + truncated BERT)."""

    def __init__(self, spec: ModelSpec):
        super().__init__()
        self.spec = spec
        for name, shape in state_dict_schema(spec).items():
            if name.startswith("backbone."):
                self._attach(name[len("backbone."):], _init_tensor(name, shape, spec))


class ModalityAdaptiveBoxHead(_Node):
    """Parameter container for reference modality_adaptive_box_head.py:10-60."""

    def __init__(self, spec: ModelSpec):
        super().__init__()
        self.spec = spec
        self.feat_sz = spec.feat_sz
        for name, shape in state_dict_schema(spec).items():
            if name.startswith("box_head."):
                self._attach(name[len("box_head."):], _init_tensor(name, shape, spec))


class UVLTrack(nn.Module):
    """Inference-path mirror of reference UVLTrack (uvltrack.py:8-45)."""

    def __init__(self, backbone: ModalityUnifiedFeatureExtractor, box_head: ModalityAdaptiveBoxHead, max_batch: int = 64):
        super().__init__()
        self.backbone = backbone
        self.box_head = box_head
        self.spec: ModelSpec = backbone.spec
        self.max_batch = max_batch
        self._engine = None
        self._engine_key = None
        self._clock = _WeightClock()
        backbone._adopt_clock(self._clock)
        box_head._adopt_clock(self._clock)
        self.eval()

    # weights changed -> the packed copy in the HIP library is stale.  The reference reads its parameters live; here every route
    # through the nn.Module API ticks the shared clock (on the model or on model.backbone / model.box_head / any child), and
    # in-place writes to a parameter (p.copy_(), p.mul_(), optimiser steps) are caught by the tensors' version counters, which
    # forward_test sums on every call.  Writes through `.data` bypass both (PyTorch keeps no record of them): call
    # `mark_weights_dirty()` after such a write.
    @property
    def _weights_version(self):
        return self._clock.ticks

    def mark_weights_dirty(self):
        self._clock.ticks += 1

    def _apply(self, fn, *a, **k):
        self._clock.ticks += 1
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._clock.ticks += 1
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _tensor_versions(self):
        """Sum of the version counters of every parameter and buffer (in-place writes bump them).  The flat tensor list is cached per
        weight-clock tick -- every nn.Module route that can REPLACE a tensor object ticks the clock -- so a call is one pass over
        ~340 attributes (~25 us for UVLTrack-B) instead of a module-tree walk (0.6-1.1 ms, comparable to the whole frame)."""
        cache = self.__dict__.get("_flat_cache")
        if cache is None or cache[0] != self._clock.ticks:
            cache = (self._clock.ticks, list(self.parameters()) + list(self.buffers()))
            self.__dict__["_flat_cache"] = cache
        return sum(t._version for t in cache[1])

    def train(self, mode: bool = True):
        if mode:
            # the reference's profile_model.py never calls eval(), which leaves dropout / batch-norm in training
            # mode; this implementation is eval-only by design (SURVEY.md 7.3 "train/eval trap")
            import warnings
            warnings.warn("uvltrack_amd.UVLTrack implements eval semantics only; train(True) is ignored")
        return super().train(False)

    def _get_engine(self, device):
        from .engine import HipEngine
        key = (str(device), self._weights_version, self._tensor_versions())
        if self._engine is None or self._engine.device != torch.device(device):
            if self._engine is not None:
                self._engine.close()
            self._engine = HipEngine(self.spec, device, self.max_batch)
            self._engine_key = None
        if self._engine_key != key:
            self._engine.load_state_dict(self.state_dict())
            self._engine_key = key
        return self._engine

    def _engine_for_aux(self, device):
        """Engine for the calls that follow a forward pass on its outputs (decode: no weights at all; forward_prompt: the prompter
        MLP packed with the rest).  They run once per frame next to forward_test / a frame step, which already did the full
        staleness check -- here one integer compare of the weight clock suffices (in-place parameter writes between a frame's
        forward and its decode are caught by the next forward_test)."""
        eng = self._engine
        if eng is not None and self._engine_key is not None and self._engine_key[1] == self._weights_version and eng.device == torch.device(device):
            return eng
        return self._get_engine(device)

    def forward_test(self, template, search, text, prompt, flag):
        """Same signature and output dict as reference uvltrack.py:41-45."""
        if not search.is_cuda:
            raise NativeLibraryError("forward_test needs tensors on a HIP device (got %s); there is no CPU fallback" % search.device)
        p0 = next(self.parameters())
        if p0.device != search.device:
            raise RuntimeError("model is on %s but inputs are on %s -- call model.to(device) first" % (p0.device, search.device))
        eng = self._get_engine(search.device)
        ids, mask = text.tensors, text.mask
        if mask is None:
            mask = torch.ones_like(ids)
        # `self.cache_text = True` (off by default; the trackers in lib/test/tracker switch it on): when `text` carries the same
        # tensor objects as the previous call, unmodified, the text branch below the first fusion layer is not recomputed
        out = eng.forward(template, search, ids, mask, prompt, flag, reuse_text=bool(getattr(self, "cache_text", False)))
        out.pop("argmax", None)
        return out

    def make_frame_step(self, template, search, text, prompt, flag):
        """forward_test bound to fixed device buffers: returns a zero-argument callable that enqueues one frame and returns the
        output dict (the SAME tensors every call -- copy what must outlive the next call).  For per-frame loops that rewrite
        `search` in place (the tracker classes): no argument checking, allocation or packing per frame.  Honours `cache_text`."""
        if not search.is_cuda:
            raise NativeLibraryError("make_frame_step needs tensors on a HIP device (got %s); there is no CPU fallback" % search.device)
        eng = self._get_engine(search.device)
        ids, mask = text.tensors, text.mask
        if mask is None:
            mask = torch.ones_like(ids)
        step = eng.make_eager_step(template, search, ids, mask, prompt, flag, reuse_text=bool(getattr(self, "cache_text", False)))
        weights_version = self._weights_version

        def run():
            # the shared clock is one integer compare per frame; in-place parameter writes between frames (which only the
            # per-tensor version counters see) are checked by forward_test / a new make_frame_step, not here
            if self._weights_version != weights_version or self._engine is not eng:
                raise RuntimeError("the model's weights or device changed: build a new frame step")
            return step()
        return run

    def forward(self, template, search, text, template_mask, context_mask, flag):
        """Reference uvltrack.py:18-24 in eval mode -- the tracker's grounding call (lib/test/tracker/uvltrack.py:57).  The
        head takes its no-prompt branch: `cont_score` is [B,S,2], `prompts` the inline prompter's output.  Training
        (autograd, BatchNorm batch statistics, DropPath) is out of scope: eval semantics only."""
        if not search.is_cuda:
            raise NativeLibraryError("forward needs tensors on a HIP device (got %s); there is no CPU fallback" % search.device)
        eng = self._get_engine(search.device)
        ids, mask = text.tensors, text.mask
        if mask is None:
            mask = torch.ones_like(ids)
        out = eng.forward_full(template, search, ids, mask, template_mask, context_mask, flag)
        out.pop("argmax", None)
        return out

    def decode(self, out_dict, window, state, resize_factor, image_hw, margin: float = 10.0, has_cont: bool = True, host_out=None):
        """The tracker's per-frame post-processing on the device (lib/test/tracker/uvltrack.py:116-125,167-173): returns
        (new_state [B,4] xywh, score [B], pred_box_net [B,4], index [B]) as device tensors."""
        eng = self._engine_for_aux(out_dict["bbox_map"].device)
        return eng.decode(out_dict, window, state, resize_factor, image_hw, margin=margin, has_cont=has_cont, host_out=host_out)

    def forward_prompt(self, out_dict, template_mask, context_mask):
        """Reference uvltrack.py:33-38: new (target, distractor, background) prompt from a forward_test output dict."""
        eng = self._engine_for_aux(out_dict["search"].device)
        return eng.forward_prompt(out_dict, template_mask, context_mask)

    def forward_prompt_init(self, template, search, text, template_mask, context_mask, flag):
        """Reference uvltrack.py:26-31: backbone on (template, context crop), then the prompter.  The backbone pass reuses
        forward_test with a zero prompt (its head outputs are simply not used)."""
        if not search.is_cuda:
            raise NativeLibraryError("forward_prompt_init needs tensors on a HIP device (got %s); there is no CPU fallback" % search.device)
        eng = self._get_engine(search.device)
        ids, mask = text.tensors, text.mask
        if mask is None:
            mask = torch.ones_like(ids)
        zero_prompt = torch.zeros(search.shape[0], 3, self.spec.dim, device=search.device)
        out = eng.forward(template, search, ids, mask, zero_prompt, flag)
        return eng.forward_prompt(out, template_mask, context_mask)


def build_backbone(cfg):
    return ModalityUnifiedFeatureExtractor(spec_from_cfg(cfg))


def build_head(cfg):
    return ModalityAdaptiveBoxHead(spec_from_cfg(cfg))


def build_model(cfg):
    """registry.MODELS['uvltrack'] (reference uvltrack.py:47-57)."""
    from lib import registry
    backbone = registry.BACKBONES[cfg.MODEL.BACKBONE.TYPE](cfg)
    head = registry.HEADS[cfg.MODEL.HEAD.TYPE](cfg)
    return UVLTrack(backbone, head)
