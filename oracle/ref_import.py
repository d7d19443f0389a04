"""Import the REAL reference (/root/reference) on CPU -- build container only.

TEST INFRASTRUCTURE.  /root/reference does not exist on the GPU box, so nothing
in `-m gpu` tests, smoke() or bench.py imports this module; it is used by
oracle/make_golden.py (fixture generation) and by the CPU-only test that
re-validates the numpy oracle against the live reference when it is present.

No reference file is edited or copied.  The import needs:
  * sys.modules stubs for easydict, timm, pytorch_pretrained_bert, torchvision
  * numpy.float alias (mae_vit.py:40 uses the removed np.float)
  * BertModel.from_pretrained -> randomly initialised BertModel(BertConfig)
  * torch.load -> {'model': {}} while the extractor constructor runs
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("UVL_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "models"))


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _install_stubs():
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _EasyDict
        sys.modules["easydict"] = m
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        vt = types.ModuleType("timm.models.vision_transformer")
        vt.PatchEmbed = object
        timm.models = models
        models.vision_transformer = vt
        sys.modules.update({"timm": timm, "timm.models": models, "timm.models.vision_transformer": vt})
    if "pytorch_pretrained_bert" not in sys.modules:
        p = types.ModuleType("pytorch_pretrained_bert")
        fu = types.ModuleType("pytorch_pretrained_bert.file_utils")
        fu.cached_path = lambda *a, **k: None
        fu.WEIGHTS_NAME = "pytorch_model.bin"
        fu.CONFIG_NAME = "bert_config.json"
        p.file_utils = fu
        sys.modules.update({"pytorch_pretrained_bert": p, "pytorch_pretrained_bert.file_utils": fu})
    if "torchvision" not in sys.modules:
        sys.modules["torchvision"] = types.ModuleType("torchvision")
    if not hasattr(np, "float"):
        np.float = float  # noqa: NPY001  (reference mae_vit.py:40)


def _purge_lib_modules():
    """The repo's own drop-in package is also called `lib`; make sure the reference's wins here."""
    for k in list(sys.modules):
        if k == "lib" or k.startswith("lib."):
            del sys.modules[k]


def build_reference_model(spec):
    """Return (model.eval(), cfg) of the real reference for a ModelSpec."""
    import torch
    sys.dont_write_bytecode = True          # never write __pycache__ into the read-only reference tree
    _install_stubs()
    _purge_lib_modules()
    sys.path.insert(0, REF_ROOT)
    try:
        import importlib
        cfg_mod = importlib.import_module("lib.config.uvltrack.config")
        cfg = cfg_mod.cfg
        is_b = (spec.dim, spec.depth, spec.heads) == (768, 12, 12)
        is_l = (spec.dim, spec.depth, spec.heads) == (1024, 24, 16)
        cfg.MODEL.HIDDEN_DIM = spec.dim
        cfg.MODEL.LEARNABLE_POSITION = False
        cfg.MODEL.BACKBONE.TYPE = "modality_unified_feature_extractor"
        cfg.MODEL.BACKBONE.PRETRAINED_PATH = "pretrain/mae_pretrain_vit_large.pth" if is_l else "pretrain/mae_pretrain_vit_base.pth"
        cfg.MODEL.BACKBONE.FUSION_LAYER = list(spec.fusion_layers)
        cfg.MODEL.BACKBONE.CONT_LOSS_LAYER = list(spec.cont_layers)
        cfg.MODEL.BACKBONE.TXT_TOKEN_MODE = spec.txt_token_mode
        cfg.MODEL.BACKBONE.LANGUAGE.TYPE = "pretrain/bert"
        cfg.MODEL.HEAD.TYPE = "modality_adaptive_box_head"
        cfg.MODEL.HEAD.HEAD_DIM = spec.head_dim
        cfg.MODEL.HEAD.CLS_TOKENIZE = spec.cls_tokenize
        cfg.MODEL.HEAD.OFFSET_SIGMOID = spec.offset_sigmoid
        cfg.MODEL.HEAD.JOINT_CLS = spec.joint_cls
        cfg.MODEL.HEAD.SOFTMAX_ONE = spec.softmax_one
        cfg.DATA.TEMPLATE.SIZE = spec.template_size
        cfg.DATA.SEARCH.SIZE = spec.search_size
        cfg.TEST.TEMPLATE_SIZE = spec.template_size
        cfg.TEST.SEARCH_SIZE = spec.search_size

        models = importlib.import_module("lib.models")
        ext = sys.modules["lib.models.backbones.modality_unified_feature_extractor"]
        bert_mod = sys.modules["lib.models.backbones.bert_backbone"]
        mae = sys.modules["lib.models.backbones.mae_vit"]

        bcfg = bert_mod.BertConfig(spec.vocab, hidden_size=spec.dim, num_hidden_layers=spec.bert_total_layers,
                                   num_attention_heads=spec.heads, intermediate_size=spec.ffn,
                                   max_position_embeddings=spec.max_pos)
        orig_fp = bert_mod.BertPreTrainedModel.from_pretrained
        orig_load = torch.load
        orig_factory = ext.mae_vit_base_patch16
        bert_mod.BertPreTrainedModel.from_pretrained = classmethod(lambda cls, *a, **k: cls(bcfg))
        torch.load = lambda *a, **k: {"model": {}}
        if not (is_b or is_l):
            from functools import partial
            ext.mae_vit_base_patch16 = lambda **kw: mae.MaskedAutoencoderViT(
                patch_size=16, embed_dim=spec.dim, depth=spec.depth, num_heads=spec.heads, mlp_ratio=spec.mlp_ratio,
                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **kw)
        try:
            model = models.uvltrack.build_model(cfg)
        finally:
            bert_mod.BertPreTrainedModel.from_pretrained = orig_fp
            torch.load = orig_load
            ext.mae_vit_base_patch16 = orig_factory
        misc = importlib.import_module("lib.utils.misc")
        model.eval()
        return model, cfg, misc.NestedTensor
    finally:
        sys.path.remove(REF_ROOT)


def release_reference_modules():
    """Drop the reference's `lib.*` from sys.modules so the repo's own `lib` can be imported afterwards."""
    _purge_lib_modules()


def run_reference(spec, sd_np, inputs, tap_layers=False, prompt_masks=None):
    """Load numpy weights into the real reference and run forward_test; returns dict of numpy arrays.
    prompt_masks=(template_mask, context_mask) additionally runs forward_prompt_init -> res['prompt_init']."""
    import torch
    model, cfg, NestedTensor = build_reference_model(spec)
    ref_sd = model.state_dict()
    # schema check: names and shapes must agree exactly (SURVEY.md §8b weight contract)
    mine = {k: tuple(np.asarray(v).shape) for k, v in sd_np.items()}
    theirs = {k: tuple(v.shape) for k, v in ref_sd.items()}
    if mine != theirs:
        missing = sorted(set(theirs) - set(mine))
        extra = sorted(set(mine) - set(theirs))
        bad = sorted(k for k in set(mine) & set(theirs) if mine[k] != theirs[k])
        raise RuntimeError("state_dict schema mismatch: missing=%s extra=%s shape=%s" % (missing[:8], extra[:8], bad[:8]))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    taps = {}
    hooks = []
    if tap_layers:
        for i, blk in enumerate(model.backbone.vit.blocks):
            hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__("blk_%d" % i, o.detach().numpy().copy())))
    with torch.no_grad():
        text = NestedTensor(torch.from_numpy(inputs["ids"]), torch.from_numpy(inputs["mask"]))
        out = model.forward_test(torch.from_numpy(inputs["template"]), torch.from_numpy(inputs["search"]), text,
                                 torch.from_numpy(inputs["prompt"]), torch.from_numpy(inputs["flag"]))
        prompt_init = None
        fwd = None
        if prompt_masks is not None:
            orig_cuda = torch.Tensor.cuda
            torch.Tensor.cuda = lambda self, *a, **k: self          # heads/utils.py:96 hard-codes .cuda()
            try:
                text2 = NestedTensor(torch.from_numpy(inputs["ids"]), torch.from_numpy(inputs["mask"]))
                prompt_init = model.forward_prompt_init(torch.from_numpy(inputs["template"]), torch.from_numpy(inputs["search"]), text2,
                                                        torch.from_numpy(prompt_masks[0]), torch.from_numpy(prompt_masks[1]),
                                                        torch.from_numpy(inputs["flag"])).detach().numpy().copy()
                # UVLTrack.forward in eval mode (the grounding call, tracker:57): head on its no-prompt branch
                text3 = NestedTensor(torch.from_numpy(inputs["ids"]), torch.from_numpy(inputs["mask"]))
                fo = model.forward(torch.from_numpy(inputs["template"]), torch.from_numpy(inputs["search"]), text3,
                                   torch.from_numpy(prompt_masks[0]), torch.from_numpy(prompt_masks[1]), torch.from_numpy(inputs["flag"]))
                fwd = {k: fo[k].detach().numpy().copy() for k in ("cont_score", "bbox_map", "pred_boxes", "cls_score", "cls_score_test", "prompts")}
            finally:
                torch.Tensor.cuda = orig_cuda
    for h in hooks:
        h.remove()
    res = {k: v.detach().numpy().copy() for k, v in out.items() if hasattr(v, "detach")}
    if prompt_init is not None:
        res["prompt_init"] = prompt_init
    if fwd is not None:
        for k, v in fwd.items():
            res["fwd." + k] = v
    res.update(taps)
    release_reference_modules()
    return res
