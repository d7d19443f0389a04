"""Model geometry of the UVLTrack per-frame forward pass and its state_dict schema.

`ModelSpec` carries every number the hot path needs (token counts, widths, layer
schedule, head switches).  It is derived from the same yaml keys the reference
reads:

  * ViT-B vs ViT-L is chosen by the substring 'base' / 'large' in
    MODEL.BACKBONE.PRETRAINED_PATH      (reference extractor.py:20,30)
  * geometry from DATA.TEMPLATE.SIZE / DATA.SEARCH.SIZE (extractor.py:21,31;
    heads/__init__.py:7)
  * BERT truncated to min(FUSION_LAYER) layers (extractor.py:28,38)

`state_dict_schema(spec)` lists (name, shape) in the order and naming of the
reference module tree (SURVEY.md §8b "Weight contract"), so released checkpoints
and the oracle's weights drop in unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import List, Tuple

PATCH = 16


@dataclass
class ModelSpec:
    dim: int = 768            # MODEL.HIDDEN_DIM == ViT embed dim == BERT hidden size
    heads: int = 12
    depth: int = 12
    fusion_layers: List[int] = field(default_factory=lambda: [6, 7, 8, 9, 10, 11])
    cont_layers: List[int] = field(default_factory=lambda: [3, 4, 5, 6, 7, 8, 9, 10, 11])
    template_size: int = 128
    search_size: int = 256
    text_len: int = 40        # MODEL.BACKBONE.LANGUAGE.BERT.MAX_QUERY_LEN
    head_dim: int = 256       # MODEL.HEAD.HEAD_DIM (conv tower width, not attention head_dim)
    vocab: int = 30522
    max_pos: int = 512
    bert_total_layers: int = 12   # layers that exist in the checkpoint; only the first n_bert run
    mlp_ratio: int = 4
    txt_token_mode: str = "cls"
    cls_tokenize: bool = False
    offset_sigmoid: bool = True
    joint_cls: bool = False
    softmax_one: bool = True

    # ---- derived ---------------------------------------------------------
    @property
    def hd(self) -> int:
        return self.dim // self.heads

    @property
    def nz(self) -> int:
        return (self.template_size // PATCH) ** 2

    @property
    def nx(self) -> int:
        return (self.search_size // PATCH) ** 2

    @property
    def feat_sz(self) -> int:
        return self.search_size // PATCH

    @property
    def nv(self) -> int:      # visual tokens: cls + z + x
        return 1 + self.nz + self.nx

    @property
    def nj(self) -> int:      # joint tokens: visual + text
        return self.nv + self.text_len

    @property
    def n_bert(self) -> int:  # BERT layers that run == first fusion layer index
        return min(self.fusion_layers)

    @property
    def ffn(self) -> int:
        return self.dim * self.mlp_ratio

    def validate(self) -> None:
        if self.dim % self.heads != 0:
            # same condition as reference block.py:37
            raise AssertionError("dim should be divisible by num_heads")
        fl = sorted(self.fusion_layers)
        if fl != list(range(fl[0], self.depth)):
            raise ValueError("FUSION_LAYER must be a contiguous tail of the ViT depth "
                             "(the reference indexes bert.encoder.layer[i] for every i not in it)")
        if self.txt_token_mode not in ("cls", "mean"):
            raise ValueError("TXT_TOKEN_MODE must be 'cls' or 'mean'")

    def to_dict(self) -> dict:
        return asdict(self)

    # ---- algorithmic work (SURVEY.md §8d formula) -------------------------
    def flops_per_frame(self, skip_text: bool = False, reuse_text: bool = False) -> float:
        D, nv = self.dim, self.nv
        nj = nv if skip_text else self.nj
        f = 0.0
        for i in range(self.depth):
            n = nj if i in self.fusion_layers else nv
            f += 24.0 * n * D * D + 4.0 * n * n * D
        if not skip_text and not reuse_text:
            T = self.text_len
            f += self.n_bert * (24.0 * T * D * D + 4.0 * T * T * D)
        f += 2.0 * (self.nz + self.nx) * 768 * D
        f += self.head_flops()
        return f

    def head_flops(self) -> float:
        S, C = self.nx, self.head_dim
        chans = [self.dim, C, C // 2, C // 4, C // 8]
        per_tower = sum(2.0 * S * 9 * chans[i] * chans[i + 1] for i in range(4))
        ones = 2.0 * S * (C // 8) * (1 + 2 + 2 + 2)
        return 4 * per_tower + ones

    def attn_flops_per_layer(self, n: int) -> float:
        return 4.0 * n * n * self.dim


def spec_b(template_size=128, search_size=256, **kw) -> ModelSpec:
    return ModelSpec(dim=768, heads=12, depth=12, fusion_layers=list(range(6, 12)),
                     cont_layers=list(range(3, 12)), template_size=template_size,
                     search_size=search_size, bert_total_layers=12, **kw)


def spec_l(template_size=128, search_size=256, **kw) -> ModelSpec:
    return ModelSpec(dim=1024, heads=16, depth=24, fusion_layers=list(range(12, 24)),
                     cont_layers=list(range(8, 24)), template_size=template_size,
                     search_size=search_size, bert_total_layers=24, **kw)


def spec_tiny(**kw) -> ModelSpec:
    """The small golden-fixture model of SURVEY.md §8c (D=128, hd=64, depth 4; head width 256 because the
    implicit-GEMM conv towers need every tower width to be a multiple of the 64-channel K chunk)."""
    base = dict(dim=128, heads=2, depth=4, fusion_layers=[2, 3], cont_layers=[1, 2, 3],
                template_size=32, search_size=64, text_len=8, head_dim=256, vocab=64,
                max_pos=32, bert_total_layers=4)
    base.update(kw)
    return ModelSpec(**base)


def spec_from_cfg(cfg) -> ModelSpec:
    """Read the reference's yaml/EasyDict keys (config.py:7-147) into a ModelSpec."""
    path = cfg.MODEL.BACKBONE.PRETRAINED_PATH
    if "base" in path:
        dim, heads, depth, bert_layers = 768, 12, 12, 12
    elif "large" in path:
        dim, heads, depth, bert_layers = 1024, 16, 24, 24
    else:
        # the reference leaves self.vit undefined in this case (extractor.py:20-38)
        raise ValueError("MODEL.BACKBONE.PRETRAINED_PATH must contain 'base' or 'large'")
    if int(cfg.MODEL.HIDDEN_DIM) != dim:
        raise ValueError("MODEL.HIDDEN_DIM %d does not match the ViT width %d" % (cfg.MODEL.HIDDEN_DIM, dim))
    spec = ModelSpec(
        dim=dim, heads=heads, depth=depth,
        fusion_layers=[int(i) for i in cfg.MODEL.BACKBONE.FUSION_LAYER],
        cont_layers=[int(i) for i in cfg.MODEL.BACKBONE.CONT_LOSS_LAYER],
        template_size=int(cfg.DATA.TEMPLATE.SIZE), search_size=int(cfg.DATA.SEARCH.SIZE),
        text_len=int(cfg.MODEL.BACKBONE.LANGUAGE.BERT.MAX_QUERY_LEN),
        head_dim=int(cfg.MODEL.HEAD.HEAD_DIM), bert_total_layers=bert_layers,
        txt_token_mode=str(cfg.MODEL.BACKBONE.TXT_TOKEN_MODE),
        cls_tokenize=bool(cfg.MODEL.HEAD.CLS_TOKENIZE),
        offset_sigmoid=bool(cfg.MODEL.HEAD.OFFSET_SIGMOID),
        joint_cls=bool(cfg.MODEL.HEAD.JOINT_CLS),
        softmax_one=bool(cfg.MODEL.HEAD.SOFTMAX_ONE),
    )
    spec.validate()
    return spec


TOWERS = ("conv_cls", "conv_offset", "conv_bbox", "conv_bbox_grounding")
TOWER_OUT = {"conv_cls": 1, "conv_offset": 2, "conv_bbox": 2, "conv_bbox_grounding": 2}


def state_dict_schema(spec: ModelSpec, include_unused: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """(name -> shape) for every tensor of the reference module's state_dict.

    `include_unused=False` drops tensors forward_test never reads (vit.norm, BERT
    pooler, prompter) -- they are still accepted by load_state_dict(strict=False).
    """
    D, F = spec.dim, spec.ffn
    sd: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    sd["backbone.logit_scale"] = ()
    v = "backbone.vit."
    sd[v + "cls_token"] = (1, 1, D)
    sd[v + "pos_embed_z"] = (1, spec.nz, D)
    sd[v + "pos_embed_x"] = (1, spec.nx, D)
    sd[v + "modal_embed"] = (2, D)
    sd[v + "patch_embed.proj.weight"] = (D, 3, PATCH, PATCH)
    sd[v + "patch_embed.proj.bias"] = (D,)
    for i in range(spec.depth):
        b = v + "blocks.%d." % i
        sd[b + "norm1.weight"] = (D,)
        sd[b + "norm1.bias"] = (D,)
        sd[b + "attn.qkv.weight"] = (3 * D, D)
        sd[b + "attn.qkv.bias"] = (3 * D,)
        sd[b + "attn.proj.weight"] = (D, D)
        sd[b + "attn.proj.bias"] = (D,)
        sd[b + "norm2.weight"] = (D,)
        sd[b + "norm2.bias"] = (D,)
        sd[b + "mlp.fc1.weight"] = (F, D)
        sd[b + "mlp.fc1.bias"] = (F,)
        sd[b + "mlp.fc2.weight"] = (D, F)
        sd[b + "mlp.fc2.bias"] = (D,)
    if include_unused:
        sd[v + "norm.weight"] = (D,)
        sd[v + "norm.bias"] = (D,)
    e = "backbone.bert.embeddings."
    sd[e + "word_embeddings.weight"] = (spec.vocab, D)
    sd[e + "position_embeddings.weight"] = (spec.max_pos, D)
    sd[e + "token_type_embeddings.weight"] = (2, D)
    sd[e + "LayerNorm.weight"] = (D,)
    sd[e + "LayerNorm.bias"] = (D,)
    for i in range(spec.n_bert):
        b = "backbone.bert.encoder.layer.%d." % i
        for nm in ("query", "key", "value"):
            sd[b + "attention.self.%s.weight" % nm] = (D, D)
            sd[b + "attention.self.%s.bias" % nm] = (D,)
        sd[b + "attention.output.dense.weight"] = (D, D)
        sd[b + "attention.output.dense.bias"] = (D,)
        sd[b + "attention.output.LayerNorm.weight"] = (D,)
        sd[b + "attention.output.LayerNorm.bias"] = (D,)
        sd[b + "intermediate.dense.weight"] = (F, D)
        sd[b + "intermediate.dense.bias"] = (F,)
        sd[b + "output.dense.weight"] = (D, F)
        sd[b + "output.dense.bias"] = (D,)
        sd[b + "output.LayerNorm.weight"] = (D,)
        sd[b + "output.LayerNorm.bias"] = (D,)
    if include_unused:
        sd["backbone.bert.pooler.dense.weight"] = (D, D)
        sd["backbone.bert.pooler.dense.bias"] = (D,)
    h = "box_head."
    C = spec.head_dim
    chans = [D, C, C // 2, C // 4, C // 8]
    for t in TOWERS:
        for l in range(4):
            p = h + "%s.%d." % (t, l)
            sd[p + "0.weight"] = (chans[l + 1], chans[l], 3, 3)
            sd[p + "0.bias"] = (chans[l + 1],)
            sd[p + "1.weight"] = (chans[l + 1],)
            sd[p + "1.bias"] = (chans[l + 1],)
            sd[p + "1.running_mean"] = (chans[l + 1],)
            sd[p + "1.running_var"] = (chans[l + 1],)
            sd[p + "1.num_batches_tracked"] = ()
        sd[h + "%s.4.weight" % t] = (TOWER_OUT[t], chans[4], 1, 1)
        sd[h + "%s.4.bias" % t] = (TOWER_OUT[t],)
    if include_unused:
        pr = h + "prompter."
        sd[pr + "logit_scale"] = ()
        sd[pr + "q.weight"] = (D, D)
        sd[pr + "kv.weight"] = (2 * D, D)
        sd[pr + "proj.weight"] = (D, D)
        sd[pr + "proj.bias"] = (D,)
        sd[pr + "query_embed.weight"] = (3, D)
        sd[pr + "norm.weight"] = (D,)
        sd[pr + "norm.bias"] = (D,)
        sd[pr + "mlp.fc1.weight"] = (F, D)
        sd[pr + "mlp.fc1.bias"] = (F,)
        sd[pr + "mlp.fc2.weight"] = (D, F)
        sd[pr + "mlp.fc2.bias"] = (D,)
    sd[h + "logit_scale"] = ()
    sd[h + "coodinate"] = (1, 2, spec.nx)
    return sd
