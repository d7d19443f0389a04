"""CPU ORACLE -- test infrastructure, NOT product code.

A straight-line float32 numpy restatement of the reference's per-frame forward
pass `UVLTrack.forward_test` (reference lib/models/uvltrack/uvltrack.py:41-45).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the shipped path (uvltrack_amd / lib.models) never does and fails
loudly when the HIP library is missing.

Parity pin: the reference holds NO test, golden vector or fixture for this path
(SURVEY.md §8c "parity unpinned by the reference"), so this oracle is pinned
against outputs of the reference itself, run in the build container by
oracle/make_golden.py (imports /root/reference with stub packages) and committed
as tests/golden/*.npz.  tests/test_oracle_golden.py checks every fixture to
atol 2e-4 (fp32 re-association only).

Every function cites the reference file:line it restates.  Weights are a dict
name -> float32 ndarray in the reference's state_dict schema.

bf16-emulating mode (`with emulate_bf16(): ...` or `emulate_bf16=True` on the forward functions): the same
restatement with the roundings of the HIP path inserted at the same places -- every GEMM / conv operand rounded to bf16
(RNE), f32 accumulation, q scaled by log2(e)/8 before its rounding, the un-normalised softmax numerators rounded before
P V, BatchNorm folded into the conv weights before their rounding, everything else (residual stream, LayerNorm, softmax,
GELU, sigmoid, normalisations) in f32.  It separates quantisation from defects: HIP-vs-emulated is gated an order of
magnitude tighter than HIP-vs-fp32 (tests/parity_util.py).  It is NOT pinned to the reference (there is no bf16
reference); the fp32 mode is.

`emulate_bf16_mode="fold"` (round 6) is the same with the roundings of the LayerNorm-FREE one-sequence frame
(uvltrack_amd/csrc/fold.h): a `norm -> Linear` pair (block.py:30-31 -> attn.qkv / mlp.fc1; the BertLayerNorms in front
of query/key/value and intermediate.dense, the embedding LayerNorm in front of layer 0) is one GEMM on the UN-normalised
rows rounded to bf16 against bf16(W gamma), y = rstd (a~ W'^T - mean colsum(W')) + (b + W beta), mean / rstd of the f32 row.
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf as _erf

f32 = np.float32
_EMU = False          # bf16-emulating mode (see the module docstring)
_FOLD = False         # ... with the LayerNorm-free frame's rounding points (norm -> Linear folded)
HEAD_ROUND = {"act": True, "w": True, "from_layer": 0}   # experiments only (tools/head_precision_experiment.py): which operands of the tower layers >= from_layer the emulation rounds
QSCALE = f32(0.18033688011112042)      # log2(e) / sqrt(64): the HIP attention kernels work in the log2 domain


class emulate_bf16:
    """Context manager: run the oracle with the HIP path's bf16 roundings."""

    def __init__(self, on=True, fold=False):
        self.on = bool(on)
        self.fold = bool(fold)

    def __enter__(self):
        global _EMU, _FOLD
        self.prev, _EMU = _EMU, self.on
        self.prev_fold, _FOLD = _FOLD, self.fold
        return self

    def __exit__(self, *exc):
        global _EMU, _FOLD
        _EMU = self.prev
        _FOLD = self.prev_fold
        return False


def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even) -> float32, as v_cvt_pk_bf16_f32 rounds."""
    u = np.ascontiguousarray(x, dtype=f32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(f32)


def _r(x):
    return bf16_round(x) if _EMU else x


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def linear(x, w, b=None):
    """nn.Linear: y = x W^T + b, W is [out, in].  (bf16 mode: both operands rounded, f32 accumulate, f32 bias.)"""
    y = _r(x) @ _r(w).T
    if b is not None:
        y = y + b
    return y.astype(f32, copy=False)


def layer_norm(x, w, b, eps):
    """nn.LayerNorm / BertLayerNorm (bert_backbone.py:231-244): biased variance, eps inside sqrt."""
    u = x.mean(-1, keepdims=True, dtype=f32)
    d = x - u
    s = (d * d).mean(-1, keepdims=True, dtype=f32)
    return (d / np.sqrt(s + f32(eps)) * w + b).astype(f32, copy=False)


def norm_linear(x, g, b, eps, w, bias):
    """nn.Linear applied to a LayerNorm'd row: linear(layer_norm(x, g, b, eps), w, bias) -- block.py:30-31 -> :42 / Mlp.fc1, bert_backbone.py:335-339 -> :366,
    :376-380 -> :289-291.  In the fold-emulating mode it is the LayerNorm-free frame's single GEMM (csrc/fold.h, gemm.hip LNF): operands bf16(x) and
    bf16(W gamma), f32 statistics of x, y = rstd (acc - mean colsum) + (bias + W beta)."""
    if not (_EMU and _FOLD):
        return linear(layer_norm(x, g, b, eps), w, bias)
    u = x.mean(-1, keepdims=True, dtype=f32)
    var = ((x * x).mean(-1, keepdims=True, dtype=f32) - u * u).clip(min=0)       # one-pass variance from (sum, sum of squares), as the partials give it
    rstd = (f32(1.0) / np.sqrt(var + f32(eps))).astype(f32)
    wf = bf16_round((w * g[None, :]).astype(f32))
    cs = wf.sum(-1, dtype=f32)
    bf = (bias + w @ b).astype(f32)
    acc = (bf16_round(x) @ wf.T).astype(f32)
    return (rstd * acc + ((-rstd * u) * cs + bf)).astype(f32)


def gelu(x):
    """exact erf GELU: nn.GELU() (backbones/utils.py:57) and bert gelu (bert_backbone.py:118-124)."""
    return (x * f32(0.5) * (f32(1.0) + _erf(x / f32(np.sqrt(2.0))))).astype(f32, copy=False)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m, dtype=f32)
    return (e / e.sum(axis=axis, keepdims=True, dtype=f32)).astype(f32, copy=False)


def sigmoid(x):
    return (f32(1.0) / (f32(1.0) + np.exp(-x, dtype=f32))).astype(f32, copy=False)


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=-1): x / max(||x||_2, eps)."""
    n = np.sqrt((x * x).sum(-1, keepdims=True, dtype=f32))
    return (x / np.maximum(n, f32(eps))).astype(f32, copy=False)


# --------------------------------------------------------------------------
# backbone pieces
# --------------------------------------------------------------------------
def patch_embed(img, w, b):
    """PatchEmbed.forward (mae_vit.py:94-100): Conv2d(3,D,k16,s16) == GEMM over
    (c,kh,kw)-ordered patch vectors; output [B, G*G, D], token s = i*G + j."""
    B, C, H, W = img.shape
    G = H // 16
    p = img.reshape(B, C, G, 16, G, 16).transpose(0, 2, 4, 1, 3, 5).reshape(B, G * G, C * 256)
    return linear(p, w.reshape(w.shape[0], -1), b)


def patchify(sd, z, x):
    """MaskedAutoencoderViT.patchify (mae_vit.py:203-215)."""
    v = "backbone.vit."
    pw, pb = sd[v + "patch_embed.proj.weight"], sd[v + "patch_embed.proj.bias"]
    zt = patch_embed(z, pw, pb) + sd[v + "pos_embed_z"]
    xt = patch_embed(x, pw, pb) + sd[v + "pos_embed_x"]
    cls = np.broadcast_to(sd[v + "cls_token"], (x.shape[0], 1, pw.shape[0]))
    return np.concatenate([cls, zt, xt], axis=1).astype(f32)


def bert_embedding(sd, ids, tmask, with_pre=False):
    """BertModel.embedding (bert_backbone.py:740-750) + BertEmbeddings.forward (:260-274), eval mode."""
    e = "backbone.bert.embeddings."
    T = ids.shape[1]
    emb = sd[e + "word_embeddings.weight"][ids] + sd[e + "position_embeddings.weight"][:T][None] \
        + sd[e + "token_type_embeddings.weight"][0][None, None]
    u0 = emb.astype(f32)
    emb = layer_norm(u0, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], 1e-12)
    bert_mask = ((f32(1.0) - tmask.astype(f32)) * f32(-10000.0))[:, None, None, :]
    if with_pre:
        return emb, bert_mask, (u0, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"])
    return emb, bert_mask


def cat_mask(tmask, flag, nz, nx):
    """ModalityUnifiedFeatureExtractor.cat_mask (extractor.py:43-50). True = key is ignored."""
    B = flag.shape[0]
    fl = flag.reshape(B, 1)
    x_m = np.ones((B, nx), f32)
    z_m = np.ones((B, nz), f32) * (fl != 1)
    c_m = np.ones((B, 1), f32) * (fl != 1)
    t_m = tmask.astype(f32) * (fl != 0)
    mask = ~np.concatenate([c_m, z_m, x_m, t_m], axis=1).astype(bool)
    vmask = ~np.concatenate([c_m, z_m, x_m], axis=1).astype(bool)
    return mask, vmask


def vit_attention(sd, pre, x, key_mask, heads, ln=None):
    """Attention.forward (block.py:47-61).  ln = (gamma, beta, eps): x is the row BEFORE norm1 and the projection applies it (norm_linear)."""
    B, N, C = x.shape
    hd = C // heads
    if ln is not None:
        qkv = norm_linear(x, ln[0], ln[1], ln[2], sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    else:
        qkv = linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if _EMU:
        o = _attention_bf16(q, k, v, None if key_mask is None else np.where(key_mask, f32(-1e10), f32(0.0)))
    else:
        attn = (q @ k.transpose(0, 1, 3, 2)) * f32(hd ** -0.5)
        if key_mask is not None:
            attn = np.where(key_mask[:, None, None, :], f32(-1e10), attn)
        attn = softmax(attn.astype(f32), -1)
        o = attn @ v
    o = o.transpose(0, 2, 1, 3).reshape(B, N, C)
    return linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def _attention_bf16(q, k, v, key_add):
    """The fused attention of the HIP path (uvltrack_amd/csrc/attention.hip) with its roundings: q * log2(e)/8, k, v in
    bf16; scores in the log2 domain (f32); numerators exp2(s - max) rounded to bf16 for P V, row sum over the f32
    numerators; output rounded to bf16.  key_add [B, N] is the additive per-key term in the natural-log domain."""
    hd = q.shape[-1]
    assert hd == 64
    qs = bf16_round(q * QSCALE)
    s = qs @ bf16_round(k).transpose(0, 1, 3, 2)
    if key_add is not None:
        s = s + (key_add * f32(1.4426950408889634))[:, None, None, :]
    s = s.astype(f32)
    p = np.exp2(s - s.max(-1, keepdims=True)).astype(f32)
    l = p.sum(-1, keepdims=True, dtype=f32)
    return bf16_round((bf16_round(p) @ bf16_round(v)) / l)


def vit_block(sd, i, x, key_mask, heads):
    """Block.forward (block.py:29-32); LayerNorm eps 1e-6 (mae_vit.py:221); DropPath/LayerScale are Identity."""
    p = "backbone.vit.blocks.%d." % i
    x = x + vit_attention(sd, p + "attn.", x, key_mask, heads, ln=(sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6))
    h = gelu(norm_linear(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))     # Mlp.forward (backbones/utils.py:63-69)
    return (x + linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])).astype(f32)


def bert_layer(sd, i, y, bert_mask, heads, pre=None):
    """BertLayer.forward (bert_backbone.py:390-394) = BertSelfAttention (:299-325) + BertSelfOutput (:335-339)
    + BertIntermediate (:363-366) + BertOutput (:376-380); post-LN, eps 1e-12, eval mode.
    pre = (u, gamma, beta) with y = LayerNorm(u): the row in front of the LayerNorm that produced y (the fold-emulating mode projects from it);
    returns y_out, or (y_out, pre_out) when pre is given."""
    p = "backbone.bert.encoder.layer.%d." % i
    B, T, C = y.shape
    hd = C // heads

    def split(t):
        return t.reshape(B, T, heads, hd).transpose(0, 2, 1, 3)

    def proj(name):
        if pre is not None:
            return norm_linear(pre[0], pre[1], pre[2], 1e-12, sd[p + "attention.self." + name + ".weight"], sd[p + "attention.self." + name + ".bias"])
        return linear(y, sd[p + "attention.self." + name + ".weight"], sd[p + "attention.self." + name + ".bias"])
    q, k, v = split(proj("query")), split(proj("key")), split(proj("value"))
    if _EMU:
        ctx = _attention_bf16(q, k, v, bert_mask[:, 0, 0, :]).transpose(0, 2, 1, 3).reshape(B, T, C)
    else:
        s = (q @ k.transpose(0, 1, 3, 2)) / f32(np.sqrt(hd))
        s = s + bert_mask
        pr = softmax(s.astype(f32), -1)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, C)
    a = linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    u1 = (a + y).astype(f32)
    g1, b1 = sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"]
    a = layer_norm(u1, g1, b1, 1e-12)
    h = gelu(norm_linear(u1, g1, b1, 1e-12, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    o = linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    u2 = (o + a).astype(f32)
    g2, b2 = sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"]
    out = layer_norm(u2, g2, b2, 1e-12)
    return out if pre is None else (out, (u2, g2, b2))


def txt_token_of(txt, tmask, mode):
    """generate_txt_token (extractor.py:79-83)."""
    if mode == "mean":
        m = tmask.astype(f32)[..., None]
        return ((txt * m).sum(1, keepdims=True) / m.sum(1, keepdims=True)).astype(f32)
    return txt[:, :1]


def backbone_contrast(sd, img, txt, tmask, flag, nz, mode):
    """ModalityUnifiedFeatureExtractor.contractive_learning (extractor.py:85-93) -> [B, nx, 1]."""
    vis_token, x = img[:, :1], img[:, 1 + nz:]
    tt = txt_token_of(txt, tmask, mode)
    tau = np.exp(sd["backbone.logit_scale"]).astype(f32)
    xn = l2_normalize(x)
    lv = tau * (xn @ l2_normalize(vis_token).transpose(0, 2, 1))
    lt = tau * (xn @ l2_normalize(tt).transpose(0, 2, 1))
    grp = np.stack([lv, lt, (lv + lt) / f32(2.0)], axis=1)
    return grp[np.arange(flag.shape[0]), flag.reshape(-1)].astype(f32)


def backbone_forward(sd, spec, template, search, ids, tmask, flag, taps=None):
    """ModalityUnifiedFeatureExtractor.forward (extractor.py:52-77)."""
    img = patchify(sd, template, search)
    txt, bert_mask, tpre = bert_embedding(sd, ids, tmask, with_pre=True)
    mask, vmask = cat_mask(tmask, flag, spec.nz, spec.nx)
    me = sd["backbone.vit.modal_embed"]
    logits = []
    if taps is not None:
        taps["embed_img"], taps["embed_txt"] = img.copy(), txt.copy()
    for i in range(spec.depth):
        if i in spec.fusion_layers:
            # MaskedAutoencoderViT.forward_joint (mae_vit.py:193-200): the modal shift is permanent
            emb = np.concatenate([img + me[0], txt + me[1]], axis=1).astype(f32)
            emb = vit_block(sd, i, emb, mask, spec.heads)
            img, txt = emb[:, :spec.nv], emb[:, spec.nv:]
        else:
            img = vit_block(sd, i, img, vmask, spec.heads)
            txt, tpre = bert_layer(sd, i, txt, bert_mask, spec.heads, pre=tpre)
        if i in spec.cont_layers:
            logits.append(backbone_contrast(sd, img, txt, tmask, flag, spec.nz, spec.txt_token_mode))
        if taps is not None:
            taps["img_%d" % i], taps["txt_%d" % i] = img.copy(), txt.copy()
    B = img.shape[0]
    F = spec.feat_sz
    return {
        "search": img[:, 1 + spec.nz:], "template": img[:, 1:1 + spec.nz], "text": txt,
        "vis_token": img[:, :1], "txt_token": txt_token_of(txt, tmask, spec.txt_token_mode),
        "flag": flag.reshape(-1),
        "logits": np.stack(logits, axis=1).reshape(B, -1, F, F),
    }


# --------------------------------------------------------------------------
# head
# --------------------------------------------------------------------------
def conv3x3_bn_relu(sd, pre, x):
    """heads/utils.py:126-131 in eval mode: Conv2d(3x3, pad 1, bias) + BatchNorm2d(running stats, eps 1e-5) + ReLU.
    x is NCHW."""
    w, b = sd[pre + "0.weight"], sd[pre + "0.bias"]
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    cols = np.empty((B, H, W, C, 3, 3), f32)
    for kh in range(3):
        for kw in range(3):
            cols[:, :, :, :, kh, kw] = xp[:, :, kh:kh + H, kw:kw + W].transpose(0, 2, 3, 1)
    if _EMU:     # BatchNorm folded into the weights before they are rounded (rowops.hip::fold_conv_bn_kernel), bf16 activations
        scale = (sd[pre + "1.weight"] / np.sqrt(sd[pre + "1.running_var"] + f32(1e-5))).astype(f32)
        layer = int(pre.rstrip(".").rsplit(".", 1)[1])
        rnd = layer >= HEAD_ROUND["from_layer"]
        wf = (w * scale[:, None, None, None]).astype(f32)
        if rnd and HEAD_ROUND["w"]:
            wf = bf16_round(wf)
        bf = ((b - sd[pre + "1.running_mean"]) * scale + sd[pre + "1.bias"]).astype(f32)
        y = (bf16_round(cols) if (rnd and HEAD_ROUND["act"]) else cols).reshape(B * H * W, C * 9) @ wf.reshape(w.shape[0], -1).T + bf
        return np.maximum(y, f32(0.0)).astype(f32).reshape(B, H, W, -1).transpose(0, 3, 1, 2)
    y = cols.reshape(B * H * W, C * 9) @ w.reshape(w.shape[0], -1).T + b
    y = (y - sd[pre + "1.running_mean"]) / np.sqrt(sd[pre + "1.running_var"] + f32(1e-5)) * sd[pre + "1.weight"] + sd[pre + "1.bias"]
    y = np.maximum(y, f32(0.0)).astype(f32)
    return y.reshape(B, H, W, -1).transpose(0, 3, 1, 2)


def tower(sd, name, x):
    """One of the four nn.Sequential towers (head:28-50)."""
    for l in range(4):
        x = conv3x3_bn_relu(sd, "box_head.%s.%d." % (name, l), x)
    w, b = sd["box_head.%s.4.weight" % name], sd["box_head.%s.4.bias" % name]
    B, C, H, W = x.shape
    y = _r(x).transpose(0, 2, 3, 1).reshape(-1, C) @ w.reshape(w.shape[0], C).T + b      # bf16 mode: bf16 activations, f32 1x1 weights
    return y.reshape(B, H, W, -1).transpose(0, 3, 1, 2).astype(f32)


def head_contrast(sd, spec, search, prompt):
    """ModalityAdaptiveBoxHead.contractive_learning, test branch (head:140-148)."""
    tau = np.exp(sd["box_head.logit_scale"]).astype(f32)
    cs = tau * (l2_normalize(search) @ l2_normalize(prompt).transpose(0, 2, 1))
    zero = np.zeros_like(cs[:, :, :1])
    if spec.softmax_one:
        mid = np.concatenate([cs[:, :, 1:], zero], axis=-1).max(-1, keepdims=True)
        return np.concatenate([cs[:, :, :1], mid, zero], axis=-1).astype(f32)
    return np.concatenate([cs[:, :, :1], cs[:, :, 1:].max(-1, keepdims=True)], axis=-1).astype(f32)


def head_contrast_train(sd, spec, out, tem_mask, ctx_mask):
    """ModalityAdaptiveBoxHead.contractive_learning with no prompt in the dict (head:123-138) -- the branch
    UVLTrack.forward takes (grounding at sequence init, lib/test/tracker/uvltrack.py:57): the prompter runs inline on a
    context rolled by half a batch and cont_score has TWO channels.  Returns (cont_score [B,S,2], prompt [B,3,D])."""
    flag = np.asarray(out["flag"]).reshape(-1)
    B = flag.shape[0]
    vt, tt = out["vis_token"], out["txt_token"]
    grp = np.concatenate([vt, tt, (vt + tt) / f32(2.0)], axis=1)
    token = grp[np.arange(B), flag]
    search = out["search"]
    context = np.concatenate([search[B // 2:], search[:B // 2]], axis=0)
    prompt = prompter_forward(sd, out["template"], tem_mask, context, ctx_mask, token, flag)
    tau = np.exp(sd["box_head.logit_scale"]).astype(f32)
    cs = tau * (l2_normalize(search) @ l2_normalize(prompt).transpose(0, 2, 1))
    if spec.softmax_one:
        zero = np.zeros_like(cs[:, :, :1])
        mid = np.concatenate([cs[:, :, 1:], zero], axis=-1).max(-1, keepdims=True)
    else:
        mid = cs[:, :, 1:].max(-1, keepdims=True)
    return np.concatenate([cs[:, :, :1], mid], axis=-1).astype(f32), prompt


def head_forward(sd, spec, out, prompt, cont=None):
    """ModalityAdaptiveBoxHead.forward (head:62-94) + convert2bbox (head:108-119).  `cont` given = the training-branch
    cont_score of head_contrast_train."""
    flag = out["flag"]
    B = flag.shape[0]
    F = spec.feat_sz
    bid = np.arange(B)
    if cont is None:
        cont = head_contrast(sd, spec, out["search"], prompt)
    x = out["search"].transpose(0, 2, 1).reshape(B, -1, F, F)
    if spec.cls_tokenize:
        vt, tt = out["vis_token"], out["txt_token"]
        grp = np.concatenate([vt, tt, (vt + tt) / f32(2.0)], axis=1)
        token = grp[bid, flag][:, :, None, None]
        cls_map = sigmoid(tower(sd, "conv_cls", x * token))[:, 0]
    else:
        cls_map = sigmoid(tower(sd, "conv_cls", x))[:, 0]
    off = tower(sd, "conv_offset", x)
    if spec.offset_sigmoid:
        off = sigmoid(off)
    tr = sigmoid(tower(sd, "conv_bbox", x))
    gr = sigmoid(tower(sd, "conv_bbox_grounding", x))
    size = np.stack([tr, gr, tr], axis=1)[bid, flag]
    # convert2bbox
    p0 = softmax(cont, -1)[:, :, 0]
    score = cls_map.reshape(B, -1) * p0
    s_idx = score.argmax(-1)
    ctr = (sd["box_head.coodinate"] + off.reshape(B, 2, -1)) / f32(F)
    bbox_map = np.concatenate([ctr, size.reshape(B, 2, -1)], axis=1).transpose(0, 2, 1).astype(f32)
    res = dict(out)
    res.update({
        "cls_score": (cls_map * p0.reshape(B, F, F)) if spec.joint_cls else cls_map,
        "bbox_map": bbox_map,
        "pred_boxes": bbox_map[bid, s_idx][:, None],
        "cont_score": cont,
        "prompts": prompt, "prompt": prompt,
        "cls_score_test": cls_map,
        "score": score.astype(f32),          # not in the reference dict: kept for tie-aware argmax checks
    })
    return res


def forward_test(sd, spec, template, search, ids, tmask, prompt, flag, taps=None, emulate_bf16_mode=False):
    """UVLTrack.forward_test (uvltrack.py:41-45), eval semantics."""
    if emulate_bf16_mode:
        with emulate_bf16(fold=(emulate_bf16_mode == "fold")):
            return forward_test(sd, spec, template, search, ids, tmask, prompt, flag, taps)
    sd = {k: (np.asarray(v, dtype=f32) if np.asarray(v).dtype.kind == "f" else np.asarray(v)) for k, v in sd.items()}
    out = backbone_forward(sd, spec, template.astype(f32), search.astype(f32), np.asarray(ids), np.asarray(tmask),
                           np.asarray(flag).reshape(-1, 1), taps)
    return head_forward(sd, spec, out, prompt.astype(f32))


def forward(sd, spec, template, search, ids, tmask, tem_mask, ctx_mask, flag, emulate_bf16_mode=False):
    """UVLTrack.forward (uvltrack.py:18-24) in eval mode: backbone, then the head on its no-prompt branch.  This is what the
    tracker's grounding() runs at sequence init in NL mode (lib/test/tracker/uvltrack.py:45-62)."""
    if emulate_bf16_mode:
        with emulate_bf16():
            return forward(sd, spec, template, search, ids, tmask, tem_mask, ctx_mask, flag)
    sd = {k: (np.asarray(v, dtype=f32) if np.asarray(v).dtype.kind == "f" else np.asarray(v)) for k, v in sd.items()}
    out = backbone_forward(sd, spec, template.astype(f32), search.astype(f32), np.asarray(ids), np.asarray(tmask),
                           np.asarray(flag).reshape(-1, 1))
    cont, prompt = head_contrast_train(sd, spec, out, np.asarray(tem_mask).astype(bool), np.asarray(ctx_mask).astype(bool))
    return head_forward(sd, spec, out, prompt, cont=cont)


# --------------------------------------------------------------------------
# prompter ("next" row SURVEY.md 8f-1): DistributionBasedCrossAttention
# --------------------------------------------------------------------------
def prompter_forward(sd, tem, tem_mask, ctx, ctx_mask, token, flag):
    """DistributionBasedCrossAttention.forward (heads/utils.py:82-99) with distribute_attn (:58-79) and
    divide_background (:45-56), eval mode.  tem [B,nz,D], ctx [B,S,D], masks bool (True = target cell),
    token [B,D], flag [B] -> prompt [B,3,D] = (target, distractor, background) tokens."""
    pr = "box_head.prompter."
    B = tem.shape[0]
    qe = sd[pr + "query_embed.weight"]
    src_ = np.broadcast_to(qe[None], (B,) + qe.shape).astype(f32).copy()
    src_[:, 0] = src_[:, 0] + token
    tgt = np.concatenate([tem, ctx], axis=1).astype(f32)
    tgt_mask = np.concatenate([tem_mask, ctx_mask], axis=1).astype(bool)[:, None, :]
    tau = np.exp(sd[pr + "logit_scale"]).astype(f32)
    sim = (l2_normalize(token)[:, None, :] @ l2_normalize(tgt).transpose(0, 2, 1)) * tau          # [B,1,L]
    NEG = f32(-1e20)
    tgt_score = softmax(np.where(~tgt_mask, NEG, sim).astype(f32), -1)
    tgt_token = tgt_score @ tgt
    bgd_logit = np.where(tgt_mask, NEG, sim).astype(f32)
    bgd_score = softmax(bgd_logit, -1)
    # divide_background: ascending sort, cumulative mass < 0.25 is "pure background"
    values = np.sort(bgd_score, axis=-1)
    cmask = np.cumsum(values, axis=-1, dtype=f32) < f32(0.25)
    thr = np.where(cmask, f32(1.0), values).min(-1, keepdims=True)
    dis_mask = bgd_score >= thr
    bgd2 = softmax(np.where(dis_mask, NEG, bgd_logit).astype(f32), -1)
    dis = softmax(np.where(~dis_mask, NEG, bgd_logit).astype(f32), -1)
    bgd_token = bgd2 @ tgt
    dis_token = dis @ tgt
    src = np.concatenate([tgt_token, dis_token, bgd_token], axis=1).astype(f32) + src_
    h = gelu(linear(src, sd[pr + "mlp.fc1.weight"], sd[pr + "mlp.fc1.bias"]))
    src = linear(h, sd[pr + "mlp.fc2.weight"], sd[pr + "mlp.fc2.bias"]) + src
    grp = np.stack([src, src_, src], axis=1)
    return grp[np.arange(B), np.asarray(flag).reshape(-1)].astype(f32)


def forward_prompt(sd, spec, out, tem_mask, ctx_mask):
    """ModalityAdaptiveBoxHead.forward_prompt (head:96-106) / UVLTrack.forward_prompt (uvltrack.py:33-38)."""
    sd = {k: (np.asarray(v, dtype=f32) if np.asarray(v).dtype.kind == "f" else np.asarray(v)) for k, v in sd.items()}
    flag = np.asarray(out["flag"]).reshape(-1)
    vt, tt = out["vis_token"], out["txt_token"]
    grp = np.concatenate([vt, tt, (vt + tt) / f32(2.0)], axis=1)
    token = grp[np.arange(flag.shape[0]), flag]
    return prompter_forward(sd, out["template"], tem_mask, out["search"], ctx_mask, token, flag)


def forward_prompt_init(sd, spec, template, search, ids, tmask, tem_mask, ctx_mask, flag):
    """UVLTrack.forward_prompt_init (uvltrack.py:26-31): backbone, then the prompter."""
    sdn = {k: (np.asarray(v, dtype=f32) if np.asarray(v).dtype.kind == "f" else np.asarray(v)) for k, v in sd.items()}
    out = backbone_forward(sdn, spec, template.astype(f32), search.astype(f32), np.asarray(ids), np.asarray(tmask),
                           np.asarray(flag).reshape(-1, 1))
    return forward_prompt(sdn, spec, out, tem_mask, ctx_mask)


def box_masks(spec, batch, seed=0):
    """Synthetic target-cell masks in the style of the tracker's anno2mask (tracker:183-194): the cells whose centre
    lies inside a box, plus the box-centre cell."""
    rng = np.random.RandomState(1234 + seed)

    def one(g):
        m = np.zeros((batch, g, g), bool)
        for b in range(batch):
            cx, cy = rng.uniform(0.3, 0.7, 2) * g
            w, h = rng.uniform(0.2, 0.5, 2) * g
            c = np.arange(g) + 0.5
            m[b] = ((c > cy - h / 2) & (c < cy + h / 2))[:, None] & ((c > cx - w / 2) & (c < cx + w / 2))[None, :]
            m[b, int(cy), int(cx)] = True
        return m.reshape(batch, g * g)
    return one(spec.template_size // 16), one(spec.feat_sz)


# --------------------------------------------------------------------------
# tracker decode ("next" row SURVEY.md 8f-2): lib/test/tracker/uvltrack.py:116-125,167-173 + box_ops.clip_box
# --------------------------------------------------------------------------
def hann_window(feat_sz):
    """window_prior (tracker:64-68): np.outer(np.hanning(F), np.hanning(F)).flatten()."""
    h = np.hanning(feat_sz)
    return np.outer(h, h).flatten().astype(f32)


def clip_box(box, H, W, margin=0):
    """lib/utils/box_ops.py:117-126."""
    x1, y1, w, h = box
    x2, y2 = x1 + w, y1 + h
    x1 = min(max(0, x1), W - margin)
    x2 = min(max(margin, x2), W)
    y1 = min(max(0, y1), H - margin)
    y2 = min(max(margin, y2), H)
    return [x1, y1, max(margin, x2 - x1), max(margin, y2 - y1)]


def tracker_decode(cls_score_test, cont_score, bbox_map, window, state, resize_factor, image_hw, search_size, has_cont=True, margin=10):
    """Per-sample post-processing of UVLTrack.track (tracker:116-125): argmax of cls * hann * softmax(cont)[0], scale the
    box to the crop, map it back around the previous centre (map_box_back, tracker:167-173) and clip it.
    Returns (new_state [B,4] xywh, score [B], pred_box_net [B,4], idx [B])."""
    B = cls_score_test.shape[0]
    new_state, scores, nets, idxs = [], [], [], []
    for b in range(B):
        pred_boxes = bbox_map[b].reshape(-1, 4)
        pred_cls = cls_score_test[b].reshape(-1)
        pred_cont = softmax(cont_score[b], -1)[:, 0] if has_cont else np.ones_like(pred_cls)
        merge = pred_cls * window * pred_cont
        i = int(np.argmax(merge))
        net = pred_boxes[i]
        score = float((pred_cls * pred_cont)[i])
        cx, cy, w, h = (net.astype(np.float64) * search_size / float(resize_factor[b])).tolist()
        sx, sy, sw, sh = [float(v) for v in state[b]]
        cx_prev, cy_prev = sx + 0.5 * sw, sy + 0.5 * sh
        half_side = 0.5 * search_size / float(resize_factor[b])
        box = [cx + (cx_prev - half_side) - 0.5 * w, cy + (cy_prev - half_side) - 0.5 * h, w, h]
        new_state.append(clip_box(box, float(image_hw[b][0]), float(image_hw[b][1]), margin=margin))
        scores.append(score)
        nets.append(net)
        idxs.append(i)
    return np.asarray(new_state, f32), np.asarray(scores, f32), np.asarray(nets, f32), np.asarray(idxs, np.int64)


OUTPUT_KEYS = ("search", "template", "text", "vis_token", "txt_token", "logits", "cls_score",
               "cls_score_test", "bbox_map", "pred_boxes", "cont_score")
