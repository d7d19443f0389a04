"""CPU ORACLE (torch form) -- test infrastructure, NOT product code.

The same restatement of `UVLTrack.forward_test` (reference lib/models/uvltrack/uvltrack.py:41-45) as oracle/uvl_oracle.py, written
with the ATen operators the reference's eager forward runs on a CPU -- `F.linear`, `F.layer_norm`, `F.conv2d`, `F.batch_norm`,
`torch.softmax`, `F.gelu` -- so that timing it (bench.py's `cpu_baseline` leg, kind "port-torch") times the kernels the reference's own CPU path
would spend its time in (tracking/profile_model.py:38-47 drives exactly these), which the numpy / OpenBLAS form does not.  The reference
itself cannot travel to the GPU box.  Only tests/ and bench.py::cpu_baseline import this module.

Parity pin: checked against every committed fixture (outputs of the real reference, oracle/make_golden.py) by
tests/test_oracle_golden.py at the numpy oracle's gate (2e-4).  Every function cites the reference lines it restates.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))


def prepare(sd):
    """state_dict (numpy or torch) -> torch float32 CPU tensors, once (the reference holds its parameters as tensors too)."""
    out = {}
    for k, v in sd.items():
        t = _t(v)
        out[k] = t.float() if t.dtype.is_floating_point else t
    return out


def patch_embed(img, w, b):
    """PatchEmbed.forward (mae_vit.py:94-100): Conv2d(3, D, k16, s16), flatten(2).transpose(1, 2)."""
    return F.conv2d(img, w, b, stride=16).flatten(2).transpose(1, 2)


def patchify(sd, z, x):
    """MaskedAutoencoderViT.patchify (mae_vit.py:203-215)."""
    v = "backbone.vit."
    pw, pb = sd[v + "patch_embed.proj.weight"], sd[v + "patch_embed.proj.bias"]
    zt = patch_embed(z, pw, pb) + sd[v + "pos_embed_z"]
    xt = patch_embed(x, pw, pb) + sd[v + "pos_embed_x"]
    cls = sd[v + "cls_token"].expand(x.shape[0], -1, -1)
    return torch.cat([cls, zt, xt], dim=1)


def bert_embedding(sd, ids, tmask):
    """BertModel.embedding (bert_backbone.py:740-750) + BertEmbeddings.forward (:260-274), eval mode."""
    e = "backbone.bert.embeddings."
    T = ids.shape[1]
    emb = F.embedding(ids, sd[e + "word_embeddings.weight"]) + sd[e + "position_embeddings.weight"][:T][None] \
        + sd[e + "token_type_embeddings.weight"][0][None, None]
    emb = bert_layer_norm(emb, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"])
    bert_mask = ((1.0 - tmask.float()) * -10000.0)[:, None, None, :]
    return emb, bert_mask


def bert_layer_norm(x, w, b):
    """BertLayerNorm (bert_backbone.py:231-244): biased variance, eps 1e-12 inside the sqrt == F.layer_norm."""
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-12)


def cat_mask(tmask, flag, nz, nx):
    """ModalityUnifiedFeatureExtractor.cat_mask (extractor.py:43-50).  True = key is ignored."""
    B = flag.shape[0]
    fl = flag.reshape(B, 1)
    x_m = torch.ones(B, nx)
    z_m = torch.ones(B, nz) * (fl != 1)
    c_m = torch.ones(B, 1) * (fl != 1)
    t_m = tmask.float() * (fl != 0)
    mask = ~torch.cat([c_m, z_m, x_m, t_m], dim=1).bool()
    vmask = ~torch.cat([c_m, z_m, x_m], dim=1).bool()
    return mask, vmask


def vit_attention(sd, pre, x, key_mask, heads):
    """Attention.forward (block.py:47-61)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * hd ** -0.5
    if key_mask is not None:
        attn = attn.masked_fill(key_mask[:, None, None, :], -1e10)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def vit_block(sd, i, x, key_mask, heads):
    """Block.forward (block.py:29-32); LayerNorm eps 1e-6 (mae_vit.py:221); Mlp.forward (backbones/utils.py:63-69)."""
    p = "backbone.vit.blocks.%d." % i
    D = x.shape[-1]
    x = x + vit_attention(sd, p + "attn.", F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6), key_mask, heads)
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def bert_layer(sd, i, y, bert_mask, heads):
    """BertLayer.forward (bert_backbone.py:390-394): self-attention (:299-325), self-output (:335-339), intermediate (:363-366),
    output (:376-380); post-LN, eval mode."""
    p = "backbone.bert.encoder.layer.%d." % i
    B, T, C = y.shape
    hd = C // heads
    split = lambda t: t.reshape(B, T, heads, hd).permute(0, 2, 1, 3)
    q = split(F.linear(y, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"]))
    k = split(F.linear(y, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"]))
    v = split(F.linear(y, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"]))
    s = (q @ k.transpose(-1, -2)) / float(np.sqrt(hd)) + bert_mask
    ctx = (s.softmax(dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    a = F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    a = bert_layer_norm(a + y, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"])
    h = F.gelu(F.linear(a, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    o = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    return bert_layer_norm(o + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"])


def txt_token_of(txt, tmask, mode):
    """generate_txt_token (extractor.py:79-83)."""
    if mode == "mean":
        m = tmask.float()[..., None]
        return (txt * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)
    return txt[:, :1]


def backbone_contrast(sd, img, txt, tmask, flag, nz, mode):
    """ModalityUnifiedFeatureExtractor.contractive_learning (extractor.py:85-93) -> [B, nx, 1]."""
    vis_token, x = img[:, :1], img[:, 1 + nz:]
    tt = txt_token_of(txt, tmask, mode)
    tau = sd["backbone.logit_scale"].exp()
    xn = F.normalize(x, dim=-1)
    lv = tau * (xn @ F.normalize(vis_token, dim=-1).transpose(-2, -1))
    lt = tau * (xn @ F.normalize(tt, dim=-1).transpose(-2, -1))
    grp = torch.stack([lv, lt, (lv + lt) / 2.0], dim=1)
    return grp[torch.arange(flag.shape[0]), flag.reshape(-1)]


def backbone_forward(sd, spec, template, search, ids, tmask, flag):
    """ModalityUnifiedFeatureExtractor.forward (extractor.py:52-77)."""
    img = patchify(sd, template, search)
    txt, bert_mask = bert_embedding(sd, ids, tmask)
    mask, vmask = cat_mask(tmask, flag, spec.nz, spec.nx)
    me = sd["backbone.vit.modal_embed"]
    logits = []
    for i in range(spec.depth):
        if i in spec.fusion_layers:
            emb = torch.cat([img + me[0], txt + me[1]], dim=1)          # forward_joint (mae_vit.py:193-200)
            emb = vit_block(sd, i, emb, mask, spec.heads)
            img, txt = emb[:, :spec.nv], emb[:, spec.nv:]
        else:
            img = vit_block(sd, i, img, vmask, spec.heads)
            txt = bert_layer(sd, i, txt, bert_mask, spec.heads)
        if i in spec.cont_layers:
            logits.append(backbone_contrast(sd, img, txt, tmask, flag, spec.nz, spec.txt_token_mode))
    B, Fz = img.shape[0], spec.feat_sz
    return {"search": img[:, 1 + spec.nz:], "template": img[:, 1:1 + spec.nz], "text": txt, "vis_token": img[:, :1],
            "txt_token": txt_token_of(txt, tmask, spec.txt_token_mode), "flag": flag.reshape(-1),
            "logits": torch.stack(logits, dim=1).reshape(B, -1, Fz, Fz)}


def tower(sd, name, x):
    """One of the four towers (head:28-50): 4 x [Conv2d 3x3 pad 1 + BatchNorm2d(eval, eps 1e-5) + ReLU] (heads/utils.py:126-131) + Conv2d 1x1."""
    for l in range(4):
        pre = "box_head.%s.%d." % (name, l)
        x = F.conv2d(x, sd[pre + "0.weight"], sd[pre + "0.bias"], padding=1)
        x = F.batch_norm(x, sd[pre + "1.running_mean"], sd[pre + "1.running_var"], sd[pre + "1.weight"], sd[pre + "1.bias"], False, 0.0, 1e-5)
        x = F.relu(x)
    return F.conv2d(x, sd["box_head.%s.4.weight" % name], sd["box_head.%s.4.bias" % name])


def head_contrast(sd, spec, search, prompt):
    """ModalityAdaptiveBoxHead.contractive_learning, test branch (head:140-148)."""
    tau = sd["box_head.logit_scale"].exp()
    cs = tau * (F.normalize(search, dim=-1) @ F.normalize(prompt, dim=-1).transpose(-2, -1))
    zero = torch.zeros_like(cs[:, :, :1])
    if spec.softmax_one:
        mid = torch.cat([cs[:, :, 1:], zero], dim=-1).max(-1, keepdim=True)[0]
        return torch.cat([cs[:, :, :1], mid, zero], dim=-1)
    return torch.cat([cs[:, :, :1], cs[:, :, 1:].max(-1, keepdim=True)[0]], dim=-1)


def head_forward(sd, spec, out, prompt):
    """ModalityAdaptiveBoxHead.forward (head:62-94) + convert2bbox (head:108-119)."""
    flag = out["flag"]
    B, Fz = flag.shape[0], spec.feat_sz
    bid = torch.arange(B)
    cont = head_contrast(sd, spec, out["search"], prompt)
    x = out["search"].transpose(1, 2).reshape(B, -1, Fz, Fz)
    if spec.cls_tokenize:
        vt, tt = out["vis_token"], out["txt_token"]
        token = torch.cat([vt, tt, (vt + tt) / 2.0], dim=1)[bid, flag][:, :, None, None]
        cls_map = tower(sd, "conv_cls", x * token).sigmoid()[:, 0]
    else:
        cls_map = tower(sd, "conv_cls", x).sigmoid()[:, 0]
    off = tower(sd, "conv_offset", x)
    if spec.offset_sigmoid:
        off = off.sigmoid()
    tr = tower(sd, "conv_bbox", x).sigmoid()
    gr = tower(sd, "conv_bbox_grounding", x).sigmoid()
    size = torch.stack([tr, gr, tr], dim=1)[bid, flag]
    p0 = cont.softmax(-1)[:, :, 0]
    score = cls_map.reshape(B, -1) * p0
    s_idx = score.argmax(-1)
    ctr = (sd["box_head.coodinate"] + off.reshape(B, 2, -1)) / float(Fz)
    bbox_map = torch.cat([ctr, size.reshape(B, 2, -1)], dim=1).transpose(1, 2)
    res = dict(out)
    res.update({"cls_score": (cls_map * p0.reshape(B, Fz, Fz)) if spec.joint_cls else cls_map, "bbox_map": bbox_map,
                "pred_boxes": bbox_map[bid, s_idx][:, None], "cont_score": cont, "prompts": prompt, "prompt": prompt,
                "cls_score_test": cls_map, "score": score})
    return res


def forward_test(sd, spec, template, search, ids, tmask, prompt, flag, prepared=False):
    """UVLTrack.forward_test (uvltrack.py:41-45), eval semantics, under torch.no_grad().  Returns a dict of numpy arrays."""
    if not prepared:
        sd = prepare(sd)
    with torch.no_grad():
        out = backbone_forward(sd, spec, _t(template).float(), _t(search).float(), _t(ids).long(), _t(tmask).bool(), _t(flag).long().reshape(-1, 1))
        res = head_forward(sd, spec, out, _t(prompt).float())
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in res.items()}
