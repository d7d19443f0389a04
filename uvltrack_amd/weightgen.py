"""Deterministic, platform-independent synthetic weights for UVLTrack.

There is no network for checkpoints, so parity fixtures and the benchmark use
weights from a counter-based integer hash: value(seed, tensor-name, flat index).
Only numpy uint64 integer ops and exactly-representable float32 conversions are
used (no np.random streams), so this container, the GPU box and the oracle all
see bit-identical tensors.  SURVEY.md §8c asks for exactly this generator.

Scales are chosen per tensor kind so that activations stay O(1) through 12-24
layers and the head maps have dynamic range (xavier-initialised heads give
cls_score in [0.494, 0.521], which makes argmax parity meaningless).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from .spec import ModelSpec, state_dict_schema

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for c in s.encode("utf-8"):
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n float32 values in [-1, 1), function of (seed, name, offset+i) only."""
    base = (_fnv1a64(name) ^ ((seed * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    out = np.empty(n, dtype=np.float32)
    CH = 1 << 22
    for s in range(0, n, CH):
        e = min(n, s + CH)
        idx = np.arange(offset + s, offset + e, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + np.uint64(base))
        u24 = (h >> np.uint64(40)).astype(np.int64)            # 24 random bits
        out[s:e] = (u24.astype(np.float32) - np.float32(8388608.0)) * np.float32(1.0 / 8388608.0)
    return out


def sincos_pos_embed(dim: int, grid: int) -> np.ndarray:
    """Fixed 2-D sin-cos table, layout of reference mae_vit.py:63-78 as probed in
    SURVEY.md §8a3: token s=i*G+j -> [sin(j w), cos(j w), sin(i w), cos(i w)]."""
    q = dim // 4
    omega = 1.0 / (10000.0 ** (np.arange(q, dtype=np.float64) / q))
    ii, jj = np.meshgrid(np.arange(grid, dtype=np.float64), np.arange(grid, dtype=np.float64), indexing="ij")
    i = ii.reshape(-1, 1)
    j = jj.reshape(-1, 1)
    emb = np.concatenate([np.sin(j * omega), np.cos(j * omega), np.sin(i * omega), np.cos(i * omega)], axis=1)
    return emb.astype(np.float32)


def coordinate_buffer(spec: ModelSpec) -> np.ndarray:
    """box_head.coodinate (reference head:54-60): ch0 = column j, ch1 = row i."""
    F = spec.feat_sz
    i, j = np.meshgrid(np.arange(F), np.arange(F), indexing="ij")
    c = np.stack([j.reshape(-1), i.reshape(-1)])[None].astype(np.float32)
    if not spec.offset_sigmoid:
        c = c + np.float32(0.5)
    return c


def _kind_scale(name: str, shape) -> tuple:
    """(mode, scale, shift): value = shift + scale*u, u in [-1,1)."""
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("logit_scale"):
        return ("const", math.log(1.0 / 0.07), 0.0)
    if name.endswith("num_batches_tracked"):
        return ("const", 100.0, 0.0)
    if "running_var" in name:
        return ("u", 0.4, 1.0)
    if "running_mean" in name:
        return ("u", 0.2, 0.0)
    parts = name.split(".")
    is_bn = name.startswith("box_head.conv") and len(parts) == 5 and parts[3] == "1"
    is_norm = ("norm" in name.lower()) or is_bn
    if is_norm:
        return ("u", 0.2, 1.0) if leaf == "weight" else ("u", 0.1, 0.0)
    if "cls_token" in name or "modal_embed" in name:
        return ("u", 0.5, 0.0)
    if "embeddings" in name or "query_embed" in name:
        return ("u", 0.5, 0.0)
    if leaf == "bias":
        return ("u", 0.1, 0.0)
    if len(shape) == 4:                      # conv weight [Co, Ci, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
        gain = 1.0
        if name.startswith("box_head"):
            # tower layer index: first conv sees the un-normalised residual stream (rms ~3), the 1x1 feeds a sigmoid
            gain = {"0": 0.45, "4": 1.6}.get(parts[2], 1.41)
        return ("u", gain * math.sqrt(3.0 / fan_in), 0.0)
    if len(shape) == 2:                      # nn.Linear [out, in]
        gain = 1.0
        if ".mlp.fc2." in name or ".attn.proj." in name or "output.dense" in name:
            gain = 0.7                       # keep the residual stream from growing layer over layer
        return ("u", gain * math.sqrt(3.0 / shape[1]), 0.0)
    return ("u", 0.1, 0.0)


def make_tensor(seed: int, name: str, shape, spec: ModelSpec) -> np.ndarray:
    if name.endswith("pos_embed_z"):
        return sincos_pos_embed(spec.dim, spec.template_size // 16)[None]
    if name.endswith("pos_embed_x"):
        return sincos_pos_embed(spec.dim, spec.search_size // 16)[None]
    if name.endswith("coodinate"):
        return coordinate_buffer(spec)
    mode, scale, shift = _kind_scale(name, shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if mode == "const":
        return np.full(shape, scale, dtype=np.int64 if name.endswith("num_batches_tracked") else np.float32)
    u = uniform_pm1(seed, name, n)
    return (np.float32(shift) + np.float32(scale) * u).astype(np.float32).reshape(shape)


def make_state_dict(spec: ModelSpec, seed: int = 0, include_unused: bool = True) -> "OrderedDict[str, np.ndarray]":
    sd = OrderedDict()
    for name, shape in state_dict_schema(spec, include_unused).items():
        sd[name] = make_tensor(seed, name, shape, spec)
    return sd


def make_inputs(spec: ModelSpec, batch: int = 1, seed: int = 0, flags=None, all_valid_text: bool = False):
    """Synthetic frame inputs in the recipe of reference tracking/profile_model.py:70-74
    (images ~ unit-scale noise, random ids, ~30 % valid text mask, random prompt)."""
    B = batch
    z = 1.7 * uniform_pm1(seed, "in.template", B * 3 * spec.template_size ** 2).reshape(B, 3, spec.template_size, spec.template_size)
    x = 1.7 * uniform_pm1(seed, "in.search", B * 3 * spec.search_size ** 2).reshape(B, 3, spec.search_size, spec.search_size)
    idu = uniform_pm1(seed, "in.ids", B * spec.text_len).reshape(B, spec.text_len)
    ids = np.minimum(((idu + 1.0) * 0.5 * spec.vocab).astype(np.int64), spec.vocab - 1)
    mu = uniform_pm1(seed, "in.mask", B * spec.text_len).reshape(B, spec.text_len)
    mask = (mu > 0.38)
    mask[:, 0] = True                                  # [CLS] is always a real token in the tracker
    if all_valid_text:
        mask[:] = True
    prompt = 1.7 * uniform_pm1(seed, "in.prompt", B * 3 * spec.dim).reshape(B, 3, spec.dim)
    if flags is None:
        flags = [(b % 3) for b in range(B)]
    flag = np.asarray(flags, dtype=np.int64).reshape(B, 1)
    return dict(template=z.astype(np.float32), search=x.astype(np.float32), ids=ids,
                mask=mask, prompt=prompt.astype(np.float32), flag=flag)
