"""Host side of the HIP forward pass: owns the native model handle, the activation workspace and
(optionally) a captured hipGraph.  PyTorch is used only for device memory and the current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _native
from .spec import ModelSpec

OUT_SHAPES = {
    "search": lambda s, B: (B, s.nx, s.dim), "template": lambda s, B: (B, s.nz, s.dim),
    "text": lambda s, B: (B, s.text_len, s.dim), "vis_token": lambda s, B: (B, 1, s.dim),
    "txt_token": lambda s, B: (B, 1, s.dim),
    "logits": lambda s, B: (B, len(s.cont_layers), s.feat_sz, s.feat_sz),
    "cls_score": lambda s, B: (B, s.feat_sz, s.feat_sz), "cls_score_test": lambda s, B: (B, s.feat_sz, s.feat_sz),
    "bbox_map": lambda s, B: (B, s.nx, 4), "pred_boxes": lambda s, B: (B, 1, 4),
    "cont_score": lambda s, B: (B, s.nx, 3 if s.softmax_one else 2),
}
_OUT_FIELD = {"search": "d_search", "template": "d_template", "text": "d_text", "vis_token": "d_vis_token",
              "txt_token": "d_txt_token", "logits": "d_logits", "cls_score": "d_cls_score",
              "cls_score_test": "d_cls_score_test", "bbox_map": "d_bbox_map", "pred_boxes": "d_pred_boxes",
              "cont_score": "d_cont_score"}


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _native.NativeLibraryError(
            "%s is on %s: the UVLTrack forward pass runs only on a HIP device (no CPU fallback)" % (what, t.device))


class HipEngine:
    """One native model per (device, process).  Not thread-safe (same contract as an nn.Module)."""

    def __init__(self, spec: ModelSpec, device: torch.device, max_batch: int = 64):
        spec.validate()
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.NativeLibraryError("HipEngine needs a HIP device, got %s" % device)
        self.lib = _native.load()
        self.max_batch = max_batch
        with torch.cuda.device(self.device):
            cfg = _native.config_from_spec(spec, max_batch)
            self.handle = self.lib.uvl_create(C.byref(cfg))
        if not self.handle:
            raise _native.NativeLibraryError("uvl_create failed: %s" % self.lib.uvl_last_error().decode())
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = 0
        self._graph_io = None
        self._text_primed = None                          # see _text_key
        self._txt_owner = None                            # see _set_primed
        self._ws_user_B = None                            # batch size the workspace was last carved for
        self.weights_loaded = False

    def close(self):
        if getattr(self, "handle", None):
            self.lib.uvl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ tuning (tools / tests)
    def tune_set(self, key: str, value: int):
        """uvl_tune_set on THIS engine's handle (there is no process-global tuning state); -1 restores the heuristic."""
        _native.check(self.lib.uvl_tune_set(self.handle, key.encode(), int(value)), "uvl_tune_set(%s)" % key)
        if key == "reset":
            self.__dict__["_tuning"] = {}                      # the handle holds heuristics only again: nothing to restore to
        else:
            self.__dict__.setdefault("_tuning", {})[key] = int(value)

    def debug_set(self, key: str, value: int):
        """uvl_debug_set on this engine's handle: A/B aids that are not launch heuristics (stop_layer, pair_text, fuse_contrast, fork_text, fuse_ln)."""
        _native.check(self.lib.uvl_debug_set(self.handle, key.encode(), int(value)), "uvl_debug_set(%s)" % key)

    def tuned(self, **kw):
        """Context manager: `with eng.tuned(gemm_cfg=11): ...` -- on exit the keys get back the values they had on entry (tune_set
        records what this engine's handle holds; an override made earlier, e.g. by bench.py --tune, survives the block), whatever happens."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            before = {k: self.__dict__.setdefault("_tuning", {}).get(k, -1) for k in kw}
            try:
                for k, v in kw.items():
                    self.tune_set(k, v)
                yield self
            finally:
                for k, v in before.items():
                    self.tune_set(k, v)
        return cm()

    # ------------------------------------------------------------------ weights
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_state_dict(self, sd: Dict[str, "torch.Tensor | np.ndarray"], strict: bool = False):
        """Upload a reference-format state_dict (names of SURVEY.md section 8b) and pack it."""
        unknown = []
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if isinstance(t, np.ndarray):
                    t = torch.from_numpy(np.ascontiguousarray(t))
                if not t.dtype.is_floating_point:
                    continue                                   # num_batches_tracked
                t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
                dims = (C.c_int64 * max(1, t.dim()))(*t.shape)
                rc = self.lib.uvl_load_tensor(self.handle, name.encode(), C.c_void_p(t.data_ptr()), t.dim(), dims, self._stream())
                if rc == -4:
                    unknown.append(name)
                else:
                    _native.check(rc, "uvl_load_tensor(%s)" % name)
                # the D2D copy is enqueued on the current stream; torch's allocator is stream-ordered, so a
                # temporary `t` cannot be recycled before the copy has run
            torch.cuda.current_stream(self.device).synchronize()
            if strict and unknown:
                raise KeyError("unexpected tensors: %s" % unknown[:5])
            _native.check(self.lib.uvl_finalize_weights(self.handle, self._stream()), "uvl_finalize_weights")
        self.weights_loaded = True
        self._graph_io = None
        self._set_primed(None)
        return unknown

    # ------------------------------------------------------------------ forward
    def _workspace(self, B: int) -> torch.Tensor:
        if self._ws is None or self._ws_batch < B:
            n = self.lib.uvl_workspace_bytes(self.handle, B)
            self._ws = torch.empty(n + 256, dtype=torch.uint8, device=self.device)
            self._ws_batch = B
            self._graph_io = None
            self._set_primed(None)
        if self._ws_user_B != B:
            self._set_primed(None)                      # the workspace is carved per batch size: another B overwrites the kept rows
            self._ws_user_B = B
        return self._ws

    # The text branch below the first fusion layer depends on the text alone; `forward(..., reuse_text=True)` skips it when the
    # workspace still holds the rows of a call with the very same text tensors (same objects, not modified since).
    def _set_primed(self, key):
        """Record whose text rows the workspace holds (None: unknown).  A pre-packed step (make_eager_step) owns them only while
        nothing else has run on the workspace: `_txt_owner` is its token, cleared by every other user."""
        self._text_primed = key
        self._txt_owner = None

    @staticmethod
    def _text_key(B, ids, mask):
        return (B, ids, ids._version, mask, mask._version)

    def _text_matches(self, key) -> bool:
        p = self._text_primed
        return p is not None and p[0] == key[0] and p[1] is key[1] and p[2] == key[2] and p[3] is key[3] and p[4] == key[4]

    def _ws_ptr(self, ws):
        p = ws.data_ptr()
        return (p + 255) // 256 * 256

    def alloc_outputs(self, B: int, want=None) -> Dict[str, torch.Tensor]:
        names = want if want is not None else list(OUT_SHAPES)
        outs = {k: torch.empty(OUT_SHAPES[k](self.spec, B), dtype=torch.float32, device=self.device) for k in names}
        outs["argmax"] = torch.empty(B, dtype=torch.int64, device=self.device)
        return outs

    def _pack_io(self, template, search, ids, mask, prompt, flag, outs, skip_text, reuse_text=False):
        B = search.shape[0]
        i = _native.UvlInputs()
        i.batch = B
        i.d_template, i.d_search = template.data_ptr(), search.data_ptr()
        i.d_text_ids = ids.data_ptr() if ids is not None else None
        i.d_text_mask = mask.data_ptr() if mask is not None else None
        i.d_prompt, i.d_flag = prompt.data_ptr(), flag.data_ptr()
        i.skip_text = 1 if skip_text else 0
        i.reuse_text = 1 if reuse_text else 0
        o = _native.UvlOutputs()
        for k, f in _OUT_FIELD.items():
            setattr(o, f, outs[k].data_ptr() if k in outs else None)
        o.d_argmax = outs["argmax"].data_ptr() if "argmax" in outs else None
        return i, o

    def _canon_inputs(self, template, search, ids, mask, prompt, flag):
        s = self.spec
        for t, w in ((template, "template"), (search, "search"), (prompt, "prompt"), (flag, "flag")):
            _require_cuda(t, w)
        B = search.shape[0]
        if tuple(template.shape) != (B, 3, s.template_size, s.template_size) or tuple(search.shape) != (B, 3, s.search_size, s.search_size):
            raise ValueError("image shapes %s / %s do not match the model geometry (%d / %d)" %
                             (tuple(template.shape), tuple(search.shape), s.template_size, s.search_size))
        template = template.to(torch.float32).contiguous()
        search = search.to(torch.float32).contiguous()
        prompt = prompt.to(torch.float32).contiguous()
        if tuple(prompt.shape) != (B, 3, s.dim):
            raise ValueError("prompt must be [B,3,%d]" % s.dim)
        flag = flag.reshape(-1).to(torch.int64).contiguous()
        if flag.numel() != B:
            raise ValueError("flag must carry one entry per sample")
        if ids is not None:
            _require_cuda(ids, "text.tensors")
            ids = ids.to(torch.int64).contiguous()
            mask = (mask != 0).to(torch.uint8).contiguous()
            if tuple(ids.shape) != (B, s.text_len) or tuple(mask.shape) != (B, s.text_len):
                raise ValueError("text must be [B,%d]" % s.text_len)
        return template, search, ids, mask, prompt, flag

    def forward(self, template, search, ids, mask, prompt, flag, skip_text: bool = False, outs=None, profile: bool = False,
                reuse_text: bool = False):
        """UVLTrack.forward_test on device tensors; returns the output dict (f32 device tensors).  `reuse_text`: allow the text
        branch of an earlier call with the same `ids` / `mask` tensor objects (unmodified) to be reused (uvl_inputs.reuse_text)."""
        if not self.weights_loaded:
            raise _native.NativeLibraryError("weights have not been loaded")
        key = self._text_key(search.shape[0], ids, mask) if (ids is not None and mask is not None and not skip_text) else None
        template, search, ids, mask, prompt, flag = self._canon_inputs(template, search, ids, mask, prompt, flag)
        B = search.shape[0]
        if B > self.max_batch:
            raise ValueError("batch %d exceeds max_batch %d" % (B, self.max_batch))
        with torch.cuda.device(self.device):
            ws = self._workspace(B)
            if outs is None:
                outs = self.alloc_outputs(B)
            reuse = bool(reuse_text) and key is not None and self._text_matches(key)
            if key is not None:
                self._set_primed(key)                   # this call leaves (or keeps) that text's rows in the workspace
            i, o = self._pack_io(template, search, ids, mask, prompt, flag, outs, skip_text, reuse)
            n = self.lib.uvl_workspace_bytes(self.handle, B)
            if profile:
                ms = (C.c_float * _native.UVL_NFAM)()
                cnt = (C.c_int * _native.UVL_NFAM)()
                _native.check(self.lib.uvl_forward_test_profiled(self.handle, C.byref(i), C.byref(o), C.c_void_p(self._ws_ptr(ws)), n,
                                                                 self._stream(), ms, cnt), "uvl_forward_test_profiled")
            else:
                _native.check(self.lib.uvl_forward_test(self.handle, C.byref(i), C.byref(o), C.c_void_p(self._ws_ptr(ws)), n, self._stream()),
                              "uvl_forward_test")
        outs["flag"] = flag
        outs["prompt"] = prompt
        outs["prompts"] = prompt
        self._keep = (template, search, ids, mask)      # inputs must outlive the asynchronous launches
        return outs

    def make_eager_step(self, template, search, ids, mask, prompt, flag, skip_text: bool = False, outs=None, reuse_text: bool = False):
        """Pre-validate the inputs once and return a zero-argument callable that enqueues one frame on the SAME device buffers
        every time (the benchmark's steady-state loop, the tracker's per-frame call): no per-step Python work besides one ctypes
        call.  `reuse_text`: the step recomputes the text branch only when something else has used the workspace since its last
        call (first call included); otherwise it passes uvl_inputs.reuse_text."""
        template, search, ids, mask, prompt, flag = self._canon_inputs(template, search, ids, mask, prompt, flag)
        B = search.shape[0]
        reuse_text = bool(reuse_text) and not skip_text and ids is not None
        with torch.cuda.device(self.device):
            ws = self._workspace(B)
            if outs is None:
                outs = self.alloc_outputs(B)
            i, o = self._pack_io(template, search, ids, mask, prompt, flag, outs, skip_text, False)
            i_reuse, _ = self._pack_io(template, search, ids, mask, prompt, flag, outs, skip_text, True)
            n = self.lib.uvl_workspace_bytes(self.handle, B)
        outs["flag"], outs["prompt"], outs["prompts"] = flag, prompt, prompt
        keep = (template, search, ids, mask, prompt, flag, ws, outs, i, i_reuse, o)
        fwd, h, wsp, bo = self.lib.uvl_forward_test, self.handle, C.c_void_p(self._ws_ptr(ws)), C.byref(o)
        bi_full, bi_reuse = C.byref(i), C.byref(i_reuse)
        stream = self._stream()
        token = object()
        eng = self

        def step(_keep=keep):
            if eng._ws is not ws:
                raise _native.NativeLibraryError("the engine's workspace was reallocated (a larger batch ran): rebuild the step")
            if eng._ws_user_B != B:
                eng._set_primed(None)                     # another batch size carved the workspace differently in between
                eng._ws_user_B = B
            if reuse_text and eng._txt_owner is token:
                rc = fwd(h, bi_reuse, bo, wsp, n, stream)
            else:
                rc = fwd(h, bi_full, bo, wsp, n, stream)
                if not skip_text:
                    eng._set_primed(None)                 # this step's text rows are in the workspace now ...
                    eng._txt_owner = token                # ... and stay valid until anything else runs on it
            if rc < 0:
                _native.check(rc, "uvl_forward_test")
            return outs
        return step

    def forward_full(self, template, search, ids, mask, template_mask, context_mask, flag):
        """UVLTrack.forward (uvltrack.py:18-24) in eval mode: the head's no-prompt branch (head:123-138) with the prompter inline.
        Returns the output dict; `cont_score` is [B,S,2] and `prompts` the computed prompt [B,3,D]."""
        if not self.weights_loaded:
            raise _native.NativeLibraryError("weights have not been loaded")
        s = self.spec
        B = search.shape[0]
        dummy = torch.zeros(B, 3, s.dim, dtype=torch.float32, device=self.device)
        template, search, ids, mask, dummy, flag = self._canon_inputs(template, search, ids, mask, dummy, flag)
        if ids is None:
            raise ValueError("UVLTrack.forward needs the text input")
        for t, w in ((template_mask, "template_mask"), (context_mask, "context_mask")):
            _require_cuda(t, w)
        tm = (template_mask.reshape(B, -1) != 0).to(torch.uint8).contiguous()
        cm = (context_mask.reshape(B, -1) != 0).to(torch.uint8).contiguous()
        if tm.shape[1] != s.nz or cm.shape[1] != s.nx:
            raise ValueError("forward: mask shapes do not match the model geometry")
        if B > self.max_batch:
            raise ValueError("batch %d exceeds max_batch %d" % (B, self.max_batch))
        with torch.cuda.device(self.device):
            ws = self._workspace(B)
            outs = self.alloc_outputs(B)
            outs["cont_score"] = torch.empty(B, s.nx, 2, dtype=torch.float32, device=self.device)
            prompts = torch.empty(B, 3, s.dim, dtype=torch.float32, device=self.device)
            self._set_primed(None)                      # this call's text branch overwrites the kept rows
            i, o = self._pack_io(template, search, ids, mask, dummy, flag, outs, False)
            n = self.lib.uvl_workspace_bytes(self.handle, B)
            p = lambda t: C.c_void_p(t.data_ptr())
            _native.check(self.lib.uvl_forward(self.handle, C.byref(i), p(tm), p(cm), C.byref(o), p(prompts), C.c_void_p(self._ws_ptr(ws)), n,
                                               self._stream()), "uvl_forward")
        outs["flag"] = flag
        outs["prompts"] = prompts
        outs["template_mask"], outs["context_mask"] = template_mask, context_mask
        self._keep = (template, search, ids, mask, tm, cm, dummy)
        return outs

    def forward_prompt(self, out_dict, template_mask, context_mask) -> torch.Tensor:
        """UVLTrack.forward_prompt (uvltrack.py:33-38) on a forward_test output dict; returns prompt [B,3,D]."""
        need = ("template", "search", "vis_token", "txt_token", "flag")
        miss = [k for k in need if k not in out_dict]
        if miss:
            raise KeyError("out_dict lacks %s" % miss)
        s = self.spec
        tem = out_dict["template"].to(torch.float32).contiguous()
        ctx = out_dict["search"].to(torch.float32).contiguous()
        vis = out_dict["vis_token"].to(torch.float32).contiguous()
        txt = out_dict["txt_token"].to(torch.float32).contiguous()
        flag = out_dict["flag"].reshape(-1).to(torch.int64).contiguous()
        B = ctx.shape[0]
        for t, w in ((tem, "template"), (ctx, "search"), (template_mask, "template_mask"), (context_mask, "context_mask")):
            _require_cuda(t, w)
        tm = (template_mask.reshape(B, -1) != 0).to(torch.uint8).contiguous()
        cm = (context_mask.reshape(B, -1) != 0).to(torch.uint8).contiguous()
        if tuple(tem.shape) != (B, s.nz, s.dim) or tuple(ctx.shape) != (B, s.nx, s.dim) or tm.shape[1] != s.nz or cm.shape[1] != s.nx:
            raise ValueError("forward_prompt: tensor shapes do not match the model geometry")
        with torch.cuda.device(self.device):
            ws = self._workspace(B)
            n = self.lib.uvl_workspace_bytes(self.handle, B)
            prompt = torch.empty(B, 3, s.dim, dtype=torch.float32, device=self.device)
            p = lambda t: C.c_void_p(t.data_ptr())
            _native.check(self.lib.uvl_forward_prompt(self.handle, B, p(tem), p(ctx), p(vis), p(txt), p(flag), p(tm), p(cm), p(prompt),
                                                      C.c_void_p(self._ws_ptr(ws)), n, self._stream()), "uvl_forward_prompt")
        self._keep_prompt = (tem, ctx, vis, txt, flag, tm, cm)
        return prompt

    def decode(self, out_dict, window, state, resize_factor, image_hw, margin: float = 10.0, has_cont: bool = True, host_out=None):
        """Tracker post-processing on the device (tracker:116-125): returns (new_state [B,4] xywh, score [B], box_net [B,4], idx [B]).
        `host_out`: a pinned float32 CPU tensor of 9*B elements -- the kernel then writes new_state | score | box_net straight into
        host memory (valid after a stream synchronisation) and the returned tensors are views of it: no read-back copy."""
        cls = out_dict["cls_score_test"].to(torch.float32).contiguous()
        B = cls.shape[0]
        cont = out_dict["cont_score"].to(torch.float32).contiguous() if has_cont else None
        bbox = out_dict["bbox_map"].to(torch.float32).contiguous()
        f = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        window, state, resize_factor, image_hw = f(window).reshape(-1), f(state).reshape(B, 4), f(resize_factor).reshape(B), f(image_hw).reshape(B, 2)
        if host_out is not None:
            if not (host_out.is_pinned() and host_out.dtype == torch.float32 and host_out.is_contiguous() and host_out.numel() >= 9 * B):
                raise ValueError("host_out must be a pinned, contiguous float32 tensor with at least 9 * B elements")
            flat = host_out.reshape(-1)
            new_state, score, net = flat[:4 * B].view(B, 4), flat[4 * B:5 * B], flat[5 * B:9 * B].view(B, 4)
        else:
            new_state = torch.empty(B, 4, device=self.device)
            score = torch.empty(B, device=self.device)
            net = torch.empty(B, 4, device=self.device)
        idx = torch.empty(B, dtype=torch.int64, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(self.device):
            _native.check(self.lib.uvl_decode(self.handle, B, p(cls), p(cont), p(bbox), p(window), p(state), p(resize_factor), p(image_hw),
                                              C.c_float(margin), p(new_state), p(score), p(net), p(idx), self._stream()), "uvl_decode")
        self._keep_decode = (cls, cont, bbox, window, state, resize_factor, image_hw)
        return new_state, score, net, idx

    def profile_entries(self):
        """Per-launch-site breakdown of the last profile=True forward."""
        n = self.lib.uvl_profile_count(self.handle)
        res = []
        for k in range(n):
            name = C.create_string_buffer(96)
            kern = C.create_string_buffer(96)
            ms, fl, by, ln = C.c_double(), C.c_double(), C.c_double(), C.c_int()
            _native.check(self.lib.uvl_profile_entry(self.handle, k, name, kern, 96, C.byref(ms), C.byref(fl), C.byref(by), C.byref(ln)))
            wb = C.c_double()
            _native.check(self.lib.uvl_profile_entry_weight_bytes(self.handle, k, C.byref(wb)))
            res.append(dict(site=name.value.decode(), kernel=kern.value.decode(), ms=ms.value, flops=fl.value, bytes=by.value,
                            bytes_weights=wb.value, launches=ln.value))
        return res

    # ------------------------------------------------------------------ hipGraph replay
    def capture(self, template, search, ids, mask, prompt, flag, skip_text: bool = False):
        """Record the frame into a hipGraph bound to private copies of the inputs; returns (static_inputs, outputs)."""
        template, search, ids, mask, prompt, flag = self._canon_inputs(template, search, ids, mask, prompt, flag)
        B = search.shape[0]
        with torch.cuda.device(self.device):
            st = dict(template=template.clone(), search=search.clone(), ids=None if ids is None else ids.clone(),
                      mask=None if mask is None else mask.clone(), prompt=prompt.clone(), flag=flag.clone())
            outs = self.alloc_outputs(B)
            ws = self._workspace(B)
            self._set_primed(None)
            i, o = self._pack_io(st["template"], st["search"], st["ids"], st["mask"], st["prompt"], st["flag"], outs, skip_text)
            n = self.lib.uvl_workspace_bytes(self.handle, B)
            torch.cuda.synchronize(self.device)
            _native.check(self.lib.uvl_graph_capture(self.handle, C.byref(i), C.byref(o), C.c_void_p(self._ws_ptr(ws)), n), "uvl_graph_capture")
        outs["flag"], outs["prompt"], outs["prompts"] = st["flag"], st["prompt"], st["prompt"]
        self._graph_io = (st, outs)
        return st, outs

    def replay(self):
        if self._graph_io is None:
            raise _native.NativeLibraryError("no captured graph")
        self._set_primed(None)
        _native.check(self.lib.uvl_graph_launch(self.handle, self._stream()), "uvl_graph_launch")
        return self._graph_io[1]
