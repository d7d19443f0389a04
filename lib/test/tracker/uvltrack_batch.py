"""Several sequences tracked in lockstep on one GPU: the per-GPU unit of the multi-GPU layout of SURVEY.md section 8e
(64 sequences = 8 GPUs x 8 sequences; the reference runs one sequence per worker process instead,
lib/test/evaluation/running.py:96-100,168-171).  Frame t of every sequence goes through ONE batched `forward_test` and ONE
batched decode kernel; the per-sequence logic (crop geometry, prompt update rule, max-score bookkeeping) is that of
lib/test/tracker/uvltrack.py:110-140, applied per sample.  Initialisation reuses the single-sequence tracker."""
import numpy as np
import torch

from lib.test.tracker.uvltrack import UVLTrack
from lib.utils.box_ops import box_cxcywh_to_xywh
from lib.utils.misc import NestedTensor
from uvltrack_amd.preprocess import WindowUploader, sample_target_fused


class BatchUVLTrack(object):
    def __init__(self, params, n_sequences: int, network=None, tokenizer=None, device="cuda:0"):
        self.single = UVLTrack(params, None, network=network, tokenizer=tokenizer, device=device)
        self.params, self.cfg, self.network, self.device = params, params.cfg, self.single.network, self.single.device
        self.B = int(n_sequences)
        self.update_interval = self.single.update_interval
        self.threshold, self.has_cont = self.single.threshold, self.single.has_cont
        self._uploaders = [WindowUploader(device=self.device) for _ in range(self.B)]
        self.frame_id = 0

    def initialize(self, images, infos):
        """images / infos: one per sequence (same contract as UVLTrack.initialize).  Modes may differ per sequence only through
        cfg.TEST.MODE of the shared params, as in the reference."""
        assert len(images) == self.B and len(infos) == self.B
        tem, txt, msk, prm, flg, tmask = [], [], [], [], [], []
        self.state = []
        for im, info in zip(images, infos):
            self.single.initialize(im, info)
            tem.append(self.single.template)
            txt.append(self.single.text.tensors)
            msk.append(self.single.text.mask)
            prm.append(self.single.prompt)
            flg.append(self.single.flag.reshape(1, 1))
            tmask.append(self.single.template_mask)
            self.state.append(list(self.single.state))
        self.template = torch.cat(tem, 0).contiguous()
        self.text = NestedTensor(torch.cat(txt, 0).contiguous(), torch.cat(msk, 0).contiguous())
        self.prompt = torch.cat(prm, 0).contiguous()
        self.flag = torch.cat(flg, 0).contiguous()
        self.template_mask = torch.cat(tmask, 0).contiguous()
        self._window_dev = self.single._window_dev
        S = self.params.search_size
        self.search = torch.empty(self.B, 3, S, S, dtype=torch.float32, device=self.device)
        self.max_score = [0.0] * self.B
        self.pred_box_net = [None] * self.B
        self.best = None                           # per-sample copy of the best frame's tokens (what forward_prompt reads)
        self.frame_id = 0
        # forward_test of the whole batch as a pre-validated step on these buffers (`search` is rewritten by the crop kernels,
        # `prompt` is updated in place); decode operands go up in one pinned upload, its results come back through pinned memory
        self._step = self.network.make_frame_step(self.template, self.search, self.text, self.prompt, self.flag)
        self._meta_host = torch.empty(self.B, 7, dtype=torch.float32).pin_memory()
        self._res_host = torch.zeros(9 * self.B, dtype=torch.float32).pin_memory()

    def _keep_best(self, out_dict, rows):
        if self.best is None:
            self.best = {k: out_dict[k].clone() for k in ("template", "search", "vis_token", "txt_token")}
            return
        idx = torch.tensor(rows, device=self.device)
        for k in self.best:
            self.best[k].index_copy_(0, idx, out_dict[k].index_select(0, idx))

    def track(self, images):
        assert len(images) == self.B
        self.frame_id += 1
        S, fac = self.params.search_size, self.params.search_factor

        def crop(b):                               # one fused crop/resize/normalise launch per sequence, straight into the batch buffer
            im = images[b]
            if isinstance(im, np.ndarray):
                r = self._uploaders[b].sample_target(im, self.state[b], fac, S, image_out=self.search[b])
            else:
                r = sample_target_fused(im, self.state[b], fac, S, want_patch=False, want_mask=False, image_out=self.search[b])
            return r["resize_factor"]
        # (a thread pool over the crops measured slower and far noisier than this loop: 2.0-2.9 vs 2.0 ms per step of 8)
        resize = [crop(b) for b in range(self.B)]
        meta = self._meta_host
        meta[:, :4] = torch.tensor(self.state, dtype=torch.float32)
        meta[:, 4] = torch.tensor(resize, dtype=torch.float32)
        meta[:, 5:] = torch.tensor([[float(im.shape[0]), float(im.shape[1])] for im in images])
        with torch.no_grad():
            meta_dev = meta.to(self.device, non_blocking=True)
            out_dict = self._step()                # = forward_test(self.template, self.search, self.text, self.prompt, self.flag)
            self.network.decode(out_dict, self._window_dev, meta_dev[:, :4], meta_dev[:, 4], meta_dev[:, 5:7], margin=10.0,
                                has_cont=self.has_cont, host_out=self._res_host)
            torch.cuda.current_stream(self.device).synchronize()
            B = self.B
            r = self._res_host
            host = torch.cat([r[:4 * B].view(B, 4), r[4 * B:5 * B].view(B, 1), r[5 * B:9 * B].view(B, 4)], dim=1)   # [B, 9]
        better = []
        for b in range(self.B):
            self.state[b] = [float(v) for v in host[b, :4]]
            sc = float(host[b, 4])
            if sc > self.max_score[b] and self.has_cont:
                self.max_score[b], self.pred_box_net[b] = sc, host[b, 5:9].clone()
                better.append(b)
        if better:
            self._keep_best(out_dict, better)
        if self.frame_id % self.update_interval == 0 and self.has_cont and self.best is not None:
            due = [b for b in range(self.B) if self.max_score[b] > self.threshold]
            if due:
                boxes = torch.stack([self.pred_box_net[b] if self.pred_box_net[b] is not None else torch.zeros(4) for b in range(self.B)])
                context_mask = self.single.anno2mask(box_cxcywh_to_xywh(boxes), S // 16)
                best = dict(self.best)
                best["flag"] = self.flag.reshape(-1)
                new_prompt = self.network.forward_prompt(best, self.template_mask, context_mask)
                idx = torch.tensor(due, device=self.device)
                self.prompt.index_copy_(0, idx, new_prompt.index_select(0, idx))
                for b in due:
                    self.max_score[b] = 0.0
        return [{"target_bbox": list(s)} for s in self.state]
