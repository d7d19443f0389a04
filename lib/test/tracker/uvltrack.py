"""The UVLTrack tracker (reference lib/test/tracker/uvltrack.py:20-233) on the device ops of this repository: the caller on
either side of the per-frame path.  Same public behaviour -- `UVLTrack(params, dataset_name)`, `initialize(image, info)`,
`track(image, info)` -> {"target_bbox": [x, y, w, h]}, modes NL / NLBBOX / BBOX (cfg.TEST.MODE), prompt update every
cfg.TEST.UPDATE_INTERVAL frames -- with every tensor step on the GPU:

    reference (per frame)                                   here
    cv2 crop + resize on the CPU, fp32 upload               uint8 crop window upload + fused crop/resize/normalise kernel
    network.forward_test (eager ATen)                       uvl_forward_test (HIP)
    three .cpu() copies + numpy argmax / box arithmetic     uvl_decode (one kernel), one 10-float read-back
    network.forward_prompt / forward_prompt_init / forward  uvl_forward_prompt / uvl_forward (HIP)

`params` carries what `lib/test/parameter/uvltrack.py:19-48` sets (cfg, template/search factor and size, grounding_size,
checkpoint); `network=` injects an already built model (tests, benchmarks) instead of reading params.checkpoint."""
import numpy as np
import torch

from lib.models.uvltrack.uvltrack import build_model
from lib.test.tracker.basetracker import BaseTracker
from lib.test.tracker.tracker_utils import Preprocessor_wo_mask
from lib.train.data.processing_utils import grounding_resize, sample_target
from lib.utils.box_ops import box_cxcywh_to_xywh, box_xywh_to_xyxy
from lib.utils.misc import NestedTensor
from uvltrack_amd.checkpoint import load_checkpoint
from uvltrack_amd.preprocess import WindowUploader, sample_target_fused
from uvltrack_amd.tokenizer import BertTokenizer, extract_token_from_nlp


class UVLTrack(BaseTracker):
    def __init__(self, params, dataset_name=None, network=None, tokenizer=None, device="cuda:0"):
        super(UVLTrack, self).__init__(params)
        self.device = torch.device(device)
        if network is None:
            network = build_model(params.cfg)
            load_checkpoint(network, self.params.checkpoint, strict=False)          # tracker:24
        self.cfg = params.cfg
        self.network = network.to(self.device)
        self.network.eval()
        # the sentence is fixed for a sequence: forward_test may keep the text branch of the previous frame (uvl_inputs.reuse_text)
        self.network.cache_text = True
        self.map_size = params.search_size // 16
        self.preprocessor = Preprocessor_wo_mask()
        self.state = None
        self.debug = getattr(self.params, "debug", 0)
        self.frame_id = 0
        self.update_interval = self.cfg.TEST.UPDATE_INTERVAL
        self.feat_size = self.params.search_size // 16
        self.tokenizer = tokenizer                                                   # built lazily: BBOX mode needs no vocabulary
        self.threshold = self.params.cfg.TEST.THRESHOLD
        self.has_cont = self.params.cfg.TRAIN.CONT_WEIGHT > 0
        self.max_score = 0
        self.max_query_len = self.cfg.MODEL.BACKBONE.LANGUAGE.BERT.MAX_QUERY_LEN
        self._uploader = None
        self._meta_host = None
        self._res_host = None
        self._step = None            # forward_test bound to this sequence's buffers (model.make_frame_step)
        self._step_key = None
        self._search_buf = None
        self._prompt_buf = None
        self._prompt_src = None

    # ------------------------------------------------------------------ helpers
    def _tok(self):
        if self.tokenizer is None:
            self.tokenizer = BertTokenizer.from_pretrained(self.cfg.MODEL.BACKBONE.LANGUAGE.VOCAB_PATH, do_lower_case=True)   # tracker:40
        return self.tokenizer

    def extract_token_from_nlp(self, nlp, seq_length):
        """tracker:196-233 -> (ids [1,L], mask [1,L]) on the device."""
        ids, mask = extract_token_from_nlp(self._tok(), nlp, seq_length)
        return torch.tensor(ids).unsqueeze(0).to(self.device), torch.tensor(mask).unsqueeze(0).to(self.device)

    def window_prior(self):
        """tracker:64-68 (only the numpy window is used by track())."""
        hanning = np.hanning(self.map_size)
        self.window = np.outer(hanning, hanning).flatten()
        self._window_dev = torch.from_numpy(self.window.astype(np.float32)).to(self.device)

    def anno2mask(self, gt_bboxes, size):
        """Target-cell mask [b, size*size] (bool, on the device) of normalised xywh boxes -- the reference's helper of the same
        name (tracker:183-194) as one small kernel (uvl_anno2mask): a box that is already on the device stays there."""
        from uvltrack_amd import _native
        import ctypes as C
        boxes = torch.as_tensor(gt_bboxes, dtype=torch.float32).reshape(-1, 4).to(self.device, non_blocking=True).contiguous()
        mask = torch.empty((boxes.shape[0], size * size), dtype=torch.uint8, device=self.device)
        _native.check(_native.load().uvl_anno2mask(C.c_void_p(boxes.data_ptr()), boxes.shape[0], int(size), C.c_void_p(mask.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "uvl_anno2mask")
        return mask.bool()

    def map_box_back(self, pred_box: list, resize_factor: float):
        """tracker:167-173."""
        cx_prev, cy_prev = self.state[0] + 0.5 * self.state[2], self.state[1] + 0.5 * self.state[3]
        cx, cy, w, h = pred_box
        half_side = 0.5 * self.params.search_size / resize_factor
        return [cx + (cx_prev - half_side) - 0.5 * w, cy + (cy_prev - half_side) - 0.5 * h, w, h]

    def _search_image(self, image, box, factor, size, with_meta=False, image_out=None):
        """Normalised crop [1,3,size,size] + resize factor: crop window upload for host frames, full-frame kernel for device frames.
        `with_meta`: also the decode kernel's operands (box, resize factor, frame size) as a device tensor [7] -- for host frames
        they ride in the window upload."""
        if isinstance(image, np.ndarray):
            if self._uploader is None:
                self._uploader = WindowUploader(device=self.device)
            r = self._uploader.sample_target(image, box, factor, size, with_meta=with_meta, image_out=image_out)
        else:
            r = sample_target_fused(image, box, factor, size, want_patch=False, want_mask=False, image_out=image_out)
        if not with_meta:
            return r["image"], r["resize_factor"]
        meta = r.get("meta")
        if meta is None:
            H, W = image.shape[:2]
            if self._meta_host is None:
                self._meta_host = torch.empty(7, dtype=torch.float32).pin_memory()
            self._meta_host.copy_(torch.tensor([float(v) for v in box] + [float(r["resize_factor"]), float(H), float(W)], dtype=torch.float32))
            meta = self._meta_host.to(self.device, non_blocking=True)
        return r["image"], r["resize_factor"], meta

    def _frame_step(self):
        """forward_test of this sequence as a pre-validated step on fixed buffers: the search crop is written into
        `_search_buf` by the pre-processing kernel, the prompt lives in `_prompt_buf` (refreshed in place when it changes)."""
        size = self.params.search_size
        if self._search_buf is None or self._search_buf.shape[-1] != size:
            self._search_buf = torch.empty(1, 3, size, size, dtype=torch.float32, device=self.device)
            self._step = None
        if self._prompt_buf is None or self._prompt_buf.shape != self.prompt.shape:
            self._prompt_buf = torch.empty_like(self.prompt, dtype=torch.float32)
            self._step = None
        key = (self.template, self.text, self.flag)
        if self._step is None or any(a is not b for a, b in zip(key, self._step_key)):
            self._step = self.network.make_frame_step(self.template, self._search_buf, self.text, self._prompt_buf, self.flag)
            self._step_key = key
            self._prompt_src = None
        if self._prompt_src is not self.prompt:          # new prompt (initialisation, update every UPDATE_INTERVAL frames)
            self._prompt_buf.copy_(self.prompt)
            self._prompt_src = self.prompt
        return self._step

    # ------------------------------------------------------------------ tracker:45-62
    def grounding(self, image, info: dict):
        h, w = image.shape[:2]
        bbox = torch.tensor([0., 0., 0., 0.])
        im_crop_padded = grounding_resize(image, self.params.grounding_size, bbox, None)[0]
        ground = self.preprocessor.process(im_crop_padded)
        template = torch.zeros([1, 3, self.params.template_size, self.params.template_size], device=self.device)
        template_mask = torch.zeros([1, (self.params.template_size // 16) ** 2], device=self.device).bool()
        context_mask = torch.zeros([1, (self.params.search_size // 16) ** 2], device=self.device).bool()
        text, mask = self.extract_token_from_nlp(info['language'], self.max_query_len)
        self.text = NestedTensor(text, mask)
        flag = torch.tensor([[1]], device=self.device)
        with torch.no_grad():
            out_dict = self.network.forward(template, ground, self.text, template_mask, context_mask, flag)
        out_dict['pred_boxes'] = box_cxcywh_to_xywh(out_dict['pred_boxes'] * np.max(image.shape[:2]))[0, 0].cpu().tolist()
        dx, dy = min(0, (w - h) / 2), min(0, (h - w) / 2)
        out_dict['pred_boxes'][0] = out_dict['pred_boxes'][0] + dx
        out_dict['pred_boxes'][1] = out_dict['pred_boxes'][1] + dy
        return out_dict

    # ------------------------------------------------------------------ tracker:70-108
    def initialize(self, image, info: dict):
        L = self.max_query_len
        if self.cfg.TEST.MODE == 'NL':
            grounding_state = self.grounding(image, info)
            init_bbox = grounding_state['pred_boxes']
            self.flag = torch.tensor([[2]], device=self.device)
        elif self.cfg.TEST.MODE == 'NLBBOX':
            text, mask = self.extract_token_from_nlp(info['language'], L)
            self.text = NestedTensor(text, mask)
            init_bbox = info['init_bbox']
            self.flag = torch.tensor([[2]], device=self.device)
        else:
            self.text = NestedTensor(torch.zeros([1, L], device=self.device).long(), torch.zeros([1, L], device=self.device))
            init_bbox = info['init_bbox']
            self.flag = torch.tensor([[0]], device=self.device)
        self.window_prior()
        z_patch_arr, _, _, bbox = sample_target(image, init_bbox, self.params.template_factor, output_sz=self.params.template_size,
                                                return_bbox=True)
        self.template_mask = self.anno2mask(bbox.reshape(1, 4), size=self.params.template_size // 16)
        self.z_patch_arr = z_patch_arr
        self.template_bbox = (bbox * self.params.template_size)[0, 0].tolist()
        self.template = self.preprocessor.process(z_patch_arr)
        # forward the context once
        y_patch_arr, _, _, y_bbox = sample_target(image, init_bbox, self.params.search_factor, output_sz=self.params.search_size,
                                                  return_bbox=True)
        self.y_patch_arr = y_patch_arr
        self.context_bbox = (y_bbox * self.params.search_size)[0, 0].tolist()
        context = self.preprocessor.process(y_patch_arr)
        context_mask = self.anno2mask(y_bbox.reshape(1, 4), self.params.search_size // 16)
        self.prompt = self.network.forward_prompt_init(self.template, context, self.text, self.template_mask, context_mask, self.flag)
        self.state = [float(v) for v in init_bbox]
        self.frame_id = 0
        self.max_score = 0

    # ------------------------------------------------------------------ tracker:110-140
    def track(self, image, info: dict = None):
        H, W, _ = image.shape
        self.frame_id += 1
        # the decode operands (previous state, resize factor, frame size) travel with the crop window: one upload per frame
        step = self._frame_step()
        search, resize_factor, meta = self._search_image(image, self.state, self.params.search_factor, self.params.search_size, with_meta=True,
                                                         image_out=self._search_buf)
        with torch.no_grad():
            out_dict = step()        # = self.network.forward_test(self.template, search, self.text, self.prompt, self.flag) on fixed buffers
            # argmax of cls * hann * softmax(cont)[0], box back to the frame, clip (tracker:116-125): one kernel that writes its nine
            # floats straight into pinned host memory -- one stream synchronisation per frame, no read-back copy
            if self._res_host is None:
                self._res_host = torch.zeros(9, dtype=torch.float32).pin_memory()
            _, _, _, idx = self.network.decode(out_dict, self._window_dev, meta[:4].reshape(1, 4), meta[4:5], meta[5:7].reshape(1, 2),
                                               margin=10.0, has_cont=self.has_cont, host_out=self._res_host)
            torch.cuda.current_stream(self.device).synchronize()
            host = self._res_host.clone()
        self.state = [float(v) for v in host[:4]]
        score = float(host[4])
        pred_box_net = host[5:9].clone()
        self.last_index = idx

        if score > self.max_score and self.has_cont:
            self.pred_box_net = pred_box_net
            # the step hands out the same output tensors every frame: keep copies of what forward_prompt will read
            self.out_dict = {k: (out_dict[k].clone() if k in ("template", "search", "vis_token", "txt_token") else out_dict[k])
                             for k in ("template", "search", "vis_token", "txt_token", "flag")}
            self.max_score = score
            self.best_frame = self.frame_id

        if self.frame_id % self.update_interval == 0 and self.has_cont and self.max_score > self.threshold:
            context_bbox = box_cxcywh_to_xywh(self.pred_box_net.reshape(1, 4))
            context_mask = self.anno2mask(context_bbox, self.params.search_size // 16)
            self.context_bbox = (context_bbox[0] * self.params.search_size).detach().cpu().tolist()
            self.prompt = self.network.forward_prompt(self.out_dict, self.template_mask, context_mask)
            self.max_score = 0

        return {"target_bbox": self.state}


def get_tracker_class():
    return UVLTrack
