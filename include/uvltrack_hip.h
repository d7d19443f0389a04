/*
 * uvltrack_hip.h -- C ABI of the MI355X (gfx950) implementation of UVLTrack's per-frame
 * forward pass, `UVLTrack.forward_test` (reference lib/models/uvltrack/uvltrack.py:41-45).
 *
 * The reference has no FFI: its hot path is eager PyTorch.  This header is the boundary a
 * maintainer binds instead of `self.backbone(...)` + `self.box_head(...)`; each entry point
 * cites the reference interface it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer into caller-owned memory (e.g. tensor.data_ptr())
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream)
 *   - all calls are asynchronous on `stream`; nothing here synchronises the device except
 *     uvl_create / uvl_finalize_weights / uvl_destroy / uvl_graph_capture
 *   - return 0 on success, negative UVL_E* on error; uvl_last_error() gives a thread-local message
 *   - no call allocates device memory per frame: activations live in a caller-provided workspace
 *     of uvl_workspace_bytes() bytes; weights are packed once into library-owned memory
 *   - one handle per device per process; calls on one handle must be serialised by the caller
 */
#ifndef UVLTRACK_HIP_H
#define UVLTRACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVL_OK            0
#define UVL_EINVAL       -1   /* bad argument / unsupported geometry */
#define UVL_ESTATE       -2   /* call order violated (e.g. forward before finalize) */
#define UVL_EHIP         -3   /* a HIP runtime call failed */
#define UVL_ENOTFOUND    -4   /* unknown tensor name */

#define UVL_MAX_LAYERS   64

/* Geometry of the model; mirrors uvltrack_amd.spec.ModelSpec, i.e. the yaml keys the reference
 * reads in ModalityUnifiedFeatureExtractor.__init__ (extractor.py:12-41) and
 * build_modality_adaptive_box_head (heads/__init__.py:4-13). */
typedef struct uvl_config {
    int32_t dim;              /* MODEL.HIDDEN_DIM: 768 (B) / 1024 (L); multiple of 64 */
    int32_t heads;            /* 12 / 16; dim/heads must be 64 */
    int32_t depth;            /* 12 / 24 */
    int32_t n_fusion_start;   /* min(FUSION_LAYER); layers [n_fusion_start, depth) are joint */
    int32_t n_cont;           /* len(CONT_LOSS_LAYER) */
    int32_t cont_layers[UVL_MAX_LAYERS];
    int32_t template_size;    /* DATA.TEMPLATE.SIZE (pixels, multiple of 16) */
    int32_t search_size;      /* DATA.SEARCH.SIZE */
    int32_t text_len;         /* BERT.MAX_QUERY_LEN (<= 64) */
    int32_t head_dim;         /* MODEL.HEAD.HEAD_DIM (multiple of 256) */
    int32_t vocab;
    int32_t max_pos;
    int32_t txt_token_mean;   /* TXT_TOKEN_MODE == 'mean' */
    int32_t cls_tokenize;     /* MODEL.HEAD.CLS_TOKENIZE */
    int32_t offset_sigmoid;   /* MODEL.HEAD.OFFSET_SIGMOID */
    int32_t joint_cls;        /* MODEL.HEAD.JOINT_CLS */
    int32_t softmax_one;      /* MODEL.HEAD.SOFTMAX_ONE */
    int32_t max_batch;        /* largest batch a forward call may carry */
} uvl_config;

/* Inputs of forward_test (uvltrack.py:41): device tensors, contiguous. */
typedef struct uvl_inputs {
    int32_t batch;
    const float*   d_template;   /* [B,3,Hz,Hz] f32 NCHW, already mean/std normalised */
    const float*   d_search;     /* [B,3,Hx,Hx] f32 */
    const int64_t* d_text_ids;   /* [B,T] i64   (NestedTensor.tensors) */
    const uint8_t* d_text_mask;  /* [B,T] u8/bool, 1 = real token (NestedTensor.mask) */
    const float*   d_prompt;     /* [B,3,D] f32 */
    const int64_t* d_flag;       /* [B] i64 in {0 BBOX, 1 NL, 2 NLBBOX} */
    int32_t skip_text;           /* 1: all flags are 0 and the caller does not need `text`/`txt_token`:
                                    run the visual stream only (exact for every box output) */
    int32_t reuse_text;          /* 1: d_text_ids / d_text_mask are the ones of the last call that ran with reuse_text = 0 on this
                                    workspace at this batch size: the text branch below the first fusion layer (BERT embedding +
                                    layers, extractor.py:54,62) is a function of the text alone, so its rows are taken from the
                                    workspace instead of being recomputed (bit-identical outputs).  Not a reference feature: a
                                    tracker's sentence is fixed for a sequence (lib/test/tracker/uvltrack.py:57,114). */
} uvl_inputs;

/* Output dict of forward_test: caller-allocated f32 device tensors.
 * Any pointer may be NULL to skip that output. */
typedef struct uvl_outputs {
    float* d_search;        /* [B,S,D]   */
    float* d_template;      /* [B,nz,D]  */
    float* d_text;          /* [B,T,D]   */
    float* d_vis_token;     /* [B,1,D]   */
    float* d_txt_token;     /* [B,1,D]   */
    float* d_logits;        /* [B,n_cont,F,F] */
    float* d_cls_score;     /* [B,F,F]  (== cls_score_test unless joint_cls) */
    float* d_cls_score_test;/* [B,F,F]   */
    float* d_bbox_map;      /* [B,S,4]   */
    float* d_pred_boxes;    /* [B,1,4]   */
    float* d_cont_score;    /* [B,S,3] (softmax_one) or [B,S,2] */
    int64_t* d_argmax;      /* [B] index of pred_boxes in bbox_map (extra, for tie-aware checks) */
} uvl_outputs;

typedef struct uvl_model uvl_model_t;

const char* uvl_last_error(void);
int  uvl_version(void);
/* The `hipcc --version` line of the compiler that built the library.  Several kernels count their s_waitcnt by hand around inline-asm /
 * generated loops and are only known-good for the toolchain uvltrack_amd/build.py names (TESTED_HIPCC); the Python binding refuses a library
 * built by another one unless UVL_ALLOW_UNTESTED_HIPCC=1. */
const char* uvl_build_toolchain(void);

/* build_model(cfg) (uvltrack.py:47-57): create an empty model for a geometry. */
uvl_model_t* uvl_create(const uvl_config* cfg);
void uvl_destroy(uvl_model_t* m);

/* load_state_dict (SURVEY.md 8b weight contract): hand one f32 tensor of the reference's state_dict
 * to the library by its reference name.  `d_data` is a device pointer, dims are the tensor's shape.
 * Tensors forward_test never reads (prompter.*, pooler.*, vit.norm.*, num_batches_tracked) are
 * accepted and ignored (return 1).  Unknown names return UVL_ENOTFOUND (strict=False callers ignore it). */
int uvl_load_tensor(uvl_model_t* m, const char* name, const float* d_data, int ndim, const int64_t* dims, void* stream);

/* Pack everything for the kernels: bf16 GEMM weights, BERT q/k/v concatenation, BatchNorm folded into
 * the conv towers (eval semantics, eps 1e-5), conv weights re-laid to [Cout][tap][Cin].
 * Fails with UVL_ESTATE listing the first missing tensor. */
int uvl_finalize_weights(uvl_model_t* m, void* stream);

size_t uvl_workspace_bytes(const uvl_model_t* m, int batch);

/* UVLTrack.forward_test (uvltrack.py:41-45).  Enqueues the whole frame on `stream`. */
int uvl_forward_test(uvl_model_t* m, const uvl_inputs* in, const uvl_outputs* out,
                     void* d_workspace, size_t workspace_bytes, void* stream);

/* UVLTrack.forward_prompt (uvltrack.py:33-38): the prompter = ModalityAdaptiveBoxHead.forward_prompt (head:96-106) ->
 * DistributionBasedCrossAttention.forward (heads/utils.py:82-99).  Inputs are entries of a previous forward_test output
 * dict (template [B,nz,D], search [B,S,D], vis_token / txt_token [B,1,D], flag [B]) plus the target-cell masks
 * (u8/bool, 1 = cell inside the box; tracker anno2mask, lib/test/tracker/uvltrack.py:183-194).
 * Output: prompt [B,3,D] f32 = (target, distractor, background) tokens; flag 1 (grounding) returns the un-updated queries.
 * Needs box_head.prompter.{logit_scale, query_embed.weight, mlp.fc1.*, mlp.fc2.*} to have been loaded; uses the frame
 * workspace as scratch (call it between frames, on the same stream). */
int uvl_forward_prompt(uvl_model_t* m, int batch, const float* d_template_tokens, const float* d_search_tokens,
                       const float* d_vis_token, const float* d_txt_token, const int64_t* d_flag,
                       const uint8_t* d_template_mask, const uint8_t* d_context_mask, float* d_prompt_out,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* Target-cell masks for uvl_forward_prompt / uvl_forward: the tracker's anno2mask (lib/test/tracker/uvltrack.py:183-194).
 * d_boxes_xywh [batch,4] f32 normalised to [0,1] (device), size = grid side (template_size/16 or search_size/16)
 * -> d_mask [batch, size*size] u8: 1 where the cell centre lies strictly inside the box, and on the cell of the box centre. */
int uvl_anno2mask(const float* d_boxes_xywh, int batch, int size, uint8_t* d_mask, void* stream);

/* Per-frame post-processing of the tracker on the device (lib/test/tracker/uvltrack.py:116-125: argmax of
 * cls_score_test * hann window * softmax(cont_score)[0]; box scaled by search_size / resize_factor; map_box_back :167-173;
 * clip_box lib/utils/box_ops.py:117-126 with `margin`).  d_cont_score may be NULL (TRAIN.CONT_WEIGHT == 0).
 * d_window [S] f32 (np.outer(np.hanning(F), np.hanning(F))), d_state [B,4] previous box (x,y,w,h), d_resize_factor [B],
 * d_image_hw [B,2] = (H, W).  Outputs: d_new_state [B,4]; optional d_score [B], d_box_net [B,4], d_index [B]. */
int uvl_decode(uvl_model_t* m, int batch, const float* d_cls_score_test, const float* d_cont_score, const float* d_bbox_map,
               const float* d_window, const float* d_state, const float* d_resize_factor, const float* d_image_hw,
               float margin, float* d_new_state, float* d_score, float* d_box_net, int64_t* d_index, void* stream);

/* UVLTrack.forward (lib/models/uvltrack/uvltrack.py:18-24) in eval mode -- the call of the tracker's grounding() at sequence
 * init in NL mode (lib/test/tracker/uvltrack.py:45-62).  Same backbone as uvl_forward_test; the head takes
 * its no-prompt branch (modality_adaptive_box_head.py:123-138): the prompter runs inline on the search tokens rolled by
 * batch/2 (d_template_mask [B,nz], d_context_mask [B,S], 1 = target cell), `out->d_cont_score` is [B,S,2] and the computed
 * prompt [B,3,D] ("prompts" of the reference's dict) goes to d_prompts_out.  out->d_search / d_template / d_vis_token /
 * d_txt_token are required; in->d_prompt is ignored.  Needs the prompter weights. */
int uvl_forward(uvl_model_t* m, const uvl_inputs* in, const uint8_t* d_template_mask, const uint8_t* d_context_mask,
                const uvl_outputs* out, float* d_prompts_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* Per-frame pre-processing on the device.  Replaces `sample_target(im, bb, factor, output_sz)`
 * (lib/train/data/processing_utils.py:159-243: square crop of side ceil(sqrt(w*h)*factor) around the box, zero border,
 * cv2.resize to output_sz, attention mask of the border) followed by `Preprocessor_wo_mask.process`
 * (lib/test/tracker/tracker_utils.py:20-29: (x/255 - mean)/std, HWC -> NCHW), as called at
 * lib/test/tracker/uvltrack.py:89-101,110-112.  d_image: uint8 HWC frame on the device (row_stride_bytes >= 3*width).
 * Outputs (each optional, at least one): d_patch_hwc [out,out,3] uint8 (what sample_target returns), d_norm_chw
 * [3,out,out] f32 (what the preprocessor returns, the layout uvl_forward_test consumes), d_att_mask [out,out] 0/1.
 * The resize follows OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic (see oracle/preprocess_oracle.py).
 * `geometry_out` (host, optional) receives the crop geometry incl. resize_factor = output_sz / crop_sz. */
typedef struct uvl_crop_geometry {
    int32_t crop_sz, x1, y1, x1_pad, x2_pad, y1_pad, y2_pad;
    float resize_factor;
} uvl_crop_geometry;
int uvl_crop_geometry_of(const float box_xywh[4], float search_area_factor, int output_sz, int height, int width,
                         uvl_crop_geometry* geometry_out);      /* host arithmetic only (processing_utils.py:173-193) */
int uvl_sample_target(const uint8_t* d_image, int height, int width, int row_stride_bytes, const float box_xywh[4],
                      float search_area_factor, int output_sz, uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask,
                      uvl_crop_geometry* geometry_out, void* stream);
/* Same, when only a window of the frame was uploaded: d_window holds frame pixels [win_y0, win_y0+win_height) x
 * [win_x0, win_x0+win_width) and must cover the part of the crop that lies inside the frame (uvl_crop_geometry_of tells
 * the host which part that is, so it can upload ~crop_sz^2 * 3 bytes instead of the frame). */
int uvl_sample_target_window(const uint8_t* d_window, int win_x0, int win_y0, int win_width, int win_height, int row_stride_bytes,
                             int frame_height, int frame_width, const float box_xywh[4], float search_area_factor, int output_sz,
                             uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask, uvl_crop_geometry* geometry_out, void* stream);
/* The window form with the upload included: the host has gathered the window rows (win_width * 3 bytes each, no padding) into
 * pinned memory at h_stage + header_bytes; header_bytes + window bytes are copied to d_stage on `stream` in ONE transfer and the
 * kernel then reads the window at d_stage + header_bytes.  The header (a multiple of 16 bytes, may be 0) is the caller's: the
 * tracker puts the seven floats of uvl_decode's operands there so they need no upload of their own. */
int uvl_sample_target_staged(const uint8_t* h_stage, uint8_t* d_stage, size_t header_bytes, int win_x0, int win_y0, int win_width,
                             int win_height, int frame_height, int frame_width, const float box_xywh[4], float search_area_factor,
                             int output_sz, uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask, uvl_crop_geometry* geometry_out,
                             void* stream);
/* grounding_resize (lib/train/data/processing_utils.py:60-141; tracker:48): the whole frame resized with its aspect ratio
 * kept (long side = output_sz, OpenCV INTER_LINEAR -- see oracle/preprocess_oracle.py::grounding_resize on the reference's
 * swallowed `interpolation` argument), centred on a zero canvas; d_att_mask = 1 on the padding.
 * image_top_coords (host, optional) = [x1_pad, y1_pad, new_w, new_h]. */
int uvl_grounding_resize(const uint8_t* d_image, int height, int width, int row_stride_bytes, int output_sz,
                         uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask, int32_t image_top_coords[4], void* stream);
/* Preprocessor_wo_mask.process alone, on an already cropped uint8 HWC patch. */
int uvl_normalize_u8(const uint8_t* d_patch_hwc, int height, int width, float* d_norm_chw, void* stream);

/* hipGraph replay of the same call: capture once for fixed pointers/batch, then launch per frame. */
int uvl_graph_capture(uvl_model_t* m, const uvl_inputs* in, const uvl_outputs* out,
                      void* d_workspace, size_t workspace_bytes);
int uvl_graph_launch(uvl_model_t* m, void* stream);
int uvl_graph_release(uvl_model_t* m);

/* Per-kernel timing of the last eager uvl_forward_test_profiled call: runs the frame with a HIP event
 * pair around every launch (on the launching stream) and returns accumulated milliseconds per kernel
 * family.  Families: 0 gemm, 1 attention, 2 layernorm, 3 conv, 4 other. */
#define UVL_NFAM 5
int uvl_forward_test_profiled(uvl_model_t* m, const uvl_inputs* in, const uvl_outputs* out,
                              void* d_workspace, size_t workspace_bytes, void* stream,
                              float ms_per_family[UVL_NFAM], int launches_per_family[UVL_NFAM]);

/* Per-launch-site breakdown of the last uvl_forward_test_profiled call: `name` = launch site
 * ("gemm.fc1", "attention", ...), `kernel` = the kernel instantiation that ran, `flops`/`bytes` =
 * algorithmic work summed over the site's launches. */
int uvl_profile_count(const uvl_model_t* m);
int uvl_profile_entry(const uvl_model_t* m, int i, char* name, char* kernel, int name_cap,
                      double* ms, double* flops, double* bytes, int* launches);
/* The same entry's bytes on SURVEY section 8(d)'s count -- weights touched once, activations cache-resident (GEMM / conv sites: the
 * weight; attention: q, k, v in + o out; row kernels: as `bytes`) -- beside `bytes`, which also counts what the epilogue moves. */
int uvl_profile_entry_weight_bytes(const uvl_model_t* m, int i, double* bytes);

/* Test hooks.  key "stop_layer": value >= 0 makes the next forwards leave the layer loop after that ViT layer
 * (the head still runs on that state) so parity tests can localise an error to a layer; -2 runs NO layer (the outputs `search` / `template` / `vis_token` / `text` are then
 * the patch embedding + position table + [cls] row and the BERT embedding: the input side alone); -1 restores normal runs.
 * "pair_text" (default 1): where the text branch rides in the visual launches instead of running on a second stream -- 0 never,
 * 1 one-sequence frames and the many-sequence frames it measured ahead on (UVLTrack-L from 5000 visual rows, any model from
 * 16000), 2 wherever the pair kernels exist (>= 2048 rows), 3 one-sequence frames only.  "fuse_contrast" (default 1): 0 selects
 * the stand-alone contrast kernels, so that tests and tools can compare the launch forms (same results).  "fold_modal"
 * (default 1): 0 keeps the fusion layers' modal embedding in their LayerNorm-1 (1: in the previous fc2 epilogue where the
 * residual GEMMs run in place; logits equal within one f32 rounding).  "prefetch_w" (default 1): next-weight requests in the
 * small-tile GEMM launches -- 0 never, 1 below 2000 visual rows for models whose weights exceed the memory-side cache, 2 always.  "fork_text" (default 1): 0 runs the
 * text branch of multi-sequence frames on the caller's stream.  "fuse_ln" (default 0): 1 launches LayerNorm and the GEMM that
 * consumes it as one kernel behind a grid barrier in one-sequence frames (96 -> 72 launches, same bits, measured slower).  "fold_ln" (default 1): 0 keeps the
 * LayerNorm launches and split-K slabs of rounds 1-5 in one-sequence frames (1: the LayerNorm-free schedule, see uvl_linear_fin below).  "rider_sk" (default 2): 1 runs
 * the text rider of a many-sequence fc2 launch in one K slice, in place (the round-4 form).  "rider_first" (default 1): 0 puts the text rider's tiles of a
 * one-sequence pair GEMM launch behind the visual tiles.  "text_nt" (default 15): which text-branch GEMMs of frames of up to four sequences load their
 * weights non-temporal -- bit 0 QKV, 1 attention output, 2 intermediate, 3 output (0: none; two UVLTrack-B sequences lose 3.5 % with it, one is level).
 * "bf16_store" (default 3): which bf16 activations of frames of >= 2048 rows are stored write-through -- bit 0 the fc1 output, 1 the q / k rows of QKV, 2 LayerNorm's
 * rows (0: plain stores, the round-4 form: 8 UVLTrack-L sequences lose 2.2 %).  "head_fin" (default 1): 0 keeps the towers' last 3x3 layer and the head tail as two
 * launches where the one-launch form applies (16 x 16 search features, HEAD_DIM 256: see uvl_head_end).  "rider_pf" (default 0): n > 0 lets the text riders of BERT layers < n
 * of a one-sequence frame request the next rider's weight into the L2 of the XCD that will read it (measured level: profiles/NOTES.md). */
int uvl_debug_set(uvl_model_t* m, const char* key, int value);

/* Overrides of the launch heuristics, for tools and tests (not part of the product path).  There is NO process-global tuning
 * state: a model handle owns one uvl_tuning (uvl_tune_set writes it; every launch of that handle's frames reads it), and the
 * per-kernel entry points below take an optional `const uvl_tuning*` (NULL = heuristics).  Every field: -1 = heuristic / default.
 *   gemm_cfg   index into the plain-GEMM tile configuration table (gemm.hip::launch_plain_cfg)
 *   gemm_gm    tile order: 0 = every XCD owns whole N panels, g > 0 = groups of g M-tiles
 *   gemm_prod  0 = the wide bf16-output GEMMs of batched frames keep the all-waves-load form (default: producer waves)
 *   gemm_big   0 = never pick the 256x256 tile (default: where its tiles fill whole rounds of CUs)
 *   gemm_kxcd  0 = split-K GEMMs of one or two sequences keep the tile map (default: K-slice map)
 *   attn_cfg   index into the attention configuration table (attention.hip::launch_attention)
 *   sk_k1 / sk_k4  split-K factor of the frame's residual GEMMs with K = D / K = 4 D
 *   gemm_pipe  0 = never pick the phase-pipelined 256-wide GEMM (default: batched frames, see gemm.hip::pick_plain_cfg)
 *   ring1      ring depth (3..5 stages of 16 KB) of the 64x64 tile that one-sequence frames use (default 4)
 *   res_store  cache policy of the in-place f32 residual stores (x += ...) of the GEMM epilogue: 0 plain, 1 non-temporal, 2 write-through (default)
 *   slab_store the same for the split-K f32 slabs of one-sequence frames (default 2)
 *   text_cfg   tile configuration (as gemm_cfg) of the text-branch GEMMs of multi-sequence frames, which run on the second stream
 *   attn_wgs   persistent workgroups of the hand-scheduled attention kernel (attention.hip::launch_attn_p64; default 512 = two per CU)
 *   gemm_dr    the direct-to-register GEMM (cfg 36, gemm_dr.hip; needs the packed weight image): 0 = never, 1 = wherever it applies,
 *              default = frames of >= 2048 rows, bf16-type epilogues (bias / GELU / QKV scatter); the f32 read-modify-write epilogue stays with cfg 30 / 31
 *   res_pre    0 = the in-place f32 residual GEMMs of many-sequence frames load their residual rows in the epilogue (the round-4 form); default:
 *              up to K = 2048 -- and at every K where the launch is a single round of tiles -- the rows are requested inside the K loop, one 16-byte load per lane
 *              and phase over eight K tiles (gemm.hip::gemm_pipe128_body, PRE); 2 = at every K
 *   fin_w      tile of the finishing residual GEMMs of LayerNorm-free frames (gemm_fin.hip): 0 = 64 x 64 on two wave groups, 1 = 64 x 32 on four; default: 64 x 32 while
 *              its tiles are at most one workgroup per CU; 2 = additionally keep the split-K slab form for every layer of the conv towers (A/B)
 *   lnf_w      tile of the LayerNorm-folded QKV / fc1 GEMMs of LayerNorm-free frames (gemm.hip LNF): 0 = 64 x 64 / four stages, 1 = 64 x 128 / two stages, 2 = 64 x 128 / three stages;
 *              default: 64 x 128 where the grid would exceed two 64 x 64 tiles per CU (one UVLTrack-L sequence)
 * uvl_tuning_init fills a struct with -1.  Keys of uvl_tune_set are the field names. */
typedef struct uvl_tuning {
    int32_t gemm_cfg, gemm_gm, gemm_prod, gemm_big, gemm_kxcd, attn_cfg, sk_k1, sk_k4, gemm_pipe, ring1, text_cfg, res_store, slab_store, attn_wgs, gemm_dr, res_pre, fin_w, lnf_w;
} uvl_tuning;
void uvl_tuning_init(uvl_tuning* t);
int uvl_tune_set(uvl_model_t* m, const char* key, int value);
/* y[sk] (f32 slabs [splitk][M,N]) = partial sums over K-range sk (bias in slab 0): the split-K form of the frame's residual GEMMs */
int uvl_linear_splitk(const void* d_x, const void* d_w, const float* d_bias, float* d_slabs,
                      int M, int N, int K, int splitk, const uvl_tuning* tune, void* stream);

/* ---- per-kernel entry points (used by the parity tests; same kernels the forward uses) ---------- */

/* y = act(x W^T + b): nn.Linear (block.py:42,44; backbones/utils.py:58,61; bert_backbone.py:289-291,338,366,379).
 * d_x [M,K] bf16 row-major, d_w [N,K] bf16 (nn.Linear layout), d_bias [N] f32 or NULL.
 * act: 0 none, 1 erf-GELU, 2 ReLU.  out_f32 != 0: d_y is f32 [M,N] and `accumulate` adds into it
 * (the residual add of block.py:30-31); otherwise d_y is bf16 [M,N].  K % 64 == 0, N % 64 == 0. */
int uvl_linear(const void* d_x, const void* d_w, const float* d_bias, void* d_y,
               int M, int N, int K, int act, int out_f32, int accumulate, const uvl_tuning* tune, void* stream);

/* The same with the weight given once more in the fragment-native layout of gemm_dr_kernel (cfg 36: 128 x 256 tiles, two workgroups per
 * CU, the weight fragments loaded straight into registers): d_w_packed = output of uvl_pack_weight (N % 16 == 0, K % 64 == 0; same size as
 * d_w); NULL = that kernel is not available for this call.  A model handle packs its own weights at uvl_finalize_weights when max_batch
 * allows frames of >= 2048 rows. */
int uvl_pack_weight(const void* d_w, void* d_w_packed, int N, int K, void* stream);
int uvl_linear_pk(const void* d_x, const void* d_w, const void* d_w_packed, const float* d_bias, void* d_y,
                  int M, int N, int K, int act, int out_f32, int accumulate, const uvl_tuning* tune, void* stream);

/* ModalityUnifiedFeatureExtractor.contractive_learning (extractor.py:79-93) of one layer, as the frame runs it: logits[b, slot, s] = exp(logit_scale) *
 * normalize(x[b, 1 + nz + s]) . normalize(token), token = [x[b, 0], text token, mean of the two logits][flag[b]]; the text token is row `text_row` of the sample
 * ('cls') or the masked mean of rows text_row .. + text_len (mean_mode; d_text_mask [batch, text_len]); skip_text: visual logits only.  d_x: f32 [batch,
 * rows_per_sample, dim]; d_logits: f32 [batch, n_cont, nx].  form 0 = contrast_kernel (LayerNorm-kernel schedule), 1 = the job that rides in the QKV launches of a
 * LayerNorm-free frame, launched alone ('cls' token only).  Exported for the parity tests. */
int uvl_contrast_logits(const float* d_x, int batch, int rows_per_sample, int dim, int nz, int nx, int text_row, int text_len, const uint8_t* d_text_mask,
                        int mean_mode, int skip_text, const int64_t* d_flag, const float* d_logit_scale, float* d_logits, int slot, int n_cont, int form, void* stream);

/* The end of ModalityAdaptiveBoxHead.forward (modality_adaptive_box_head.py:71-94) from the towers' third 3x3 layer onwards: the last conv3x3 + BatchNorm(eval) + ReLU
 * of the four towers (heads/utils.py:126-131; C/4 -> C/8 channels), their 1x1 convs, the sigmoids, the size-map select by flag and convert2bbox (:108-119) incl. the
 * argmax -- as the frame runs it.  d_g3: bf16 [batch * feat^2][4 * cin] (tower-major channels), d_w_packed / d_bias_folded: uvl_fold_conv_bn's outputs for cout = cin / 2,
 * d_w1 [7][cin / 2] / d_b1 [7]: the 1x1 weights in the order cls | offset x, y | bbox w, h | bbox_grounding w, h; d_coord: box_head.coodinate [2][feat^2].
 * form 0: the layer as a conv launch + the tail kernel (any geometry; d_scratch: bf16 [batch * feat^2][4 * cin / 2]); form 1: ONE launch, one workgroup per sample
 * (16 x 16 features, cin = 64: UVL_EINVAL otherwise; d_scratch: 4 * 32 * 9 * 64 bf16, receives the weights in fragment order).  Outputs may be null except bbox_map. */
int uvl_head_end(const void* d_g3, int batch, int feat, int cin, const void* d_w_packed, const float* d_bias_folded, const float* d_w1, const float* d_b1,
                 const float* d_cont_score, int cont_channels, const int64_t* d_flag, const float* d_coord, int offset_sigmoid, int joint_cls, int form, void* d_scratch,
                 float* d_cls_score, float* d_cls_score_test, float* d_bbox_map, float* d_pred_boxes, int64_t* d_argmax, void* stream);

/* Fused multi-head self-attention core of Attention.forward (block.py:50-58) and BertSelfAttention
 * (bert_backbone.py:311-324): softmax(q k^T / sqrt(64) + key_add) v.
 * d_q, d_k: [B,H,Npad,64] bf16; d_vt: [B,H,64,Npad] bf16 (V transposed); d_key_add: [B,Npad] f32 additive
 * per-key term (-1e10 reproduces masked_fill for |score| < 512, -10000 is BERT's mask); d_o: [B*N, H*64] bf16.
 * q_prescaled != 0: d_q already carries the factor UVL_ATTN_QSCALE = log2(e)/sqrt(64) (the kernels work in the log2 domain;
 * the frame's QKV projection applies it before rounding q to bf16, uvl_qkv_project with q_scale = UVL_ATTN_QSCALE does the
 * same); 0: d_q is the plain projection and the kernel applies the factor itself. */
#define UVL_ATTN_QSCALE 0.18033688011112042f
int uvl_attention(const void* d_q, const void* d_k, const void* d_vt, const float* d_key_add, void* d_o,
                  int B, int H, int N, int Npad, int q_prescaled, const uvl_tuning* tune, void* stream);

/* QKV projection with the scatter epilogue the attention kernel consumes (block.py:49-50):
 * d_x [B*N, D] bf16, d_w [3D, D] bf16, d_bias [3D] f32 -> q,k [B,H,Npad,64], vt [B,H,64,Npad]; q is multiplied by
 * q_scale before it is rounded (1.0f = the plain projection). */
int uvl_qkv_project(const void* d_x, const void* d_w, const float* d_bias, void* d_q, void* d_k, void* d_vt,
                    int B, int N, int Npad, int D, float q_scale, const uvl_tuning* tune, void* stream);
int uvl_qkv_project_pk(const void* d_x, const void* d_w, const void* d_w_packed, const float* d_bias, void* d_q, void* d_k, void* d_vt,
                       int B, int N, int Npad, int D, float q_scale, const uvl_tuning* tune, void* stream);     /* d_w_packed: see uvl_linear_pk */

/* One layer of the box head's four 3x3 conv towers, conv(3x3, pad 1) + BatchNorm2d(eval) + ReLU (heads/utils.py:126-131;
 * towers of modality_adaptive_box_head.py:28-50), as the frame runs it: BatchNorm folded into bf16 weights, the four towers as
 * groups of one implicit GEMM over NHWC tokens.
 * uvl_fold_conv_bn: d_w [cout,cin,3,3] f32, conv bias and BatchNorm weight/bias/running_mean/running_var [cout] (eps 1e-5)
 *   -> d_w_packed bf16 [cout][9][cin] (tap-major, as the kernel reads it) and d_bias_folded f32 [cout], for ONE tower.
 * uvl_conv_tower_layer: d_x bf16 [batch*feat*feat, x_ld] NHWC tokens; tower g reads channels [x_group_offset[g], +cin);
 *   d_w_packed [4][cout][9*cin] bf16, d_bias_folded [4*cout] -> d_y bf16 [batch*feat*feat, 4*cout] (tower-major channels).
 *   d_slabs: optional f32 scratch of 8 * batch*feat*feat * 4*cout elements; when given, small batches split K as the frame does. */
int uvl_fold_conv_bn(const float* d_w, const float* d_b, const float* d_bn_w, const float* d_bn_b, const float* d_bn_mean,
                     const float* d_bn_var, void* d_w_packed, float* d_bias_folded, int cout, int cin, void* stream);
int uvl_conv_tower_layer(const void* d_x, int batch, int feat, int x_ld, const int32_t x_group_offset[4], int cin, int cout,
                         const void* d_w_packed, const float* d_bias_folded, void* d_y, float* d_slabs, const uvl_tuning* tune, void* stream);

/* nn.LayerNorm / BertLayerNorm over the last dim (block.py:30-31 eps 1e-6; bert_backbone.py:231-244 eps 1e-12).
 * d_x [M,D] f32 -> d_y_bf16 [M,D] bf16 (may be NULL) and d_y_f32 [M,D] f32 (may be NULL, may alias d_x). */
int uvl_layernorm(const float* d_x, const float* d_gamma, const float* d_beta, float eps,
                  void* d_y_bf16, float* d_y_f32, int M, int D, void* stream);

/* ---- the LayerNorm-free forms one- / two-sequence frames run (uvl_debug_set "fold_ln", default 1) ---------------------------------------------
 * A `norm -> Linear` pair (block.py:30-31 -> attn.qkv / mlp.fc1; BertLayerNorm -> query/key/value / intermediate.dense, bert_backbone.py:335-339,366,376-380)
 * runs as ONE GEMM on the UN-normalised rows rounded to bf16:  y = rstd (a~ W'^T - mean colsum(W')) + b'  with W' = bf16(W gamma), b' = b + W beta, mean / rstd
 * of the f32 rows a, a~ = bf16(a).  The rows' statistics travel as per-32-column partials  stats[((j / 2) * M + row) * 4 + (j % 2) * 2 + {0,1}] (j < K/32: planes of column-block PAIRS, all M rows each) = (sum, sum of squares) of
 * the f32 values of columns [32 j, 32 j + 32) (before their rounding to bf16), written by whoever writes the bf16 row.
 * uvl_fold_ln_linear: d_w [N,K] f32, d_bias [N] (or NULL), gamma / beta [K] -> d_w_folded bf16 [N,K], d_bias_folded [N], d_colsum [N] (row sums of the ROUNDED W').
 * uvl_linear_fin:  x (+)= a W^T + b finished inside the launch (block.py:29-32 residual adds; no split-K slabs): one eight-wave workgroup per 64 x 64 tile, its two
 *   wave groups on the two K halves.  d_a [M,K] bf16, d_w [N,K] bf16, d_x [M,N] f32 in/out (accumulate != 0 adds), d_xn [M,N] bf16 = bf16(x) and d_stats
 *   [N/64, M, 2, 2] = its partials (both optional).  d_res_stats != NULL: post-LayerNorm residual (bert_backbone.py:335-339) -- d_x holds PRE-norm rows u, d_res_stats
 *   the partials of u, and the residual added is LayerNorm(u; gamma, beta, res_eps); d_res_copy (optional) receives those normalised rows.  N % 64 == 0, K % 128 == 0.
 * uvl_linear_lnf / uvl_qkv_project_lnf: the consumer GEMMs (epilogues of uvl_linear / uvl_qkv_project), d_a = bf16 rows + d_stats.  K % 128 == 0, K <= 1024. */
int uvl_fold_ln_linear(const float* d_w, const float* d_bias, const float* d_gamma, const float* d_beta, void* d_w_folded, float* d_bias_folded, float* d_colsum,
                       int N, int K, void* stream);
int uvl_linear_fin(const void* d_a, const void* d_w, const float* d_bias, float* d_x, void* d_xn, float* d_stats, int M, int N, int K, int accumulate,
                   const float* d_res_stats, const float* d_res_gamma, const float* d_res_beta, float res_eps, float* d_res_copy, const uvl_tuning* tune, void* stream);
int uvl_linear_lnf(const void* d_a, const float* d_stats, const void* d_w_folded, const float* d_bias_folded, const float* d_colsum, float eps, void* d_y,
                   int M, int N, int K, int act, const uvl_tuning* tune, void* stream);
int uvl_qkv_project_lnf(const void* d_a, const float* d_stats, const void* d_w_folded, const float* d_bias_folded, const float* d_colsum, float eps,
                        void* d_q, void* d_k, void* d_vt, int B, int N, int Npad, int D, float q_scale, const uvl_tuning* tune, void* stream);

/* f32 -> bf16 (round to nearest even) helper for tests. */
int uvl_f32_to_bf16(const float* d_in, void* d_out, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UVLTRACK_HIP_H */
