// Host-visible launchers of the gfx950 kernels (internal to the library; the public ABI is include/uvltrack_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "../../include/uvltrack_hip.h"

namespace uvl {

// name of the kernel instantiation the last launcher picked (for per-kernel profiles)
extern thread_local const char* g_last_kernel;
// Overrides of the launch heuristics (include/uvltrack_hip.h: uvl_tuning), owned by a model handle or passed by the caller of a
// per-kernel entry point -- never process-global.  tune_get(t, &uvl_tuning::field, dflt): the field, or dflt when t is null / -1.
inline int tune_get(const uvl_tuning* t, int32_t uvl_tuning::*field, int dflt) { return (t && t->*field >= 0) ? t->*field : dflt; }

struct GemmParams {
    const bf16_t* A = nullptr; int lda = 0;      // [M,K] bf16 (plain) or NHWC activations (conv)
    const bf16_t* W = nullptr; int ldw = 0;      // [groups*N, K] bf16, K-contiguous
    const bf16_t* Wp = nullptr;                  // optional: the same weight in the fragment-native layout of gemm_dr_kernel (launch_pack_w_dr)
    const float* bias = nullptr;                 // [groups*N] or null
    int M = 0, N = 0, K = 0;                     // per group
    int epi = 0;                                 // EPI_BF16 / EPI_F32 / EPI_QKV
    void* C = nullptr; int ldc = 0;              // output (bf16 or f32)
    int act = 0;                                 // 0 none, 1 erf-GELU, 2 ReLU       (EPI_BF16)
    int accumulate = 0;                          // C += ...                         (EPI_F32)
    int rpb = 1 << 30, obs = 0, oro = 0;         // row m -> (b = m / rpb, t = m % rpb) -> output row b*obs + oro + t
    const float* addtab = nullptr;               // [rpb, N] f32 added by t            (EPI_F32; pos-embed)
    int addtab_split = 0;                        // > 0: addtab is [2, N], row (t >= addtab_split) (the modal embeddings of the fusion layers, mae_vit.py:196)
    bf16_t *q = nullptr, *k = nullptr, *vt = nullptr; int H = 0, Npad = 0, D = 0;   // EPI_QKV
    float q_scale = 1.0f;                        // EPI_QKV: factor applied to the q columns before rounding (log2(e)/8 in the frame)
    int groups = 1;
    int splitk = 1; size_t part_stride = 0;      // split-K: slab sk of C (f32, + sk*part_stride elements) holds partial sums of K-range sk
    int conv_F = 0, cin_g = 0;                   // conv mode: feature-map side, input channels per group
    int a_goff[4] = {0, 0, 0, 0};                // conv mode: channel offset of each group's input inside a row
    int w_stream = 0;                            // 1: W is read once per frame and should not displace resident weights in the
                                                 // Infinity Cache (text branch): non-temporal weight-tile loads where instantiated
    const uvl_tuning* tune = nullptr;            // host side only: overrides of the launch heuristics (null = heuristics)
    int c_store = 0;                             // EPI_F32 stores: 0 plain, 1 non-temporal, 2 write-through (sc1); A/B knob uvl_tuning.res_store
    const void* pf = nullptr; uint32_t pf_bytes = 0;   // optional: bytes the launch requests for the NEXT launch (its weight), see common.h::prefetch_issue
    const void* pf2 = nullptr; uint32_t pf2_tiles = 0, pf2_lp = 0;   // a rider's params: the NEXT rider's weight, requested XCD-matched (common.h::prefetch_issue_xcd)
                                                 // (the 64 x 64-tile kernels of small frames; the large-tile kernels ignore it)
    int group_m = 0;                             // set by the launcher: 0 = each XCD owns whole N panels (weights stream once; small M),
                                                 // g > 0 = grouped order, g M-tiles x all N-tiles per group, contiguous runs per XCD (large M)
    // set by the launcher (gemm_derive): divisors of the tile decode and of the row map with their reciprocals -- no integer division in a kernel prologue
    FastDiv fd_mt, fd_gsz, fd_gm, fd_gml;        // row tiles; tiles of a full group; row tiles of a full group / of the last group
    FastDiv fd_rpb;                              // rpb
    int ks_log2 = 0, ks_ntp = 0;                 // K-slice map (group_m < 0): log2(8 / splitk), column tiles per XCD share
    int kspan = 0;                               // K / splitk
    int rider_first = 1;                         // pair launches of one-sequence frames (gemm_glds_pair_kernel): this problem, as the RIDER, takes the first block indices
    // ---- LayerNorm-free frames (round 6; gemm_fin.hip and the LNF forms of gemm.hip) -------------------------------------------------------------------
    // Row statistics travel as PARTIALS: st[fold.h::st_off(j, row, rows) + {0, 1}] = [j / 2][row][j % 2][.] = (sum, sum of squares) of the f32 values of columns [32 j, 32 j + 32) of
    // a row (taken before the row is rounded to bf16), j < np = D / 32, `rows` = the row count of the array (plane j holds all rows: the lanes of a consumer wave,
    // one row each, then read CONSECUTIVE addresses -- row-major partials made every statistics load touch 64 lines and cost the UVLTrack-L QKV / fc1 launches +2.7 us).  Whoever writes a bf16 row that a LayerNorm-folded GEMM will read leaves them; that GEMM adds them up (mean / rstd per row).
    // Producer side (launch_gemm_fin: x (+)= A W^T + b finished in the launch, no slabs):
    bf16_t* xn = nullptr; int xn_bs = 0, xn_ro = 0;   // bf16 copy of the finished rows: GEMM row (b, t) -> xn row b * xn_bs + xn_ro + t, [., N]
    float* st_out = nullptr; int st_rows = 0;    // partials of those rows, row index like xn, st_rows = rows per plane (= M of the GEMM that will read them)
    // post-LN residual (bert_backbone.py:335-339,376-380): the residual operand is LayerNorm(row of C) -- C holds the PRE-norm rows u; res_st = partials of
    // u indexed like xn, res_g / res_b / res_eps the LayerNorm; res_copy (optional, [M, N] compact) receives the normalised rows (the text snapshot)
    const float* res_st = nullptr; const float *res_g = nullptr, *res_b = nullptr; float res_eps = 0.f; float* res_copy = nullptr;
    // Consumer side (launch_gemm_lnf: y = act(LayerNorm(a) W^T + b) on A = bf16(a) UN-normalised): W = bf16(W gamma), bias = b + W beta, colsum[n] = sum_k W[n, k]
    // (of the rounded weight), st_in = partials of A's rows (compact index m): y = rstd (acc - mean colsum) + bias
    const float* st_in = nullptr; const float* colsum = nullptr; float ln_eps = 0.f;      // (st_in: planes of M rows)
};
// The contrastive logits of one layer (extractor.py:85-93) as extra workgroups of a GEMM launch (LayerNorm-free frames: the job has no LayerNorm launch to ride on):
// one wave per search row; the rows are complete in x when the hosting launch starts.
struct CtJob {
    const float* x = nullptr; int xbs = 0, D = 0, B = 0;       // residual stream [B, xbs, D]
    int nz = 0, nv = 0, nx = 0, skip_text = 0;
    const float* txt = nullptr; int txt_bs = 0;                // text token rows: sample b at txt + b * txt_bs * D (null: row nv of x)
    const float *txt_g = nullptr, *txt_b = nullptr; float txt_eps = 0.f;   // non-null: the text row is PRE-norm, the job normalises it (bert_backbone.py:376-380)
    const float* txt_st = nullptr; int txt_st_bs = 0, txt_st_rows = 0;   // ... with the row's partial statistics: row b * txt_st_bs of planes of txt_st_rows rows
    const float *sub_vis = nullptr, *sub_txt = nullptr;        // the next fusion layer's modal embedding, already added to x by the fc2 epilogue: taken off again
    const int64_t* flag = nullptr; const float* logit_scale = nullptr; float* logits = nullptr; int slot = 0, ncont = 0;
};
// Text rows entering the first fusion layer of a LayerNorm-free frame (extractor.py:62-63, mae_vit.py:196): x_text = LayerNorm(u) (the last BERT layer's
// output LayerNorm, or rows kept from an earlier frame) -> snapshot, + modal_embed[1] -> residual stream, bf16 copy + partials for the first joint QKV GEMM.
struct TextJoinParams {
    float* x = nullptr; int xbs = 0, xro = 0;                  // residual stream rows (b * xbs + xro + t): in (pre-norm u) and out
    const float* alt = nullptr;                                // non-null: already-normalised rows [B*T, D] to use instead (text reuse); no LayerNorm
    const float *gamma = nullptr, *beta = nullptr; float eps = 1e-12f;
    float* snap = nullptr;                                     // optional [B*T, D]: the normalised rows
    const float* add = nullptr;                                // optional [D] vector added after the snapshot (modal_embed[1])
    bf16_t* xn = nullptr; int xn_bs = 0, xn_ro = 0; float* st = nullptr; int st_rows = 0;
    int B = 0, T = 0, D = 0;
};
hipError_t launch_text_join(const TextJoinParams& p, hipStream_t s);
// W' = bf16(W gamma), b' = b + W beta, colsum = row sums of W' (one nn.Linear whose input is a LayerNorm: block.py:30-31 -> :42 / mlp.fc1; bert_backbone.py)
hipError_t launch_fold_ln_linear(const float* W, const float* bias, const float* gamma, const float* beta, bf16_t* Wf, float* bf, float* colsum, int N, int K, hipStream_t s);
// x (+)= A W^T + b finished inside the launch: one eight-wave workgroup per 64 x 64 tile, the two K halves on its two wave groups (gemm_fin.hip); b = optional rider
hipError_t launch_gemm_fin(const GemmParams& a, const GemmParams* b, hipStream_t s);
bool gemm_fin_ok(const GemmParams& p);
// one layer of the 3x3 conv towers (relu, bf16 out) in the same workgroup shape, no split-K slabs: conv_fin_form != 0 says whether the layer fits
int conv_fin_form(const GemmParams& p);
hipError_t launch_conv_fin(const GemmParams& p, hipStream_t s);
// LayerNorm-folded consumer GEMM on the 64 x 64 tiles (EPI_BF16 / EPI_QKV); b = optional rider (plain or folded: b->st_in), ct = optional logits job
hipError_t launch_gemm_lnf(const GemmParams& a, const GemmParams* b, const CtJob* ct, hipStream_t s);
// fills the derived fields for a tile grid of BM x BN tiles (group_m already chosen)
static inline void gemm_derive(GemmParams& p, int BM, int BN) {
    const int MT = (p.M + BM - 1) / BM, NT = p.N / BN;
    p.fd_mt = fastdiv_of((uint32_t)MT);
    const int g = p.group_m > 0 ? p.group_m : 1;
    p.fd_gsz = fastdiv_of((uint32_t)(g * (NT > 0 ? NT : 1)));
    p.fd_gm = fastdiv_of((uint32_t)g);
    p.fd_gml = fastdiv_of((uint32_t)((MT % g) ? (MT % g) : g));
    p.fd_rpb = fastdiv_of((uint32_t)p.rpb);
    const int sk = p.splitk > 1 ? p.splitk : 1;
    p.kspan = p.K / sk;
    if (p.group_m < 0) {
        const int nparts = 8 / sk;
        int l2 = 0;
        while ((1 << l2) < nparts) ++l2;
        p.ks_log2 = l2;
        p.ks_ntp = NT / nparts;
    }
}
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);
// two independent plain GEMMs; one launch when both resolve to the batch-1 instantiation, else two launches (same results)
hipError_t launch_gemm_pair(const GemmParams& a, const GemmParams& b, hipStream_t s);

struct AttnParams {
    const bf16_t *q = nullptr, *k = nullptr, *vt = nullptr;   // [B,H,Npad,64], [B,H,Npad,64], [B,H,64,Npad]
    const float* key_add = nullptr; int key_add_stride = 0;   // [B, stride] additive per-key score term
    bf16_t* o = nullptr;                                      // [B*N, H*64]
    int B = 0, H = 0, N = 0, Npad = 0;
    int xcd_map = 0;                                          // set by the launcher: query blocks of a head share an XCD (see attn_decode_block)
    FastDiv fd_nqb, fd_h;                                     // set by the launcher: query blocks per head (of the launched configuration), heads
    const uvl_tuning* tune = nullptr;                         // host side only: overrides of the launch heuristics (null = heuristics)
    int q_prescaled = 0;                                      // 1: q already carries the factor log2(e)/8 (GemmParams.q_scale of the QKV GEMM)
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);
hipError_t launch_attention_pair(const AttnParams& a, const AttnParams& b, hipStream_t s);   // b (few keys) rides on a's configuration

struct LnParams {
    const float* x = nullptr;                    // input rows, f32
    int M = 0, D = 0;                            // compact row count
    int rpb = 1 << 30, xbs = 0, xro = 0;         // compact row m -> x row (m/rpb)*xbs + xro + m%rpb
    FastDiv fd_rpb;                              // set by the launcher: fastdiv_of(rpb)
    const float* part = nullptr; int nsplit = 0, part_rows = 0; size_t part_stride = 0;   // pending split-K slabs [nsplit][B*part_rows, D]
                                                 // added to the row first and written back: x += sum_s part[s]
    const float* pre_add0 = nullptr;             // optional vector added to rows with t <  split (then written back to x)
    const float* pre_add1 = nullptr;             //                          rows with t >= split
    int split = 0;
    const float* x_alt = nullptr; int x_alt_rows = 0;   // optional: rows with t >= split are READ from x_alt[b * x_alt_rows + t - split]
                                                 // (text rows kept from an earlier frame); the fold is still written to x
    const float *gamma = nullptr, *beta = nullptr; float eps = 1e-6f;
    bf16_t* y_bf16 = nullptr;                    // [M,D] compact, optional
    int y_wt = 0;                                // 1: y_bf16 is stored write-through (set by the fused LayerNorm + GEMM launch only)
    float* y_f32 = nullptr; int y_remap = 0;     // optional f32 output; y_remap: same row map as x (in-place LN) else compact
    float* y_copy = nullptr;                     // optional second f32 output, compact [M,D] (text snapshot)
    float* x_snap = nullptr;                     // optional copy of the row AFTER the slab fold and BEFORE pre_add (= the previous
                                                 // layer's output), same row map as x
    // Optional second job of the launch: the contrastive logits of the PREVIOUS layer (extractor.py:85-93) from the snapshot the
    // previous LayerNorm left in ct_x -- the wave that normalises search row s of sample b also writes logits[b, slot, s].
    // No extra launch, no cross-stream event ('cls' text token only; 'mean' keeps the stand-alone contrast kernel).
    const float* ct_x = nullptr;                 // snapshot [B, xbs, D]; row 0 = vis token, rows 1+ct_nz.. = search, row ct_nv = text row 0
    const float* ct_txt = nullptr;               // [B, ct_T, D] text rows of that layer (pre-fusion layers) or null = snapshot row ct_nv
    int ct_nz = 0, ct_nv = 0, ct_nx = 0, ct_T = 0, ct_skip_text = 0, ct_slot = 0, ct_ncont = 0;
    const int64_t* ct_flag = nullptr; const float* ct_logit_scale = nullptr; float* ct_logits = nullptr;
    // ct_self: the job's rows are this launch's own input rows (ct_x = x, which the launch must not modify: no slabs, no pre_add on the rows it
    // reads) -- many-sequence frames, whose fc2 epilogue has already added the next layer's modal embedding (GemmParams.addtab_split): the
    // job takes it off again (ct_sub_vis from the search rows and the vis token, ct_sub_txt from the text token), no snapshot is written.
    int ct_self = 0;
    const float *ct_sub_vis = nullptr, *ct_sub_txt = nullptr;
};
hipError_t launch_layernorm(const LnParams& p, hipStream_t s);
hipError_t launch_layernorm_pair(const LnParams& a, const LnParams& b, hipStream_t s);
// one-sequence frames: LayerNorm(_pair) + the GEMM(_pair) that consumes it in one launch behind a grid barrier (gemm.hip); falls back to the
// two launches where the fused form does not apply.  bar: 4 KB of zero-initialised device memory owned by the model, gen: 1, 2, 3, ..., *base: arrivals per group so far (updated)
hipError_t launch_pack_w_dr(const bf16_t* W, bf16_t* Wp, int N, int K, hipStream_t s);   // [N, K] -> fragment-native image (same size)
hipError_t launch_gemm_dr(const GemmParams& p, int epi, hipStream_t s);      // gemm_dr.hip: 128 x 256 on four waves, two workgroups per CU, W straight into registers (cfg 36)
bool gemm_dr_pairable(const GemmParams& a, const GemmParams& b);
hipError_t launch_gemm_dr_pair(const GemmParams& a, const GemmParams& b, hipStream_t s);   // both problems on cfg 36, one launch (b = the text rider)
hipError_t launch_ln_gemm_pair(const LnParams& la, const LnParams* lb, const GemmParams& a, const GemmParams* b, unsigned* bar, unsigned gen, unsigned* base, bool* fused, hipStream_t s);   // two independent problems, one launch

// set-up + BERT embedding + im2row of a single-stream frame in one launch (rowops.hip::prologue_kernel)
struct PrologueParams {
    const uint8_t* text_mask = nullptr; const int64_t* flag = nullptr; const float* cls_token = nullptr;
    float* x = nullptr; float* key_add = nullptr; float* bert_add = nullptr;
    int nz = 0, nv = 0, nj = 0, npad = 0, T = 0, D = 0, B = 0;
    const int64_t* ids = nullptr; const float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    bf16_t* tn = nullptr; int vocab = 0;
    const float *z = nullptr, *ximg = nullptr; bf16_t* patches = nullptr; int Hz = 0, Hx = 0;
    int skip_text = 0, setup_what = 3;           // as launch_setup; ids == nullptr: no embedding workgroups
    int n_setup = 0, n_embed = 0;                // filled by the launcher
    // LayerNorm-free frames (fold.h): the [cls] row also as bf16 + partials (row b * cls_xn_bs of cls_xn / cls_st: the first QKV GEMM reads it un-normalised);
    // embed_raw: the BERT embedding row is left PRE-norm (f32 row, bf16 copy in tn, partials in embed_st [B*T, D/32, 2]) -- its LayerNorm (emb_g / emb_b, eps 1e-12)
    // is folded into the first QKV GEMM and applied to the residual by the first attention.output GEMM
    bf16_t* cls_xn = nullptr; int cls_xn_bs = 0; float* cls_st = nullptr; int cls_st_rows = 0;
    int embed_raw = 0; float* embed_st = nullptr;            // (embed_st: planes of B * T rows)
};
hipError_t launch_prologue(const PrologueParams& p, hipStream_t s);

// images -> bf16 patch rows [(b, z tokens..., x tokens...), 768] in (c,kh,kw) order (mae_vit.py:94-100)
hipError_t launch_im2row(const float* z, const float* x, bf16_t* out, int B, int Hz, int Hx, hipStream_t s);

// BertEmbeddings.forward + BertModel.embedding mask (bert_backbone.py:260-274,740-750)
hipError_t launch_bert_embed(const int64_t* ids, const float* word, const float* pos, const float* type0,
                             const float* gamma, const float* beta, float* x, int xbs, int xro, bf16_t* y_bf16,
                             int B, int T, int D, int vocab, hipStream_t s);

// cat_mask (extractor.py:43-50) + cls-token rows: per-key additive terms and the [cls] row of the residual stream.
hipError_t launch_setup(const uint8_t* text_mask, const int64_t* flag, const float* cls_token, float* x,
                        float* key_add, float* bert_add, int B, int nz, int nv, int nj, int npad, int T, int D,
                        int skip_text, int what /* 1 = key_add + cls rows, 2 = BERT mask */, hipStream_t s);

// ModalityUnifiedFeatureExtractor.contractive_learning (extractor.py:85-93) for one layer.
struct ContrastParams {
    const float* x = nullptr; int nj = 0, nz = 0, nx = 0, nv = 0, D = 0, T = 0, B = 0;
    const uint8_t* text_mask = nullptr; const int64_t* flag = nullptr; const float* logit_scale = nullptr;
    int mean_mode = 0, skip_text = 0;
    float* logits = nullptr; int slot = 0, n_cont = 0;
    const float* part = nullptr; int nsplit = 0, part_rows = 0; size_t part_stride = 0;   // pending split-K slabs
    const float* txt_snap = nullptr;             // [B,T,D] text rows of this layer (pre-fusion layers) or null = read x
};
hipError_t launch_contrast(const ContrastParams& p, hipStream_t s);
hipError_t launch_ct_job(const CtJob& j, hipStream_t s);

// Head prologue: copy residual rows to the output dict, emit the bf16 NHWC conv input and cont_score (head:140-148).
struct HeadPrepParams {
    const float* x = nullptr; int nj = 0, nv = 0, nz = 0, nx = 0, T = 0, D = 0, B = 0;
    const float* prompt = nullptr; const float* logit_scale = nullptr;
    const uint8_t* text_mask = nullptr; const int64_t* flag = nullptr;
    int softmax_one = 1, mean_mode = 0, cls_tokenize = 0, skip_text = 0;
    int cont_only = 0;                           // 1: only cont_score, search rows read from o_search (compact [B,S,D]); x is not touched
    int train_cont = 0;                          // 1: cont_score layout of the head's no-prompt branch, [B,S,2] (head:134-138)
    bf16_t* g0 = nullptr; int g0_ld = 0;
    // optional second job: the backbone's contrastive logits of the LAST layer (extractor.py:85-93) for the same search rows
    float* ct_logits = nullptr; const float* ct_logit_scale = nullptr; int ct_slot = 0, ct_ncont = 0;
    float *o_search = nullptr, *o_template = nullptr, *o_text = nullptr, *o_vis = nullptr, *o_txt = nullptr, *o_cont = nullptr;
};
hipError_t launch_head_prep(const HeadPrepParams& p, hipStream_t s);

// Head tail: the four 1x1 convs + sigmoid + size select + convert2bbox + argmax (head:74-94,108-119).
struct HeadTailParams {
    const bf16_t* g4 = nullptr; int ld = 0, c8 = 0;          // [B*S, 4*c8] bf16 (towers cls, offset, bbox, bbox_grounding)
    const float* w1 = nullptr; const float* b1 = nullptr;    // packed 1x1 weights [7, c8], bias [7]
    const float* cont = nullptr; int cont_ch = 3;            // cont_score [B,S,cont_ch]
    const int64_t* flag = nullptr; const float* coord = nullptr;   // coodinate buffer [2,S]
    int B = 0, S = 0, F = 0, offset_sigmoid = 1, joint_cls = 0;
    float *o_cls = nullptr, *o_cls_test = nullptr, *o_bbox_map = nullptr, *o_pred = nullptr; int64_t* o_argmax = nullptr;
};
hipError_t launch_head_tail(const HeadTailParams& p, hipStream_t s);

// The last 3x3 tower layer + the tail as one launch, one workgroup per sample (head_fin.hip): 16 x 16 search features, 4 x 64 -> 4 x 32 channels
struct HeadFinParams {
    HeadTailParams t;                                        // g4 / ld unused; c8 = 32
    const bf16_t* g3 = nullptr; int g3_ld = 0;               // [B*S, 4*64] bf16: the output of tower layer 2
    const bf16_t* wf = nullptr; const float* bias3 = nullptr;   // layer 3's weights in fragment order (launch_head_fin_pack), bias [4*32]
};
bool head_fin_ok(const HeadFinParams& p);
hipError_t launch_head_fin(const HeadFinParams& p, hipStream_t s);
hipError_t launch_head_fin_pack(const bf16_t* w, bf16_t* wf, hipStream_t s);

// The prompter (DistributionBasedCrossAttention, heads/utils.py:23-99): token sums before the MLP, and the flag switch.
struct PrompterParams {
    const float *tem = nullptr, *ctx = nullptr, *vis = nullptr, *txt = nullptr;     // [B,nz,D], [B,S,D], [B,1,D], [B,1,D] f32
    const uint8_t *tem_mask = nullptr, *ctx_mask = nullptr;                         // [B,nz], [B,S]  1 = target cell
    const int64_t* flag = nullptr;
    const float *query_embed = nullptr, *logit_scale = nullptr;
    int B = 0, nz = 0, S = 0, D = 0;
    int ctx_roll = 0;                                                               // sample b reads the context tokens of sample (b + ctx_roll) % B
                                                                                    // (head:132, the batch-rolled context of the no-prompt branch)
    float *src = nullptr, *src0 = nullptr;                                          // [B,3,D] f32: tokens + src_, and src_
    bf16_t* src_bf16 = nullptr;                                                     // [3B, D] MLP operand
};
hipError_t launch_prompter_tokens(const PrompterParams& p, hipStream_t s);
hipError_t launch_anno2mask(const float* boxes_xywh, int B, int size, uint8_t* mask, hipStream_t s);   // tracker:183-194
hipError_t launch_prompter_select(const float* src, const float* src0, const int64_t* flag, float* out, int B, int n_per_sample, hipStream_t s);

// Tracker decode (tracker:116-125,167-173; box_ops.clip_box): argmax(cls * hann * softmax(cont)[0]) -> box in image coordinates.
struct DecodeParams {
    const float *cls = nullptr, *cont = nullptr, *bbox_map = nullptr, *window = nullptr;   // [B,S], [B,S,cont_ch] or null, [B,S,4], [S]
    const float *state = nullptr, *resize_factor = nullptr, *image_hw = nullptr;            // [B,4] xywh, [B], [B,2] (H, W)
    int B = 0, S = 0, cont_ch = 3; float search_size = 256.f, margin = 10.f;
    float *new_state = nullptr, *score = nullptr, *box_net = nullptr; int64_t* index = nullptr;
};
hipError_t launch_decode(const DecodeParams& p, hipStream_t s);

// sample_target + Preprocessor_wo_mask on the uint8 frame (processing_utils.py:159-243, tracker_utils.py:20-29); see preprocess.hip
struct PreprocParams {
    const uint8_t* img = nullptr; int H = 0, W = 0, stride = 0;          // HWC uint8 frame (or a window of it), `stride` bytes per row
    int ox = 0, oy = 0;                                                   // frame coordinates of img's pixel (0,0) when img is a window
    int crop_sz = 0, x1 = 0, y1 = 0, x1_pad = 0, x2_pad = 0, y1_pad = 0, y2_pad = 0;   // crop geometry (host, integer)
    int out = 0;                                                          // output side
    uint8_t* patch = nullptr;                                             // [out,out,3] uint8, optional
    float* norm = nullptr;                                                // [3,out,out] f32 normalised, optional
    uint8_t* att = nullptr;                                               // [out,out] 0/1 attention mask of the padded area, optional
};
hipError_t launch_preprocess(const PreprocParams& p, hipStream_t s);
struct GroundingParams {                                                  // grounding_resize, processing_utils.py:60-141
    const uint8_t* img = nullptr; int H = 0, W = 0, stride = 0;
    int new_w = 0, new_h = 0, x1_pad = 0, y1_pad = 0, out = 0;
    uint8_t* patch = nullptr; float* norm = nullptr; uint8_t* att = nullptr;
};
hipError_t launch_grounding_resize(const GroundingParams& p, hipStream_t s);
hipError_t launch_normalize_u8(const uint8_t* src, float* dst, int n_pix, hipStream_t s);

// out = relu(sum of split-K slabs) as bf16 (conv towers)
hipError_t launch_slab_relu(const float* slabs, int nsplit, size_t stride, bf16_t* out, size_t n, hipStream_t s);

// weight packing
hipError_t launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
// conv [Co,Ci,3,3] f32 + BN(eval) -> bf16 [Co][tap][Ci] and folded bias f32 [Co]
hipError_t launch_fold_conv_bn(const float* w, const float* b, const float* bn_w, const float* bn_b, const float* bn_mean,
                               const float* bn_var, bf16_t* w_out, float* b_out, int Co, int Ci, hipStream_t s);
hipError_t launch_copy_f32(const float* in, float* out, size_t n, hipStream_t s);

}  // namespace uvl
