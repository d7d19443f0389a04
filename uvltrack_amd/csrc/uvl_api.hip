// C ABI + host-side frame scheduler of the gfx950 UVLTrack forward pass (see include/uvltrack_hip.h).
//
// The frame is a fixed sequence of launches -- one stream for a single sequence (the text-branch kernels ride in the visual
// launches of their kind), two streams for several (the BERT text branch runs beside the visual ViT branch until the first
// fusion layer: extractor.py:57-65) -- with no host sync and no allocation; uvl_graph_capture() records it for replay.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/uvltrack_hip.h"
#include "kernels.h"

using namespace uvl;

namespace uvl { thread_local const char* g_last_kernel = "?"; }

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return fail(UVL_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ---- Dispatch: every frame-size rule of the scheduler, in one table (DESIGN.md, section "Dispatch", mirrors it row by row) --------------------------------
// Rows = B x tokens of the frame.  The kernel-level choices behind these rules live with the kernels: gemm.hip::pick_plain_cfg (tile configuration by M / N / K),
// attention.hip::pick_attn_cfg (attention form by workgroup count), gemm_fin.hip::fin_w32 / conv_fin_form (tile width of the finishing GEMMs).
struct DispatchRules {
    int cus;                      // 256   CUs of an MI355X: "one workgroup per CU" in the rules below
    long fold_max_tiles;          // 256   LayerNorm-free schedule (one sequence): residual GEMMs of at most this many 64 x 64 tiles, i.e. one eight-wave workgroup per CU
                                  //       (profiles/r06_lnfold_b1.md: 1416 -> 1552 frames/s on one UVLTrack-B sequence; tools/probes/combine2_probe.hip)
    int splitk_cap;               // 4     split-K slices of a LayerNorm-kernel-schedule residual GEMM (UVL_SKMAX slabs in the workspace)
    int splitk_min_ktiles;        // 6     ... a slice is at least this many K tiles (profiles/r01_gemm_sweep.md)
    int splitk_long_half;         // 24    ... and from a grid that fits one workgroup per CU K is halved once more only into halves this long (profiles/r05_summary.md)
    long dr_min_rows;             // 2048  frames of >= this many rows: packed weight images at finalize, gemm_dr_kernel for QKV / fc1 (profiles/r04_gemm_dr.md), riders possible
    long rider_min_rows;          // 5000  the text branch rides in the visual launches of a many-sequence frame from here on ... (profiles/r04_text_branch.md)
    int rider_wide_dim;           // 1024  ... for models at least this wide (UVLTrack-L: +0.7..1.9 %), or
    long rider_any_rows;          // 16000 ... for any model from here on (UVLTrack-B x 32: +1.3..1.8 %; x 12 / 16 / 24 lose 1-1.6 % and keep the second stream)
    long text_dr_min_rows;        // 6000  text branch on the second stream: its GEMMs on gemm_dr_kernel's tiles from here on (8 UVLTrack-B sequences lose 3.4 % below it)
    long prefetch_max_rows;       // 2000  next-weight requests in the small-tile GEMM launches below this many visual rows ... (tools/probes/wprefetch_probe.py)
    double prefetch_min_weights;  // 2e8   ... for models whose ViT weights (bytes) cannot stay in the 256 MB memory-side cache (UVLTrack-L x 1 +2.6..3.4 %; -B: 0)
    long conv_sk_max_blocks;      // 768   split-K of a conv tower layer: slices while tiles x slices stay within three workgroups per CU (profiles/r01_summary.md)
    int head_fin_max_batch;       // 1<<30 last tower layer + head tail as one launch, one workgroup per sample, at 16 x 16 search features / HEAD_DIM 256 (head_fin.hip; profiles/r06_head_fin.md)
};
static const DispatchRules kDispatch = {256, 256, 4, 6, 24, 2048, 5000, 1024, 16000, 6000, 2000, 2.0e8, 768, 1 << 30};

struct RawTensor {
    float* d = nullptr;
    std::vector<int64_t> shape;
    size_t numel = 0;
};

struct VitBlockW {
    const float *ln1g, *ln1b, *ln2g, *ln2b, *bqkv, *bproj, *bfc1, *bfc2;
    bf16_t *wqkv, *wproj, *wfc1, *wfc2;
    bf16_t *pqkv, *pproj, *pfc1, *pfc2;          // the same weights in the fragment-native layout of gemm_dr_kernel (null: frames of this handle never reach 2048 rows)
    // LayerNorm-free frames (fold.h): attn.qkv with norm1 and mlp.fc1 with norm2 folded in -- bf16(W gamma), b + W beta, row sums of the folded weight
    bf16_t *fqkv, *ffc1; float *fbqkv, *fbfc1, *csqkv, *csfc1;
};
struct BertLayerW {
    const float *bao, *bi, *bo, *ln1g, *ln1b, *ln2g, *ln2b;
    float* bqkv;
    bf16_t *wqkv, *wao, *wi, *wo;
    bf16_t *pqkv, *pao, *pi, *po;                // fragment-native images for gemm_dr_kernel (null: see VitBlockW)
    // LayerNorm-free frames: query/key/value with the LayerNorm in front of them folded in (layer 0: the embedding LayerNorm; layer l: output.LayerNorm of l - 1),
    // intermediate.dense with this layer's attention.output.LayerNorm
    bf16_t *fqkv, *fi; float *fbqkv, *fbi, *csqkv, *csi;
};
struct ConvLayerW {
    bf16_t* w;      // [4][Cout][9*Cin]
    float* b;       // [4*Cout]
    int cin, cout;
};

struct ProfEntry {
    std::string name, kernel;
    double ms = 0, flops = 0, bytes = 0;
    double wbytes = 0;              // SURVEY 8(d)'s count: weights touched once, activations cache-resident (attention: q, k, v in + o out)
    int launches = 0;
};
struct Profiler {
    struct Rec { hipEvent_t a, b; const char* name; const char* kernel; double flops, bytes, wbytes; };
    std::vector<Rec> recs;
};

struct uvl_model {
    uvl_config cfg;
    int D, H, depth, nf, nz, nx, nv, nj, npad, T, F, S, C, ffn;
    std::map<std::string, RawTensor> raw;
    std::vector<void*> owned;           // packed allocations
    bool finalized = false;
    // packed
    bf16_t* w_patch = nullptr; const float* b_patch = nullptr; float* pos_tab = nullptr;
    const float *cls_token = nullptr, *modal = nullptr, *logit_scale_bb = nullptr, *logit_scale_head = nullptr, *coord = nullptr;
    const float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    std::vector<VitBlockW> vit;
    std::vector<BertLayerW> bert;
    ConvLayerW conv[4];
    float *w1 = nullptr, *b1 = nullptr;
    // prompter (optional: only forward_prompt needs it)
    bool has_prompter = false;
    const float *pr_logit_scale = nullptr, *pr_query = nullptr, *pr_b1 = nullptr, *pr_b2 = nullptr;
    bf16_t *pr_w1 = nullptr, *pr_w2 = nullptr;
    // streams / events
    hipStream_t aux = nullptr;                   // text-branch stream (frames of several sequences)
    int pair_text = 1;                           // uvl_debug_set("pair_text", v): 0 = the text branch always on its own stream, 1 = riders for one sequence and for the
                                                 // many-sequence frames text_rides() lists, 2 = riders wherever the pair forms exist, 3 = one sequence only (A/B)
    int text_dr_res = 0;                         // uvl_debug_set("text_dr_res", 1): the text branch's residual GEMMs (12 tiles) on gemm_dr_kernel too (A/B)
    int fork_text = 1;                           // uvl_debug_set("fork_text", 0): multi-sequence frames run the text branch on the caller's stream (A/B)
    int fuse_contrast = 1;                       // uvl_debug_set("fuse_contrast", 0): stand-alone contrast kernels
    int prefetch_w = 1;                          // uvl_debug_set("prefetch_w", v): 0 = no next-weight requests in the GEMM launches, 1 = in frames below 2000 visual rows, 2 = always (A/B)
    int bf16_store = 3;                          // uvl_debug_set("bf16_store", mask): which bf16 activations are stored write-through (sc1), see run_gemm.  The request is set on every visual
                                                 // epi 0 / 2 GEMM; only gemm_dr_kernel (frames of >= 2048 rows) honours it -- gemm_epilogue_lds's bf16 / QKV stores ignore c_store
    int text_nt = 15;                            // uvl_debug_set("text_nt", mask): which text-branch GEMMs load their weights non-temporal (1 QKV, 2 attention output, 4 intermediate, 8 output).
                                                 // UNPAIRED text launches only (frames of 2-4 sequences): the riders of pair launches (gemm_glds_pair_kernel, gemm_lnf_pair_kernel,
                                                 // gemm_fin_pair_kernel) always load their weight tiles non-temporal
    int rider_first = 1;                         // uvl_debug_set("rider_first", 0): the text rider's tiles of a one-sequence pair GEMM launch behind the visual tiles (the round-4 order; A/B)
    int rider_sk = 2;                            // uvl_debug_set("rider_sk", 1): the text rider's output GEMM in one K slice, in place (the round-4 form; A/B)
    int fold_modal = 1;                          // uvl_debug_set("fold_modal", 0): the fusion layers' modal embedding always added by their LayerNorm-1 (A/B)
    bf16_t* head_wf = nullptr;                   // tower layer 3's weights in head_fin_kernel's fragment order (finalize; null when the geometry does not fit)
    int rider_pf = 0;                            // uvl_debug_set("rider_pf", n): the riders of BERT layers < n request the NEXT rider's weight XCD-matched (common.h::prefetch_issue_xcd).  Off: the
                                                 // pair launches gain 1.1 us each, but the requests allocate the BERT weights in the memory-side cache and the ViT weights of the later blocks leave it (profiles/NOTES.md, round 6)
    int head_fin = 1;                            // uvl_debug_set("head_fin", 0): tower layer 3 and head_tail as two launches (rounds 1-5; A/B)
    int fold_ln = 1;                             // uvl_debug_set("fold_ln", 0): one-sequence frames keep their LayerNorm launches and split-K slabs (the round-1..5 schedule; A/B)
    int fuse_ln = 0;                             // uvl_debug_set("fuse_ln", 1): one-sequence frames launch LayerNorm + its consumer GEMM as ONE kernel behind a
                                                 // grid barrier (96 -> 72 launches; measured 3-4 % SLOWER than the two launches, so off: profiles/r03_summary.md)
    unsigned* gbar = nullptr;                    // 4 KB of counters for the fused launches' grid barrier (monotonic: never reset)
    unsigned gbar_gen = 0, gbar_base = 0;        // generation of the last fused launch; arrivals per counter group so far
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<hipEvent_t> ev_bert, ev_cont;
    // graph
    hipStream_t cap_stream = nullptr;
    hipGraph_t graph3[3] = {nullptr, nullptr, nullptr};          // T, V1, V2 (or only [1] = whole frame without text)
    hipGraphExec_t graph_exec3[3] = {nullptr, nullptr, nullptr};
    bool graph_has_text = false;
    // profile of the last profiled call
    std::vector<ProfEntry> prof;
    int debug_stop_layer = -1;          // >= 0: leave the layer loop after this layer (tests localise errors with it)
    uvl_tuning tune;                    // overrides of the launch heuristics for THIS handle (uvl_tune_set); all -1 = heuristics
    uvl_tuning tune_text;               // = tune with gemm_cfg replaced by text_cfg: what the text-branch GEMMs of multi-sequence frames see
    uvl_model() { uvl_tuning_init(&tune); }
};

extern "C" void uvl_tuning_init(uvl_tuning* t) {
    if (!t) return;
    int32_t* f = reinterpret_cast<int32_t*>(t);
    for (size_t i = 0; i < sizeof(uvl_tuning) / sizeof(int32_t); ++i) f[i] = -1;
}
extern "C" const char* uvl_last_error(void) { return g_err; }
extern "C" int uvl_version(void) { return 4; }
#ifndef UVL_BUILD_TOOLCHAIN
#define UVL_BUILD_TOOLCHAIN "unknown"
#endif
extern "C" const char* uvl_build_toolchain(void) { return UVL_BUILD_TOOLCHAIN; }

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" uvl_model_t* uvl_create(const uvl_config* c) {
    if (!c) { fail(UVL_EINVAL, "null config"); return nullptr; }
    if (c->dim <= 0 || c->dim % 64 != 0 || c->dim > 1024) { fail(UVL_EINVAL, "dim %d must be a multiple of 64 and <= 1024", c->dim); return nullptr; }
    if (c->heads <= 0 || c->dim != c->heads * 64) { fail(UVL_EINVAL, "dim/heads must be 64 (got %d/%d)", c->dim, c->heads); return nullptr; }
    if (c->depth <= 0 || c->depth > UVL_MAX_LAYERS || c->n_fusion_start < 0 || c->n_fusion_start > c->depth) { fail(UVL_EINVAL, "bad depth/fusion"); return nullptr; }
    if (c->template_size % 16 || c->search_size % 16 || c->template_size <= 0 || c->search_size <= 0) { fail(UVL_EINVAL, "image sizes must be multiples of 16"); return nullptr; }
    if (c->text_len <= 0 || c->text_len > 64 || c->text_len > c->max_pos) { fail(UVL_EINVAL, "text_len must be in [1,64]"); return nullptr; }
    if (c->head_dim <= 0 || c->head_dim % 256 != 0) { fail(UVL_EINVAL, "head_dim must be a multiple of 256"); return nullptr; }
    if (c->n_cont < 0 || c->n_cont > UVL_MAX_LAYERS) { fail(UVL_EINVAL, "bad n_cont"); return nullptr; }
    uvl_model* m = new uvl_model();
    m->cfg = *c;
    m->D = c->dim; m->H = c->heads; m->depth = c->depth; m->nf = c->n_fusion_start;
    m->nz = (c->template_size / 16) * (c->template_size / 16);
    m->F = c->search_size / 16;
    m->nx = m->F * m->F; m->S = m->nx;
    m->nv = 1 + m->nz + m->nx; m->T = c->text_len; m->nj = m->nv + m->T;
    m->npad = (int)align_up(m->nj, 64);
    m->C = c->head_dim; m->ffn = 4 * c->dim;
    if (hipStreamCreateWithFlags(&m->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess) {
        fail(UVL_EHIP, "stream/event creation failed (is a HIP device present?)");
        delete m;
        return nullptr;
    }
    if (hipMalloc(&m->gbar, 4096) == hipSuccess) hipMemset(m->gbar, 0, 4096); else m->gbar = nullptr;   // no barrier memory: the fused launches fall back
    m->ev_bert.resize(c->depth);
    m->ev_cont.resize(c->depth);
    for (auto& e : m->ev_bert) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& e : m->ev_cont) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    return m;
}

extern "C" void uvl_destroy(uvl_model_t* m) {
    if (!m) return;
    hipDeviceSynchronize();
    for (int k = 0; k < 3; ++k) {
        if (m->graph_exec3[k]) hipGraphExecDestroy(m->graph_exec3[k]);
        if (m->graph3[k]) hipGraphDestroy(m->graph3[k]);
    }
    if (m->cap_stream) hipStreamDestroy(m->cap_stream);
    for (auto& kv : m->raw) if (kv.second.d) hipFree(kv.second.d);
    for (void* p : m->owned) hipFree(p);
    for (auto& e : m->ev_bert) hipEventDestroy(e);
    for (auto& e : m->ev_cont) hipEventDestroy(e);
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    if (m->ev_join) hipEventDestroy(m->ev_join);
    if (m->aux) hipStreamDestroy(m->aux);
    if (m->gbar) hipFree(m->gbar);
    delete m;
}

static bool name_is_prompter_used(const std::string& n) {
    static const char* used[] = {"box_head.prompter.logit_scale", "box_head.prompter.query_embed.weight", "box_head.prompter.mlp.fc1.weight",
                                 "box_head.prompter.mlp.fc1.bias", "box_head.prompter.mlp.fc2.weight", "box_head.prompter.mlp.fc2.bias"};
    for (const char* u : used) if (n == u) return true;
    return false;
}
static bool name_is_ignored(const std::string& n) {
    // prompter.{q,kv,proj,norm} exist in the checkpoint but are never used by the reference either (heads/utils.py:31-40)
    return (n.find("box_head.prompter.") == 0 && !name_is_prompter_used(n)) || n.find("backbone.bert.pooler.") == 0 ||
           n.find("backbone.vit.norm.") == 0 || n.find("num_batches_tracked") != std::string::npos;
}

static bool name_is_known(const uvl_model* m, const std::string& n) {
    static const char* fixed[] = {"backbone.logit_scale", "backbone.vit.cls_token", "backbone.vit.pos_embed_z", "backbone.vit.pos_embed_x",
                                  "backbone.vit.modal_embed", "backbone.vit.patch_embed.proj.weight", "backbone.vit.patch_embed.proj.bias",
                                  "backbone.bert.embeddings.word_embeddings.weight", "backbone.bert.embeddings.position_embeddings.weight",
                                  "backbone.bert.embeddings.token_type_embeddings.weight", "backbone.bert.embeddings.LayerNorm.weight",
                                  "backbone.bert.embeddings.LayerNorm.bias", "box_head.logit_scale", "box_head.coodinate"};
    for (const char* f : fixed) if (n == f) return true;
    if (name_is_prompter_used(n)) return true;
    if (n.find("backbone.vit.blocks.") == 0 || n.find("backbone.bert.encoder.layer.") == 0 || n.find("box_head.conv_") == 0) return true;
    (void)m;
    return false;
}

extern "C" int uvl_load_tensor(uvl_model_t* m, const char* name, const float* d_data, int ndim, const int64_t* dims, void* stream) {
    if (!m || !name || !d_data || ndim < 0 || ndim > 8) return fail(UVL_EINVAL, "uvl_load_tensor: bad argument");
    const std::string n(name);
    if (name_is_ignored(n)) return 1;
    if (!name_is_known(m, n)) return fail(UVL_ENOTFOUND, "unknown tensor '%s'", name);
    // BERT layers beyond the ones that run are accepted and dropped (extractor.py:28 truncates the encoder)
    if (n.find("backbone.bert.encoder.layer.") == 0) {
        const int li = atoi(n.c_str() + strlen("backbone.bert.encoder.layer."));
        if (li >= m->nf) return 1;
    }
    RawTensor& t = m->raw[n];
    size_t numel = 1;
    std::vector<int64_t> shape;
    for (int i = 0; i < ndim; ++i) { numel *= (size_t)dims[i]; shape.push_back(dims[i]); }
    if (t.d && t.numel != numel) { hipFree(t.d); t.d = nullptr; }
    if (!t.d) HIPCHK(hipMalloc(&t.d, numel * sizeof(float) + 16));
    t.shape = shape; t.numel = numel;
    HIPCHK(hipMemcpyAsync(t.d, d_data, numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    m->finalized = false;
    return UVL_OK;
}

// ---- finalize helpers ---------------------------------------------------------------------------
struct Packer {
    uvl_model* m;
    hipStream_t s;
    int err = 0;
    const RawTensor* get(const std::string& n, size_t numel) {
        auto it = m->raw.find(n);
        if (it == m->raw.end()) { if (!err) err = fail(UVL_ESTATE, "missing tensor '%s'", n.c_str()); return nullptr; }
        if (it->second.numel != numel) { if (!err) err = fail(UVL_EINVAL, "tensor '%s' has %zu elements, expected %zu", n.c_str(), it->second.numel, numel); return nullptr; }
        return &it->second;
    }
    const float* f32(const std::string& n, size_t numel) { const RawTensor* t = get(n, numel); return t ? t->d : nullptr; }
    template <class T> T* alloc(size_t count) {
        void* p = nullptr;
        if (hipMalloc(&p, count * sizeof(T) + 256) != hipSuccess) { if (!err) err = fail(UVL_EHIP, "hipMalloc of %zu bytes failed", count * sizeof(T)); return nullptr; }
        m->owned.push_back(p);
        return (T*)p;
    }
    bf16_t* bf16(const std::string& n, size_t numel, bf16_t* dst = nullptr) {
        const RawTensor* t = get(n, numel);
        if (!t) return nullptr;
        if (!dst) dst = alloc<bf16_t>(numel);
        if (!dst) return nullptr;
        if (launch_f32_to_bf16(t->d, dst, numel, s) != hipSuccess && !err) err = fail(UVL_EHIP, "f32->bf16 launch failed");
        return dst;
    }
    void drop(const std::string& n) {   // free the f32 copy of a tensor that now lives packed
        auto it = m->raw.find(n);
        if (it != m->raw.end() && it->second.d) { hipFree(it->second.d); it->second.d = nullptr; m->raw.erase(it); }
    }
};

// Fragment-native images of the RESIDUAL GEMMs' weights (ViT proj / fc2, BERT attention.output / output): read only when cfg 36 is forced onto the f32
// read-modify-write epilogue (uvl_tune_set "gemm_dr" 1) or the text branch's residual GEMMs are sent there (uvl_debug_set "text_dr_res" 1) -- made on
// the first such request (or at finalize when the request is already standing), not for every handle.  Synchronous on `s`.
static int pack_residual_images(uvl_model* m, hipStream_t s) {
    if (!m->finalized || (long)m->cfg.max_batch * m->nj < kDispatch.dr_min_rows) return UVL_OK;
    const int D = m->D, Fn = m->ffn;
    int err = 0;
    auto pack = [&](const bf16_t* src, int N_, int K_) -> bf16_t* {
        void* p = nullptr;
        if (!src || err) return nullptr;
        if (hipMalloc(&p, (size_t)N_ * K_ * sizeof(bf16_t) + 256) != hipSuccess) { err = fail(UVL_EHIP, "hipMalloc of a packed weight image failed"); return nullptr; }
        m->owned.push_back(p);
        if (launch_pack_w_dr(src, (bf16_t*)p, N_, K_, s) != hipSuccess) err = fail(UVL_EHIP, "weight packing launch failed");
        return (bf16_t*)p;
    };
    for (auto& w : m->vit) {
        if (!w.pproj) w.pproj = pack(w.wproj, D, D);
        if (!w.pfc2) w.pfc2 = pack(w.wfc2, D, Fn);
    }
    for (auto& w : m->bert) {
        if (!w.pao) w.pao = pack(w.wao, D, D);
        if (!w.po) w.po = pack(w.wo, D, Fn);
    }
    if (!err && hipStreamSynchronize(s) != hipSuccess) err = fail(UVL_EHIP, "sync after weight packing failed");
    return err ? err : UVL_OK;
}

extern "C" int uvl_finalize_weights(uvl_model_t* m, void* stream) {
    if (!m) return fail(UVL_EINVAL, "null model");
    if (m->finalized) return UVL_OK;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipStreamSynchronize(s));
    for (void* p : m->owned) hipFree(p);
    m->owned.clear();
    m->vit.clear(); m->bert.clear();
    Packer P{m, s};
    const size_t D = m->D, Fn = m->ffn;
    const std::string v = "backbone.vit.";
    m->logit_scale_bb = P.f32("backbone.logit_scale", 1);
    m->cls_token = P.f32(v + "cls_token", D);
    m->modal = P.f32(v + "modal_embed", 2 * D);
    m->w_patch = P.bf16(v + "patch_embed.proj.weight", D * 768);
    m->b_patch = P.f32(v + "patch_embed.proj.bias", D);
    m->pos_tab = P.alloc<float>((size_t)(m->nz + m->nx) * D);
    {
        const float* pz = P.f32(v + "pos_embed_z", (size_t)m->nz * D);
        const float* px = P.f32(v + "pos_embed_x", (size_t)m->nx * D);
        if (pz && px && m->pos_tab) {
            launch_copy_f32(pz, m->pos_tab, (size_t)m->nz * D, s);
            launch_copy_f32(px, m->pos_tab + (size_t)m->nz * D, (size_t)m->nx * D, s);
        }
    }
    for (int i = 0; i < m->depth; ++i) {
        const std::string b = v + "blocks." + std::to_string(i) + ".";
        VitBlockW w{};
        w.ln1g = P.f32(b + "norm1.weight", D); w.ln1b = P.f32(b + "norm1.bias", D);
        w.ln2g = P.f32(b + "norm2.weight", D); w.ln2b = P.f32(b + "norm2.bias", D);
        w.wqkv = P.bf16(b + "attn.qkv.weight", 3 * D * D); w.bqkv = P.f32(b + "attn.qkv.bias", 3 * D);
        w.wproj = P.bf16(b + "attn.proj.weight", D * D); w.bproj = P.f32(b + "attn.proj.bias", D);
        w.wfc1 = P.bf16(b + "mlp.fc1.weight", Fn * D); w.bfc1 = P.f32(b + "mlp.fc1.bias", Fn);
        w.wfc2 = P.bf16(b + "mlp.fc2.weight", D * Fn); w.bfc2 = P.f32(b + "mlp.fc2.bias", D);
        {   // LayerNorm-folded images for the LayerNorm-free one- / two-sequence frame (from the f32 weights, before they are dropped)
            w.fqkv = P.alloc<bf16_t>(3 * D * D); w.fbqkv = P.alloc<float>(3 * D); w.csqkv = P.alloc<float>(3 * D);
            w.ffc1 = P.alloc<bf16_t>(Fn * D); w.fbfc1 = P.alloc<float>(Fn); w.csfc1 = P.alloc<float>(Fn);
            const float* wq = P.f32(b + "attn.qkv.weight", 3 * D * D);
            const float* w1 = P.f32(b + "mlp.fc1.weight", Fn * D);
            if (wq && w1 && w.fqkv && w.fbqkv && w.csqkv && w.ffc1 && w.fbfc1 && w.csfc1 && w.ln1g && w.ln1b && w.ln2g && w.ln2b && w.bqkv && w.bfc1) {
                if (launch_fold_ln_linear(wq, w.bqkv, w.ln1g, w.ln1b, w.fqkv, w.fbqkv, w.csqkv, 3 * (int)D, (int)D, s) != hipSuccess ||
                    launch_fold_ln_linear(w1, w.bfc1, w.ln2g, w.ln2b, w.ffc1, w.fbfc1, w.csfc1, (int)Fn, (int)D, s) != hipSuccess)
                    if (!P.err) P.err = fail(UVL_EHIP, "LayerNorm fold launch failed");
            }
        }
        if ((long)m->cfg.max_batch * m->nj >= kDispatch.dr_min_rows) {          // many-sequence frames: second image of the weights for the direct-to-register GEMM
            auto pack = [&](const bf16_t* src, int N_, int K_) -> bf16_t* {
                bf16_t* dst = src ? P.alloc<bf16_t>((size_t)N_ * K_) : nullptr;
                if (dst && launch_pack_w_dr(src, dst, N_, K_, s) != hipSuccess && !P.err) P.err = fail(UVL_EHIP, "weight packing launch failed");
                return dst;
            };
            // QKV / fc1 only: the f32 read-modify-write GEMMs (proj, fc2) stay on the tile-grid kernels by default, so their images (5 / 12 of the packed
            // bytes: ~250 MB for UVLTrack-L) are made when a caller first asks for cfg 36 there (pack_residual_images: uvl_tune_set "gemm_dr" 1)
            w.pqkv = pack(w.wqkv, 3 * (int)D, (int)D); w.pfc1 = pack(w.wfc1, (int)Fn, (int)D);
        }
        m->vit.push_back(w);
    }
    const std::string e = "backbone.bert.embeddings.";
    m->word = P.f32(e + "word_embeddings.weight", (size_t)m->cfg.vocab * D);
    m->pos = P.f32(e + "position_embeddings.weight", (size_t)m->cfg.max_pos * D);
    m->type0 = P.f32(e + "token_type_embeddings.weight", 2 * D);
    m->emb_g = P.f32(e + "LayerNorm.weight", D); m->emb_b = P.f32(e + "LayerNorm.bias", D);
    for (int i = 0; i < m->nf; ++i) {
        const std::string b = "backbone.bert.encoder.layer." + std::to_string(i) + ".";
        BertLayerW w{};
        w.wqkv = P.alloc<bf16_t>(3 * D * D);
        w.bqkv = P.alloc<float>(3 * D);
        const char* nm[3] = {"query", "key", "value"};
        for (int k = 0; k < 3; ++k) {
            if (w.wqkv) P.bf16(b + "attention.self." + nm[k] + ".weight", D * D, w.wqkv + (size_t)k * D * D);
            const float* bb = P.f32(b + "attention.self." + nm[k] + ".bias", D);
            if (bb && w.bqkv) launch_copy_f32(bb, w.bqkv + (size_t)k * D, D, s);
        }
        w.wao = P.bf16(b + "attention.output.dense.weight", D * D); w.bao = P.f32(b + "attention.output.dense.bias", D);
        w.ln1g = P.f32(b + "attention.output.LayerNorm.weight", D); w.ln1b = P.f32(b + "attention.output.LayerNorm.bias", D);
        w.wi = P.bf16(b + "intermediate.dense.weight", Fn * D); w.bi = P.f32(b + "intermediate.dense.bias", Fn);
        w.wo = P.bf16(b + "output.dense.weight", D * Fn); w.bo = P.f32(b + "output.dense.bias", D);
        w.ln2g = P.f32(b + "output.LayerNorm.weight", D); w.ln2b = P.f32(b + "output.LayerNorm.bias", D);
        {   // LayerNorm-folded images (see VitBlockW): query/key/value read the LayerNorm in FRONT of this layer, intermediate.dense this layer's attention.output.LayerNorm
            w.fqkv = P.alloc<bf16_t>(3 * D * D); w.fbqkv = P.alloc<float>(3 * D); w.csqkv = P.alloc<float>(3 * D);
            w.fi = P.alloc<bf16_t>(Fn * D); w.fbi = P.alloc<float>(Fn); w.csi = P.alloc<float>(Fn);
            const float* pg = i == 0 ? m->emb_g : m->bert[i - 1].ln2g;
            const float* pb = i == 0 ? m->emb_b : m->bert[i - 1].ln2b;
            bool okf = w.fqkv && w.fbqkv && w.csqkv && w.fi && w.fbi && w.csi && pg && pb && w.ln1g && w.ln1b && w.bi;
            for (int k = 0; k < 3 && okf; ++k) {
                const float* wk = P.f32(b + "attention.self." + nm[k] + ".weight", D * D);
                const float* bk = P.f32(b + "attention.self." + nm[k] + ".bias", D);
                if (!wk || !bk) { okf = false; break; }
                if (launch_fold_ln_linear(wk, bk, pg, pb, w.fqkv + (size_t)k * D * D, w.fbqkv + (size_t)k * D, w.csqkv + (size_t)k * D, (int)D, (int)D, s) != hipSuccess && !P.err)
                    P.err = fail(UVL_EHIP, "LayerNorm fold launch failed");
            }
            const float* wi32 = okf ? P.f32(b + "intermediate.dense.weight", Fn * D) : nullptr;
            if (wi32 && launch_fold_ln_linear(wi32, w.bi, w.ln1g, w.ln1b, w.fi, w.fbi, w.csi, (int)Fn, (int)D, s) != hipSuccess && !P.err) P.err = fail(UVL_EHIP, "LayerNorm fold launch failed");
        }
        if ((long)m->cfg.max_batch * m->nj >= kDispatch.dr_min_rows) {          // the text branch of many-sequence frames runs on gemm_dr_kernel too (see run_gemm)
            auto pack = [&](const bf16_t* src, int N_, int K_) -> bf16_t* {
                bf16_t* dst = src ? P.alloc<bf16_t>((size_t)N_ * K_) : nullptr;
                if (dst && launch_pack_w_dr(src, dst, N_, K_, s) != hipSuccess && !P.err) P.err = fail(UVL_EHIP, "weight packing launch failed");
                return dst;
            };
            w.pqkv = pack(w.wqkv, 3 * (int)D, (int)D); w.pi = pack(w.wi, (int)Fn, (int)D);      // (pao / po: pack_residual_images, on demand)
        }
        m->bert.push_back(w);
    }
    // head: fold BN into the conv towers, tower-major packing
    const char* towers[4] = {"conv_cls", "conv_offset", "conv_bbox", "conv_bbox_grounding"};
    const int chans[5] = {m->D, m->C, m->C / 2, m->C / 4, m->C / 8};
    for (int l = 0; l < 4; ++l) {
        const int ci = chans[l], co = chans[l + 1];
        ConvLayerW& cw = m->conv[l];
        cw.cin = ci; cw.cout = co;
        cw.w = P.alloc<bf16_t>((size_t)4 * co * 9 * ci);
        cw.b = P.alloc<float>((size_t)4 * co);
        for (int t = 0; t < 4; ++t) {
            const std::string p = std::string("box_head.") + towers[t] + "." + std::to_string(l) + ".";
            const float* w = P.f32(p + "0.weight", (size_t)co * ci * 9);
            const float* b = P.f32(p + "0.bias", co);
            const float* g = P.f32(p + "1.weight", co);
            const float* be = P.f32(p + "1.bias", co);
            const float* mu = P.f32(p + "1.running_mean", co);
            const float* var = P.f32(p + "1.running_var", co);
            if (w && b && g && be && mu && var && cw.w && cw.b)
                launch_fold_conv_bn(w, b, g, be, mu, var, cw.w + (size_t)t * co * 9 * ci, cw.b + (size_t)t * co, co, ci, s);
        }
    }
    if (m->F == 16 && m->C == 256) {             // the geometry head_fin_kernel is written for (head_fin.hip)
        m->head_wf = P.alloc<bf16_t>((size_t)4 * 32 * 9 * 64);
        if (m->head_wf && m->conv[3].w && launch_head_fin_pack(m->conv[3].w, m->head_wf, s) != hipSuccess) return fail(UVL_EHIP, "head weight packing failed");
    }
    const int c8 = m->C / 8;
    m->w1 = P.alloc<float>(7 * c8);
    m->b1 = P.alloc<float>(8);
    {
        const int rows[4] = {1, 2, 2, 2};
        int off = 0;
        for (int t = 0; t < 4; ++t) {
            const std::string p = std::string("box_head.") + towers[t] + ".4.";
            const float* w = P.f32(p + "weight", (size_t)rows[t] * c8);
            const float* b = P.f32(p + "bias", rows[t]);
            if (w && b && m->w1 && m->b1) {
                launch_copy_f32(w, m->w1 + (size_t)off * c8, (size_t)rows[t] * c8, s);
                launch_copy_f32(b, m->b1 + off, rows[t], s);
            }
            off += rows[t];
        }
    }
    m->has_prompter = m->raw.count("box_head.prompter.mlp.fc1.weight") && m->raw.count("box_head.prompter.mlp.fc2.weight") &&
                      m->raw.count("box_head.prompter.query_embed.weight") && m->raw.count("box_head.prompter.logit_scale") &&
                      m->raw.count("box_head.prompter.mlp.fc1.bias") && m->raw.count("box_head.prompter.mlp.fc2.bias");
    if (m->has_prompter) {
        m->pr_logit_scale = P.f32("box_head.prompter.logit_scale", 1);
        m->pr_query = P.f32("box_head.prompter.query_embed.weight", 3 * D);
        m->pr_w1 = P.bf16("box_head.prompter.mlp.fc1.weight", Fn * D); m->pr_b1 = P.f32("box_head.prompter.mlp.fc1.bias", Fn);
        m->pr_w2 = P.bf16("box_head.prompter.mlp.fc2.weight", D * Fn); m->pr_b2 = P.f32("box_head.prompter.mlp.fc2.bias", D);
    }
    m->logit_scale_head = P.f32("box_head.logit_scale", 1);
    m->coord = P.f32("box_head.coodinate", 2 * (size_t)m->S);
    if (P.err) return P.err;
    HIPCHK(hipStreamSynchronize(s));
    // the f32 copies of the GEMM / conv weights are no longer needed
    P.drop(v + "patch_embed.proj.weight");
    for (int i = 0; i < m->depth; ++i) {
        const std::string b = v + "blocks." + std::to_string(i) + ".";
        P.drop(b + "attn.qkv.weight"); P.drop(b + "attn.proj.weight"); P.drop(b + "mlp.fc1.weight"); P.drop(b + "mlp.fc2.weight");
    }
    for (int i = 0; i < m->nf; ++i) {
        const std::string b = "backbone.bert.encoder.layer." + std::to_string(i) + ".";
        P.drop(b + "attention.self.query.weight"); P.drop(b + "attention.self.key.weight"); P.drop(b + "attention.self.value.weight");
        P.drop(b + "attention.output.dense.weight"); P.drop(b + "intermediate.dense.weight"); P.drop(b + "output.dense.weight");
    }
    for (int l = 0; l < 4; ++l)
        for (int t = 0; t < 4; ++t) P.drop(std::string("box_head.") + towers[t] + "." + std::to_string(l) + ".0.weight");
    if (m->has_prompter) { P.drop("box_head.prompter.mlp.fc1.weight"); P.drop("box_head.prompter.mlp.fc2.weight"); }
    for (int k = 0; k < 3; ++k) {
        if (m->graph_exec3[k]) { hipGraphExecDestroy(m->graph_exec3[k]); m->graph_exec3[k] = nullptr; }
        if (m->graph3[k]) { hipGraphDestroy(m->graph3[k]); m->graph3[k] = nullptr; }
    }
    m->finalized = true;
    if (m->tune.gemm_dr == 1 || m->tune.gemm_cfg == 36 || m->tune.text_cfg == 36 || m->text_dr_res) return pack_residual_images(m, s);      // the request is already standing
    return UVL_OK;
}

// ---- workspace ------------------------------------------------------------------------------------
#define UVL_SKMAX 4
#define UVL_QSCALE UVL_ATTN_QSCALE            // log2(e) / sqrt(64): the attention kernels work in the log2 domain (attention.hip)
#define UVL_CONV_SKMAX 8

// Split-K factor for an `x += A W^T` GEMM that would otherwise leave most CUs idle (batch-1 shapes): each split
// writes an f32 slab, the consuming LayerNorm / contrast kernel adds the slabs on read (deterministic, no atomics).
static int choose_splitk(int M, int N, int K, const uvl_tuning* tune) {
    const int cap = kDispatch.splitk_cap;
    const int forced = tune_get(tune, K > N ? &uvl_tuning::sk_k4 : &uvl_tuning::sk_k1, -1);   // uvl_tune_set("sk_k1" / "sk_k4")
    if (forced > 0 && (K / 64) % forced == 0 && forced <= cap) return forced;
    const long tiles = (long)((M + 63) / 64) * (N / 64);
    const int nk = K / 64;
    // One more halving of K while the grid still fits one workgroup per CU -- or, from a grid that does, if the halves are still 24 K tiles long.  Rounds 1-4 split
    // while tiles * sk < 256 (fc2 of one UVLTrack-B sequence: 108 tiles x 4 slices of 12 K tiles).  Round 5, interleaved tools/ab_tune.py sk_k4 / sk_k1 on one box
    // (frames/s, this rule / the old one): fc2 of one UVLTrack-B sequence as 2 slices 1372-1386 / 1361-1372, UVLTrack-L x 1 448.5 / 444.6; proj of two UVLTrack-B
    // sequences (216 tiles, 12 K tiles) unsplit 2042-2049 / 2019-2027, proj of one UVLTrack-L sequence (224 tiles, 16 K tiles) unsplit 459.8 / 455.6; fc2 of two
    // UVLTrack-B sequences keeps its 2 slices (2018-2037 against 1958-1981 unsplit and 1941-1945 with 4).
    int sk = 1;
    while (sk * 2 <= cap && nk % (sk * 2) == 0 && nk / (sk * 2) >= kDispatch.splitk_min_ktiles &&
           (tiles * sk * 2 <= kDispatch.cus || (tiles * sk < kDispatch.cus && nk / (sk * 2) >= kDispatch.splitk_long_half))) sk *= 2;
    return sk;
}

struct Pending { const float* part = nullptr; int nsplit = 0, rows = 0; size_t stride = 0; };

struct Workspace {
    float* X; bf16_t *Xn, *Q, *K, *Vt, *O, *Hb, *P;
    bf16_t *Tn, *Tq, *Tk, *Tvt, *To, *Th;
    float *key_add, *bert_add, *cont, *bbox, *Part, *PartT, *TxtSnap, *ConvPart, *XSnap;
    bf16_t *G0, *G1, *G2, *G3, *G4;
    float *St, *StT0, *StT1;            // LayerNorm-free frames: partial row statistics (fold.h) of the visual / joint rows and, ping-pong, of the text rows
    size_t total;
};
static Workspace carve(const uvl_model* m, int B, char* base) {
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    const size_t D = m->D, nj = m->nj, H = m->H, npad = m->npad, T = m->T, S = m->S, C = m->C;
    w.X = (float*)take(B * nj * D * 4);
    w.Xn = (bf16_t*)take(B * nj * D * 2);
    w.Q = (bf16_t*)take(B * H * npad * 64 * 2);
    w.K = (bf16_t*)take(B * H * npad * 64 * 2);
    w.Vt = (bf16_t*)take(B * H * npad * 64 * 2);
    w.O = (bf16_t*)take(B * nj * D * 2);
    w.Hb = (bf16_t*)take(B * nj * 4 * D * 2);
    w.P = (bf16_t*)take(B * (size_t)(m->nz + m->nx) * 768 * 2);
    w.Tn = (bf16_t*)take(B * T * D * 2);
    w.Tq = (bf16_t*)take(B * H * 64 * 64 * 2);
    w.Tk = (bf16_t*)take(B * H * 64 * 64 * 2);
    w.Tvt = (bf16_t*)take(B * H * 64 * 64 * 2);
    w.To = (bf16_t*)take(B * T * D * 2);
    w.Th = (bf16_t*)take(B * T * 4 * D * 2);
    w.key_add = (float*)take(B * npad * 4);
    w.bert_add = (float*)take(B * 64 * 4);
    w.Part = (float*)take((size_t)UVL_SKMAX * B * nj * D * 4);
    w.PartT = (float*)take((size_t)UVL_SKMAX * B * T * D * 4);
    w.TxtSnap = (float*)take((size_t)(m->nf > 0 ? m->nf : 1) * B * T * D * 4);
    w.XSnap = (float*)take(B * nj * D * 4);
    w.ConvPart = (float*)take((size_t)UVL_CONV_SKMAX * B * S * 4 * C * 4);
    w.cont = (float*)take(B * S * 3 * 4);
    w.bbox = (float*)take(B * S * 4 * 4);
    w.G0 = (bf16_t*)take(B * S * 2 * D * 2);
    w.G1 = (bf16_t*)take(B * S * 4 * C * 2);
    w.G2 = (bf16_t*)take(B * S * 2 * C * 2);
    w.G3 = (bf16_t*)take(B * S * C * 2);
    w.G4 = (bf16_t*)take(B * S * (C / 2) * 2);
    w.St = (float*)take(B * nj * (D / 32) * 2 * 4);
    w.StT0 = (float*)take(B * T * (D / 32) * 2 * 4);
    w.StT1 = (float*)take(B * T * (D / 32) * 2 * 4);
    w.total = off;
    return w;
}

extern "C" size_t uvl_workspace_bytes(const uvl_model_t* m, int batch) {
    if (!m || batch <= 0) return 0;
    return carve(m, batch, nullptr).total;
}

// ---- the frame -----------------------------------------------------------------------------------
// The frame is walked in full on the host every time (so split-K / pending-slab state evolves identically), but only
// the launches of the enabled parts are issued: eager runs enable everything, graph capture records one part per graph.
enum { PART_TEXT = 1, PART_V1 = 2, PART_V2 = 4, PART_ALL = 7 };

struct Launcher {
    Profiler* prof;
    int parts = PART_ALL, cur = PART_V1;
    int err = 0;
    double wb_next = -1;            // the weights-once byte count of the NEXT run() (GEMM / conv sites set it); < 0: the same as `bytes`
    void run(hipStream_t s, const char* what, double flops, double bytes, hipError_t (*fn)(void*, hipStream_t), void* ctx) {
        const double wbytes = wb_next >= 0 ? wb_next : bytes;
        wb_next = -1;
        if (err || !(parts & cur)) return;
        Profiler::Rec r{};
        if (prof) {
            hipEventCreate(&r.a); hipEventCreate(&r.b);
            hipEventRecord(r.a, s);
        }
        g_last_kernel = what;
        hipError_t e = fn(ctx, s);
        if (e != hipSuccess) { err = fail(UVL_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e)); return; }
        if (prof) {
            hipEventRecord(r.b, s);
            r.name = what; r.kernel = g_last_kernel; r.flops = flops; r.bytes = bytes; r.wbytes = wbytes;
            prof->recs.push_back(r);
        }
    }
};
template <class P_, hipError_t (*F)(const P_&, hipStream_t)>
static hipError_t tramp(void* ctx, hipStream_t s) { return F(*(const P_*)ctx, s); }

#define RUN_GEMM(L, s, p, what) (L).wb_next = 2.0 * (double)(p).N * (p).K * ((p).groups > 0 ? (p).groups : 1), (L).run((s), (what), 2.0 * (p).M * (p).N * (p).K * ((p).groups > 0 ? (p).groups : 1), \
    2.0 * ((double)(p).M * (p).K + (double)(p).N * (p).K * ((p).groups > 0 ? (p).groups : 1)), tramp<GemmParams, launch_gemm>, &(p))

// One layer of the four conv towers (heads/utils.py:126-131 with BatchNorm folded, modality_adaptive_box_head.py:28-50): grouped
// implicit GEMM over NHWC tokens; few output tiles and a long K (9 * Cin) split K into f32 slabs that a small kernel folds (+ReLU).
static void run_conv_layer(Launcher& L, hipStream_t s, int layer, const bf16_t* x, int in_ld, const int goff[4], const bf16_t* wpk, const float* bias,
                           int B, int F, int cin, int cout, bf16_t* y, float* slabs, const uvl_tuning* tune, const void* pf = nullptr, size_t pf_bytes = 0) {
    static const char* const conv_site[4] = {"conv3x3.0", "conv3x3.1", "conv3x3.2", "conv3x3.3"};
    GemmParams p;
    p.A = x; p.lda = in_ld; p.W = wpk; p.ldw = 9 * cin; p.bias = bias;
    p.M = B * F * F; p.N = cout; p.K = 9 * cin; p.ldc = 4 * cout;
    p.groups = 4; p.conv_F = F; p.cin_g = cin; p.tune = tune; p.pf = pf; p.pf_bytes = (uint32_t)pf_bytes;
    for (int g = 0; g < 4; ++g) p.a_goff[g] = goff[g];
    const long tiles = (long)((p.M + 63) / 64) * (p.N / 64 > 0 ? p.N / 64 : 1) * 4;
    const int nk = p.K / 64;
    int sk = 1;
    if (p.N % 64 == 0 && slabs)
        for (int c = 2; c <= 8 && c <= UVL_CONV_SKMAX; ++c)
            if (nk % c == 0 && nk / c >= kDispatch.splitk_min_ktiles && tiles * c <= kDispatch.conv_sk_max_blocks) sk = c;
    const char* site = conv_site[layer & 3];
    {   // one-sequence frames: layers whose K fits run as ONE launch -- K quarters / halves on the wave groups of an eight-wave workgroup (gemm_fin.hip::conv_fin_body)
        GemmParams q = p;
        q.epi = 0; q.C = y; q.act = 2;
        if (conv_fin_form(q)) {
            L.wb_next = 2.0 * (double)q.N * q.K * 4;
            L.run(s, site, 2.0 * q.M * q.N * q.K * 4, 2.0 * ((double)q.M * q.K + (double)q.N * q.K * 4), tramp<GemmParams, launch_conv_fin>, &q);
            return;
        }
    }
    if (sk > 1) {
        p.epi = 1; p.C = slabs; p.splitk = sk; p.part_stride = (size_t)p.M * p.ldc; p.c_store = tune_get(tune, &uvl_tuning::slab_store, 2);
        RUN_GEMM(L, s, p, site);
        struct RCtx { const float* slabs; int sk; size_t stride; bf16_t* out; size_t n; } rc{slabs, sk, p.part_stride, y, p.part_stride};
        L.run(s, "conv_fold", 0, 0, [](void* c, hipStream_t st) { auto* x = (RCtx*)c; return launch_slab_relu(x->slabs, x->sk, x->stride, x->out, x->n, st); }, &rc);
    } else {
        p.epi = 0; p.C = y; p.act = 2;
        RUN_GEMM(L, s, p, site);
    }
}

// The prompter on tokens that already sit in compact f32 buffers (heads/utils.py:82-99).  Scratch: src / src_ at the start of
// X, the bf16 MLP operand in Xn, the MLP hidden in Hb -- all idle once the head input has been gathered.
static int run_prompter(uvl_model* m, const Workspace& w, int B, const float* tem, const float* ctx, const float* vis, const float* txt,
                        const int64_t* flag, const uint8_t* tem_mask, const uint8_t* ctx_mask, int ctx_roll, float* prompt_out, hipStream_t s) {
    const int D = m->D, Fn = m->ffn;
    float* src = w.X;
    float* src0 = w.X + (size_t)B * 3 * D;
    PrompterParams p;
    p.tem = tem; p.ctx = ctx; p.vis = vis; p.txt = txt;
    p.tem_mask = tem_mask; p.ctx_mask = ctx_mask; p.flag = flag; p.ctx_roll = ctx_roll;
    p.query_embed = m->pr_query; p.logit_scale = m->pr_logit_scale; p.B = B; p.nz = m->nz; p.S = m->S; p.D = D;
    p.src = src; p.src0 = src0; p.src_bf16 = w.Xn;
    HIPCHK(launch_prompter_tokens(p, s));
    {   // src = mlp(src) + src  (utils.py:94)
        GemmParams g;
        g.A = w.Xn; g.lda = D; g.W = m->pr_w1; g.ldw = D; g.bias = m->pr_b1; g.M = 3 * B; g.N = Fn; g.K = D; g.epi = 0; g.C = w.Hb; g.ldc = Fn; g.act = 1; g.tune = &m->tune;
        HIPCHK(launch_gemm(g, s));
        GemmParams h;
        h.A = w.Hb; h.lda = Fn; h.W = m->pr_w2; h.ldw = Fn; h.bias = m->pr_b2; h.M = 3 * B; h.N = D; h.K = Fn; h.epi = 1; h.C = src; h.ldc = D; h.accumulate = 1; h.tune = &m->tune;
        HIPCHK(launch_gemm(h, s));
    }
    HIPCHK(launch_prompter_select(src, src0, flag, prompt_out, B, 3 * D, s));
    return UVL_OK;
}

// UVLTrack.forward (uvltrack.py:18-24): the head runs its no-prompt branch (head:123-138) -- prompter inline on a context
// rolled by half a batch, two-channel cont_score.
struct TrainBranch { const uint8_t* template_mask; const uint8_t* context_mask; float* prompts_out; };

// Does the text branch ride in the visual launches (single-stream frame), or run on its own stream?
// One sequence: always (unless pair_text = 0).  Many sequences: only where the visual GEMMs take the large-tile kernels, which have pair forms
// (launch_gemm_pair), and by default only where the riders measured ahead of the second stream (interleaved tools/ab_tune.py debug.pair_text 3 1,
// profiles/r04_text_branch.md): UVLTrack-L x 6 / 8 / 16 / 32 +0.7 / +1.9 / +1.4 / +0.3 % (x 4: -4.8 %, its kernels have no pair forms), UVLTrack-B x 32 +1.3..1.8 %,
// but UVLTrack-B x 12 / 16 / 24 -1.0 / -1.0 / -1.6 % (a rider tile
// streams its BERT weights at HBM latency -- three to ten row tiles per weight panel -- and holds its slot about twice as long as a visual
// tile; the six-layer branch of UVLTrack-B hides better on the second stream until the frame is long).
static bool text_rides(const uvl_model* m, int B, int skip, int reuse) {
    if (skip || reuse || m->nf <= 0 || !m->pair_text) return false;
    if (B == 1) return true;
    if (m->pair_text == 3) return false;
    const long rows = (long)B * m->nv;
    const bool forms = rows >= kDispatch.dr_min_rows && !m->bert.empty() && m->bert[0].pqkv && tune_get(&m->tune, &uvl_tuning::gemm_dr, -1) != 0 &&
                       tune_get(&m->tune, &uvl_tuning::gemm_pipe, 1) != 0 && m->tune.text_cfg < 0;
    if (!forms) return false;
    if (m->pair_text >= 2) return true;
    return rows >= kDispatch.rider_min_rows && (m->D >= kDispatch.rider_wide_dim || rows >= kDispatch.rider_any_rows) && m->tune.gemm_cfg < 0 && m->tune.attn_cfg < 0;
}

static int run_forward(uvl_model* m, const uvl_inputs* in, const uvl_outputs* out, void* d_ws, size_t ws_bytes, hipStream_t s, Profiler* prof,
                       int parts = PART_ALL, const TrainBranch* tb = nullptr) {
    if (!m || !in || !out) return fail(UVL_EINVAL, "null argument");
    if (!m->finalized) return fail(UVL_ESTATE, "uvl_finalize_weights has not been called");
    const int B = in->batch;
    if (B <= 0 || B > m->cfg.max_batch) return fail(UVL_EINVAL, "batch %d outside [1, %d]", B, m->cfg.max_batch);
    if (!in->d_template || !in->d_search || !in->d_prompt || !in->d_flag) return fail(UVL_EINVAL, "missing input pointer");
    const int skip = in->skip_text ? 1 : 0;
    if (!skip && (!in->d_text_ids || !in->d_text_mask)) return fail(UVL_EINVAL, "text inputs required unless skip_text");
    // reuse_text: the text branch below the first fusion layer depends on the text alone, and a tracker's text does not change
    // over a sequence -- take its results (final text rows + the per-layer snapshots for the logits) from the workspace, where
    // the last full call left them.  The frame is then the single-stream visual schedule with nj rows from layer nf on.
    const int reuse = (!skip && in->reuse_text && m->nf > 0 && m->nf < m->depth) ? 1 : 0;
    if (reuse && tb) return fail(UVL_EINVAL, "uvl_forward: reuse_text is a forward_test option");
    // a standing request for cfg 36 on the residual GEMMs (uvl_tune_set "gemm_dr" 1 / "gemm_cfg" 36 / "text_cfg" 36, uvl_debug_set "text_dr_res") needs their
    // fragment-native weight images: made once, here, on the caller's stream (allocation + pack + one synchronize of THAT stream) -- never inside a capture
    if ((m->tune.gemm_dr == 1 || m->tune.gemm_cfg == 36 || m->tune.text_cfg == 36 || m->text_dr_res) && (long)m->cfg.max_batch * m->nj >= kDispatch.dr_min_rows && !m->vit.empty() &&
        (!m->vit[0].pproj || !m->vit[0].pfc2 || (!m->bert.empty() && (!m->bert[0].pao || !m->bert[0].po)))) {
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cst) != hipSuccess) { (void)hipGetLastError(); cst = hipStreamCaptureStatusNone; }
        if (cst != hipStreamCaptureStatusNone) return fail(UVL_ESTATE, "cfg 36 on the residual GEMMs was requested but their weight images are not packed yet: run one eager frame before capturing");
        const int prc = pack_residual_images(m, s);
        if (prc) return prc;
    }
    const Workspace w = carve(m, B, (char*)d_ws);
    if (!d_ws || ws_bytes < w.total) return fail(UVL_EINVAL, "workspace too small: %zu < %zu", ws_bytes, w.total);
    if ((uintptr_t)d_ws % 256) return fail(UVL_EINVAL, "workspace must be 256-byte aligned");

    const int D = m->D, H = m->H, nz = m->nz, nx = m->nx, nv = m->nv, nj = m->nj, npad = m->npad, T = m->T, Fn = m->ffn;
    // Next-weight requests in the GEMM launches (common.h::prefetch_issue): frames below ~2000 visual rows -- launch-bound GEMMs that start on a cold
    // weight -- of a model whose weights cannot stay in the 256 MB memory-side cache from frame to frame.  Interleaved tools/ab_tune.py debug.prefetch_w 0 2:
    // UVLTrack-L (909 MB of weights) x 1 426-428 -> 437-443 frames/s (+2.6..3.4 %), x 2 596 -> 620 (+4.1 %), x 4 867 -> 861 (-0.7 %); UVLTrack-B (its 170 MB
    // of ViT weights stay resident, see launch_gemm_pair) x 1 1362-1375 -> 1362-1364, x 2 0, x 4 -3 %.  The text branch's weights (riders: each byte read
    // once, by one CU) gain nothing from it (UVLTrack-L x 1 +2.6 % with them against +3.4 % without).
    const bool pfw = m->prefetch_w == 2 || (m->prefetch_w == 1 && (long)B * nv < kDispatch.prefetch_max_rows && (double)m->depth * 12.0 * D * D * 2.0 > kDispatch.prefetch_min_weights);
    Launcher L{prof};
    L.parts = parts;
    // eager full-frame runs fork the text branch onto the library's second stream and join with events; a partial walk
    // (graph capture of one part) launches on `s` only and leaves the cross-part ordering to uvl_graph_launch
    // One sequence: every text-branch kernel rides in the SAME launch as the visual kernel of its kind (GEMM with GEMM,
    // attention with attention, LayerNorm with LayerNorm -- the two layer structures line up op for op), so the 40-token branch
    // costs neither launches nor a second queue; measured, the two-stream form slows each visual layer by ~11 us through
    // contention and ends level with visual layer nf-1 (profiles/r01_summary.md).  Larger batches keep the second stream.
    // Many sequences (uvl_debug_set "pair_text" 2): the same riders on the large-tile kernels -- the text GEMMs' 12-48 tiles behind the visual
    // tiles of gemm_dr_pair_kernel / gemm_pipe_pair_kernel, the 40 x 40 attention items behind the persistent walk (attn_p64_rider_kernel),
    // the LayerNorm rows in ln_pair_kernel; in-place residual epilogue for the text rows (no split-K slabs).
    const bool paired = text_rides(m, B, skip, reuse);
    const bool paired_many = paired && B > 1;
    // LayerNorm-free frame (round 6; fold.h, gemm_fin.hip): one or two sequences whose residual GEMMs are at most one eight-wave workgroup per CU.  No LayerNorm launch,
    // no split-K slabs: the residual GEMMs finish x in the launch and leave bf16 rows + partial statistics, QKV / fc1 (and the text riders' query/key/value and
    // intermediate GEMMs) run on the un-normalised rows with the LayerNorm folded into their weights, the logits ride on the next QKV launch, and ONE small launch
    // normalises the text rows where they join the visual rows.  96 -> 72 launches for one UVLTrack-B sequence.  Needs the single-stream frame (riders, or no text
    // branch), the 'cls' text token, a fusion tail (0 < nf < depth); tests cut below the first fusion layer on the LayerNorm-kernel schedule.
    const bool fold = m->fold_ln && B == 1 && (paired || skip || reuse) && !m->cfg.txt_token_mean && m->fuse_contrast && m->nf > 0 && m->nf < m->depth &&
                      m->D % 128 == 0 && (m->debug_stop_layer == -1 || m->debug_stop_layer >= m->nf) && !m->fuse_ln &&
                      (long)((B * m->nj + 63) / 64) * (m->D / 64) <= kDispatch.fold_max_tiles && m->tune.gemm_cfg < 0 && m->tune.text_cfg < 0 && !m->vit.empty() && m->vit[0].fqkv;
    // The text branch of a many-sequence frame (B x T rows: 320 at 8 sequences) overlaps the visual layers on the second stream, and what it
    // costs the frame is the CU time of its workgroups: as 64 x 64 tiles (240 workgroups of ~6 us per GEMM at ~15 % MFMA efficiency) that was
    // 8 % of the UVLTrack-L x 8 frame (1241 against 1351 frames/s without the branch).  On gemm_dr_kernel's 128 x 256 tiles the same GEMM is
    // 36 workgroups of ~10 us: a seventh of the CU time -- worth +0.7 % on that frame and +1.9 % on 32 UVLTrack-B sequences, and it LOSES where
    // the longer text kernels reach the critical path (8 UVLTrack-B sequences -3.4 %, 4: -17 %), so only frames of >= 6000 visual rows take it.
    // The rest of the branch's cost is not CU time (profiles/r04_text_branch.md).  In-place residual epilogue (no split-K slabs).
    // uvl_tuning.text_cfg >= 0 overrides the tile configuration as before; gemm_dr = 0 switches this off.
    const bool text_dr = !paired && (long)B * m->nv >= kDispatch.text_dr_min_rows && m->tune.text_cfg < 0 && tune_get(&m->tune, &uvl_tuning::gemm_dr, -1) != 0 && !m->bert.empty() && m->bert[0].pqkv;
    const bool fork = !skip && !reuse && !prof && m->nf > 0 && parts == PART_ALL && !paired && m->fork_text;
    hipStream_t sa = fork ? m->aux : s;          // text branch stream (serialised when profiling)
    enum { R_GEMM = 0, R_ATTN = 1, R_LN = 2 };
    struct Rider { int kind; const char* what; double flops, bytes, wbytes; GemmParams g; AttnParams a; LnParams l; int layer; };
    int rider_layer = 0;                         // BERT layer whose launches are being queued
    std::vector<Rider> riders;                   // text-branch launches waiting for a visual launch of the same kind
    size_t rider_at = 0;
    struct GemmPair { GemmParams a, b; };
    struct AttnPair { AttnParams a, b; };
    struct LnPair { LnParams a, b; };
    // launch a visual kernel; if the next waiting text kernel is of the same kind (and, for GEMMs, the same epilogue), take it along
    // One-sequence frames: a visual LayerNorm (with its text rider) is not launched but held, and the GEMM that consumes it -- always
    // the next visual launch: LN-1 -> QKV, LN-2 -> fc1 -- takes it along (launch_ln_gemm_pair: LayerNorm rows, grid barrier, GEMM tiles
    // in one launch).  Graph capture keeps the two-launch form: the barrier's generation is a launch argument.
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    if (m->fuse_ln && hipStreamIsCapturing(s, &cap_status) != hipSuccess) {      // (only the default-off fused launch needs to know)
        (void)hipGetLastError();                                                 // do not leave the error for the next launch check
        cap_status = hipStreamCaptureStatusActive;                               // unknown: keep the two-launch form
    }
    const bool fuse_ln = B == 1 && m->fuse_ln && m->gbar && cap_status == hipStreamCaptureStatusNone && (paired || skip || reuse);
    struct PendLn { bool on = false, has_b = false; LnParams a, b; double bytes = 0; } pend_ln;
    struct LnGemm { LnParams la, lb; GemmParams ga, gb; bool has_lb, has_gb; unsigned* bar; unsigned gen; unsigned* base; bool fused; };
    auto flush_ln = [&](hipStream_t st) {            // safety: a held LayerNorm whose consumer did not come next
        if (!pend_ln.on) return;
        pend_ln.on = false;
        if (pend_ln.has_b) {
            struct LnPair2 { LnParams a, b; } lp{pend_ln.a, pend_ln.b};
            L.run(st, "layernorm", 0, pend_ln.bytes, [](void* c, hipStream_t q) { auto* x = (LnPair2*)c; return launch_layernorm_pair(x->a, x->b, q); }, &lp);
        } else {
            L.run(st, "layernorm", 0, pend_ln.bytes, tramp<LnParams, launch_layernorm>, &pend_ln.a);
        }
    };
    auto run_gemm = [&](hipStream_t st, GemmParams& p, const char* what, bool is_text) {
        p.tune = &m->tune;
        if (is_text && !paired && m->tune.text_cfg >= 0) { m->tune_text = m->tune; m->tune_text.gemm_cfg = m->tune.text_cfg; p.tune = &m->tune_text; }
        else if (is_text && text_dr && p.Wp) { m->tune_text = m->tune; m->tune_text.gemm_cfg = 36; p.tune = &m->tune_text; }
        // BERT weights are read once per frame: up to four sequences (<= 3 M tiles share a weight tile through L2) they are loaded
        // non-temporal so that they do not displace the ViT weights from the Infinity Cache (+2-3 % at 2-4 sequences, -0.5 % from 8 on)
        if (is_text && p.M <= 192) {
            const int bit = !strcmp(what, "gemm.bert_qkv") ? 1 : !strcmp(what, "gemm.bert_ao") ? 2 : !strcmp(what, "gemm.bert_i") ? 4 : 8;
            p.w_stream = (m->text_nt & bit) ? 1 : 0;
        }
        p.rider_first = m->rider_first;
        // The bf16 rows gemm_dr_kernel stores (fc1: 57 MB per launch at 8 UVLTrack-L sequences; the q / k rows of QKV) go out write-through (sc1): nothing of them is left
        // dirty in the L2s for the end-of-kernel write-back, and the next kernel reads them from memory anyway (another XCD's L2 at best).  Same box, interleaved
        // tools/ab_tune.py debug.bf16_store 0 1 / 1 3 / 0 3: fc1 alone 1272 -> 1298 frames/s (+2.1 %; non-temporal instead: +1.9 %), the q / k rows +0.2 % more; 8 / 32
        // UVLTrack-B sequences +1.0 / +0.95 %, 4 UVLTrack-L sequences +1.0 %, 32 level.  Not for: LayerNorm's bf16 rows (bit 2: 0), the attention output (generator
        // option osc1: -1.6 %), V^T (2-byte scattered stores).
        if (!is_text && p.epi == 0 && (m->bf16_store & 1)) p.c_store = 2;
        if (!is_text && p.epi == 2 && (m->bf16_store & 2)) p.c_store = 2;
        // algorithmic bytes: both operands once + what the epilogue moves (bf16 rows; f32 rows, read too by the in-place residual form, once per slab)
        const double out_b = (double)p.M * p.N * (p.epi == 1 ? 4.0 * (p.accumulate ? 2 : 1) * (p.splitk > 1 ? p.splitk : 1) : 2.0);
        const double fl = 2.0 * p.M * p.N * p.K, by = 2.0 * ((double)p.M * p.K + (double)p.N * p.K) + out_b;
        const double wb = 2.0 * (double)p.N * p.K;          // the weight, once
        if (paired && is_text) { Rider r{}; r.kind = R_GEMM; r.what = what; r.flops = fl; r.bytes = by; r.wbytes = wb; r.g = p; r.layer = rider_layer; riders.push_back(r); return; }
        if (pend_ln.on && !is_text) {
            LnGemm c{};
            c.la = pend_ln.a; c.lb = pend_ln.b; c.has_lb = pend_ln.has_b; c.ga = p; c.has_gb = false; c.bar = m->gbar; c.gen = m->gbar_gen + 1; c.base = &m->gbar_base; c.fused = false;
            double fl2 = fl, by2 = by + pend_ln.bytes, wb2 = wb;
            if (paired && rider_at < riders.size() && riders[rider_at].kind == R_GEMM && riders[rider_at].g.epi == p.epi) {
                const Rider& r = riders[rider_at++];
                c.gb = r.g; c.has_gb = true; fl2 += r.flops; by2 += r.bytes; wb2 += r.wbytes;
            }
            pend_ln.on = false;
            L.wb_next = wb2;
            L.run(st, what, fl2, by2, [](void* cc, hipStream_t q) {
                auto* x = (LnGemm*)cc;
                return launch_ln_gemm_pair(x->la, x->has_lb ? &x->lb : nullptr, x->ga, x->has_gb ? &x->gb : nullptr, x->bar, x->gen, x->base, &x->fused, q);
            }, &c);
            if (c.fused) m->gbar_gen = c.gen;
            return;
        }
        if (paired && rider_at < riders.size() && riders[rider_at].kind == R_GEMM && riders[rider_at].g.epi == p.epi) {
            GemmPair gp{p, riders[rider_at].g};
            const Rider& r = riders[rider_at++];
            L.wb_next = wb + r.wbytes;
            L.run(st, what, fl + r.flops, by + r.bytes, [](void* c, hipStream_t q) { auto* x = (GemmPair*)c; return launch_gemm_pair(x->a, x->b, q); }, &gp);
            return;
        }
        L.wb_next = wb;
        L.run(st, what, fl, by, tramp<GemmParams, launch_gemm>, &p);
    };
    auto run_attn = [&](hipStream_t st, AttnParams& p, const char* what, double fl, double by, bool is_text) {
        p.tune = &m->tune;
        if (!is_text) flush_ln(st);
        if (paired && is_text) { Rider r{}; r.kind = R_ATTN; r.what = what; r.flops = fl; r.bytes = by; r.wbytes = by; r.a = p; r.layer = rider_layer; riders.push_back(r); return; }
        if (paired && rider_at < riders.size() && riders[rider_at].kind == R_ATTN) {
            AttnPair ap{p, riders[rider_at].a};
            const Rider& r = riders[rider_at++];
            L.run(st, what, fl + r.flops, by + r.bytes, [](void* c, hipStream_t q) { auto* x = (AttnPair*)c; return launch_attention_pair(x->a, x->b, q); }, &ap);
            return;
        }
        L.run(st, what, fl, by, tramp<AttnParams, launch_attention>, &p);
    };
    auto run_ln = [&](hipStream_t st, LnParams& p, double by, bool is_text) {
        if (paired && is_text) { Rider r{}; r.kind = R_LN; r.what = "layernorm"; r.flops = 0; r.bytes = by; r.wbytes = by; r.l = p; r.layer = rider_layer; riders.push_back(r); return; }
        flush_ln(st);
        if (fuse_ln && !is_text && p.y_bf16 && (L.parts & L.cur)) {      // held for the GEMM that reads p.y_bf16
            pend_ln.on = true; pend_ln.a = p; pend_ln.has_b = false; pend_ln.bytes = by;
            if (paired && rider_at < riders.size() && riders[rider_at].kind == R_LN && riders[rider_at].l.D == p.D) {
                const Rider& r = riders[rider_at++];
                pend_ln.b = r.l; pend_ln.has_b = true; pend_ln.bytes += r.bytes;
            }
            return;
        }
        if (paired && rider_at < riders.size() && riders[rider_at].kind == R_LN && riders[rider_at].l.D == p.D) {
            LnPair lp{p, riders[rider_at].l};
            const Rider& r = riders[rider_at++];
            L.run(st, "layernorm", 0, by + r.bytes, [](void* c, hipStream_t q) { auto* x = (LnPair*)c; return launch_layernorm_pair(x->a, x->b, q); }, &lp);
            return;
        }
        L.run(st, "layernorm", 0, by, tramp<LnParams, launch_layernorm>, &p);
    };
    auto flush_riders = [&](hipStream_t st, int upto_layer = 1 << 30) {    // whatever did not find a partner runs alone, in order
        for (; rider_at < riders.size() && riders[rider_at].layer <= upto_layer; ++rider_at) {
            Rider& r = riders[rider_at];
            if (r.kind == R_GEMM) { L.wb_next = r.wbytes; L.run(st, r.what, r.flops, r.bytes, tramp<GemmParams, launch_gemm>, &r.g); }
            else if (r.kind == R_ATTN) L.run(st, r.what, r.flops, r.bytes, tramp<AttnParams, launch_attention>, &r.a);
            else L.run(st, r.what, 0, r.bytes, tramp<LnParams, launch_layernorm>, &r.l);
        }
    };

    // -- setup: masks, cls rows (visual part) / BERT additive mask (text part)
    struct SetupCtx { const uvl_model* m; const uvl_inputs* in; Workspace w; int B, skip, what; } sc{m, in, w, B, skip, 1};
    auto setup_fn = [](void* c, hipStream_t st) {
        auto* x = (SetupCtx*)c;
        return launch_setup(x->in->d_text_mask, x->in->d_flag, x->m->cls_token, x->w.X, x->w.key_add, x->w.bert_add, x->B,
                            x->m->nz, x->m->nv, x->m->nj, x->m->npad, x->m->T, x->m->D, x->skip, x->what, st);
    };
    if (fork) {
        if (hipEventRecord(m->ev_fork, s) != hipSuccess || hipStreamWaitEvent(sa, m->ev_fork, 0) != hipSuccess) return fail(UVL_EHIP, "fork failed");
    }
    L.cur = PART_V1;
    PrologueParams pro;
    const bool one_queue = paired || skip || reuse;   // single-stream frame: set-up, BERT embedding (if any) and im2row in ONE launch
    if (one_queue) {
        pro.text_mask = in->d_text_mask; pro.flag = in->d_flag; pro.cls_token = m->cls_token; pro.x = w.X; pro.key_add = w.key_add;
        pro.bert_add = w.bert_add; pro.nz = nz; pro.nv = nv; pro.nj = nj; pro.npad = npad; pro.T = T; pro.D = D; pro.B = B;
        pro.ids = paired ? in->d_text_ids : nullptr; pro.skip_text = skip; pro.setup_what = skip ? 1 : 3; pro.word = m->word; pro.pos = m->pos; pro.type0 = m->type0; pro.emb_g = m->emb_g; pro.emb_b = m->emb_b;
        pro.tn = w.Tn; pro.vocab = m->cfg.vocab;
        pro.z = in->d_template; pro.ximg = in->d_search; pro.patches = w.P; pro.Hz = m->cfg.template_size; pro.Hx = m->cfg.search_size;
        if (fold) {         // the [cls] row as layer 0's QKV GEMM reads it; the embedding rows stay pre-norm (their LayerNorm is folded into the first query/key/value GEMM)
            pro.cls_xn = w.Xn; pro.cls_xn_bs = nv; pro.cls_st = w.St; pro.cls_st_rows = B * nv;
            pro.embed_raw = paired ? 1 : 0; pro.embed_st = w.StT0;
        }
        L.run(s, "prologue", 0, 0, tramp<PrologueParams, launch_prologue>, &pro);
    } else
    L.run(s, "setup", 0, 0, setup_fn, &sc);
    SetupCtx sct = sc;
    sct.what = 2;
    if (!one_queue) { L.cur = PART_TEXT; L.run(sa, "setup", 0, 0, setup_fn, &sct); L.cur = PART_V1; }
    // -- text branch (extractor.py:54,62): embedding + the first nf BERT layers depend on the text only, so the whole
    //    chain is enqueued up front on its own stream; layers whose output the contrastive logits need leave a snapshot
    Pending pend_t;
    auto consume = [](LnParams& p, Pending& pd) { p.part = pd.part; p.nsplit = pd.nsplit; p.part_rows = pd.rows; p.part_stride = pd.stride; pd = Pending(); };
    auto is_cont_layer = [&](int i) { bool c = false; for (int k = 0; k < m->cfg.n_cont; ++k) c |= (m->cfg.cont_layers[k] == i); return c; };
    // tab (optional): a [2, D] table whose row (t >= tab_split) is added too -- only the in-place form takes it; returns whether it did
    auto residual_gemm = [&](hipStream_t st, const char* what, const bf16_t* A, int lda, const bf16_t* Wt, const float* bias, int Mr, int K,
                             int rpb, int oro, float* slab, Pending& pd, bool allow_split, bool is_text = false, const bf16_t* Wpk = nullptr,
                             const float* tab = nullptr, int tab_split = 0, const void* pf = nullptr, size_t pf_bytes = 0) -> bool {
        GemmParams p;
        p.A = A; p.lda = lda; p.W = Wt; p.Wp = Wpk; p.ldw = K; p.bias = bias; p.M = Mr; p.N = D; p.K = K; p.epi = 1; p.ldc = D;
        p.pf = pf; p.pf_bytes = (uint32_t)pf_bytes;
        // rider_sk: the text branch's OUTPUT GEMM riding in a many-sequence fc2 launch.  Its 3 x 4 tiles walk the same 64 K tiles as the visual tiles, on
        // BERT weights nobody has read this frame: each one ran ~10 % longer than a visual tile and was the tail of the launch (+6.6 us on 12 launches
        // of 8 UVLTrack-L sequences).  Two K halves (f32 slabs, folded by the BERT LayerNorm that follows, as in one-sequence frames) end under the visual tiles.
        const int rsk = (is_text && paired_many && !allow_split && K >= 4096 && (K / 64) % 2 == 0 && m->rider_sk > 1) ? 2 : 1;
        const int sk = rsk > 1 ? rsk : (allow_split ? choose_splitk(Mr, D, K, &m->tune) : 1);
        if (sk > 1) {             // slabs [sk][Mr, D], folded in by the next LayerNorm / contrast kernel
            p.C = slab; p.splitk = sk; p.part_stride = (size_t)Mr * D; p.c_store = tune_get(&m->tune, &uvl_tuning::slab_store, 2);   // write-through slabs: +1 % at one sequence (both A/B orders)
            pd.part = slab; pd.nsplit = sk; pd.rows = rpb; pd.stride = p.part_stride;
        } else {                  // x += A W^T + b in place
            // write-through (sc1) stores of x: no dirty lines wait for the end-of-kernel write-back, and the LayerNorm that follows reads
            // x from memory, not from another XCD's L2 anyway (+0.9 % at 8 sequences of UVLTrack-B / -L, 0 at 32; tools/ab_tune.py res_store 0 2)
            p.C = w.X; p.accumulate = 1; p.rpb = rpb; p.obs = nj; p.oro = oro; p.c_store = tune_get(&m->tune, &uvl_tuning::res_store, 2);
            if (tab) { p.addtab = tab; p.addtab_split = tab_split; }
        }
        run_gemm(st, p, what, is_text);
        return sk <= 1 && tab != nullptr;
    };
    if (!one_queue) {
        struct BeCtx { const uvl_model* m; const uvl_inputs* in; Workspace w; int B; } bc{m, in, w, B};
        L.cur = PART_TEXT;
        L.run(sa, "bert_embed", 0, 0, [](void* c, hipStream_t st) {
            auto* x = (BeCtx*)c;
            return launch_bert_embed(x->in->d_text_ids, x->m->word, x->m->pos, x->m->type0, x->m->emb_g, x->m->emb_b, x->w.X, x->m->nj, x->m->nv,
                                     x->w.Tn, x->B, x->m->T, x->m->D, x->m->cfg.vocab, st);
        }, &bc);
        L.cur = PART_V1;
    }
    // (stop_layer -2: no layer at all -- the residual stream after the patch embedding / the BERT embedding goes straight to the head's output copies: tests localise the input side with it)
    const int last_bert = (skip || reuse) ? -1 : (m->debug_stop_layer == -2 ? -1 : (m->debug_stop_layer >= 0 && m->debug_stop_layer < m->nf - 1) ? m->debug_stop_layer : m->nf - 1);
    int text_err = 0;
    // one BERT layer (BertLayer.forward, bert_backbone.py:390-394); launched interleaved with the visual layers so both
    // hardware queues are fed in step (the host enqueues ~3.5 us per launch)
    auto text_layer = [&](int i) {
            const int saved_part = L.cur;
            L.cur = PART_TEXT;
            const BertLayerW& bw = m->bert[i];
            const int Mt = B * T;
            {
                GemmParams p;
                p.A = w.Tn; p.lda = D; p.W = bw.wqkv; p.Wp = bw.pqkv; p.ldw = D; p.bias = bw.bqkv; p.M = Mt; p.N = 3 * D; p.K = D;
                p.epi = 2; p.rpb = T; p.q = w.Tq; p.k = w.Tk; p.vt = w.Tvt; p.H = H; p.Npad = 64; p.D = D; p.q_scale = UVL_QSCALE;
                run_gemm(sa, p, "gemm.bert_qkv", true);
            }
            {
                AttnParams p;
                p.q = w.Tq; p.k = w.Tk; p.vt = w.Tvt; p.key_add = w.bert_add; p.key_add_stride = 64; p.o = w.To; p.B = B; p.H = H; p.N = T; p.Npad = 64; p.q_prescaled = 1;
                run_attn(sa, p, "attention.bert", 4.0 * T * (double)T * D * B, 8.0 * Mt * D, true);
            }
            residual_gemm(sa, "gemm.bert_ao", w.To, D, bw.wao, bw.bao, Mt, D, T, nv, w.PartT, pend_t, !(text_dr && m->text_dr_res) && !paired_many, true, (text_dr && m->text_dr_res) ? bw.pao : nullptr);
            {
                LnParams p;        // post-LN in place on the text rows
                p.x = w.X; p.M = Mt; p.D = D; p.rpb = T; p.xbs = nj; p.xro = nv;
                consume(p, pend_t);
                p.gamma = bw.ln1g; p.beta = bw.ln1b; p.eps = 1e-12f; p.y_bf16 = w.Tn; p.y_f32 = w.X; p.y_remap = 1;
                run_ln(sa, p, (double)Mt * D * 10, true);
            }
            {
                GemmParams p;
                p.A = w.Tn; p.lda = D; p.W = bw.wi; p.Wp = bw.pi; p.ldw = D; p.bias = bw.bi; p.M = Mt; p.N = Fn; p.K = D;
                p.epi = 0; p.C = w.Th; p.ldc = Fn; p.act = 1;
                run_gemm(sa, p, "gemm.bert_i", true);
            }
            residual_gemm(sa, "gemm.bert_o", w.Th, Fn, bw.wo, bw.bo, Mt, Fn, T, nv, w.PartT, pend_t, !(text_dr && m->text_dr_res) && !paired_many, true, (text_dr && m->text_dr_res) ? bw.po : nullptr);
            {
                LnParams p;
                p.x = w.X; p.M = Mt; p.D = D; p.rpb = T; p.xbs = nj; p.xro = nv;
                consume(p, pend_t);
                p.gamma = bw.ln2g; p.beta = bw.ln2b; p.eps = 1e-12f; p.y_bf16 = w.Tn; p.y_f32 = w.X; p.y_remap = 1;
                // this layer's text rows: for the logits, and (last BERT layer) for later frames that reuse the text branch
                if ((is_cont_layer(i) && out->d_logits) || i == m->nf - 1) p.y_copy = w.TxtSnap + (size_t)i * Mt * D;
                run_ln(sa, p, (double)Mt * D * 10, true);
            }
            if (fork && is_cont_layer(i) && out->d_logits) {
                if (hipEventRecord(m->ev_bert[i], sa) != hipSuccess) text_err = fail(UVL_EHIP, "bert event failed");
            }
        if (i == last_bert && fork && hipEventRecord(m->ev_join, sa) != hipSuccess) text_err = fail(UVL_EHIP, "join record failed");
        L.cur = saved_part;
    };
    if (last_bert < 0 && fork && hipEventRecord(m->ev_join, sa) != hipSuccess) return fail(UVL_EHIP, "join record failed");
    if (paired && !fold) {                       // parameters of the whole text branch, in order; launched as riders below
        for (int i = 0; i <= last_bert; ++i) { rider_layer = i; text_layer(i); }
        if (text_err) return text_err;
    }
    // -- patch embed (mae_vit.py:203-215)
    struct ImCtx { const uvl_inputs* in; Workspace w; int B, hz, hx; } ic{in, w, B, m->cfg.template_size, m->cfg.search_size};
    if (!one_queue) L.run(s, "im2row", 0, 0, [](void* c, hipStream_t st) { auto* x = (ImCtx*)c; return launch_im2row(x->in->d_template, x->in->d_search, x->w.P, x->B, x->hz, x->hx, st); }, &ic);
    {
        GemmParams p;
        p.A = w.P; p.lda = 768; p.W = m->w_patch; p.ldw = 768; p.bias = m->b_patch;
        p.M = B * (nz + nx); p.N = D; p.K = 768; p.epi = 1; p.C = w.X; p.ldc = D;
        p.rpb = nz + nx; p.obs = nj; p.oro = 1; p.addtab = m->pos_tab; p.tune = &m->tune;
        if (pfw && m->depth > 0) { p.pf = fold ? m->vit[0].fqkv : m->vit[0].wqkv; p.pf_bytes = (uint32_t)((size_t)3 * D * D * 2); }
        if (fold) {         // finished in the launch, + the bf16 rows and partial statistics layer 0's QKV GEMM reads (its norm1 is folded into that GEMM)
            p.xn = w.Xn; p.xn_bs = nv; p.xn_ro = 1; p.st_out = w.St; p.st_rows = B * nv;
            L.wb_next = 2.0 * (double)p.N * p.K;
            L.run(s, "gemm.patch", 2.0 * p.M * p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K) + 6.0 * p.M * p.N,
                  [](void* c, hipStream_t q) { return launch_gemm_fin(*(const GemmParams*)c, nullptr, q); }, &p);
        } else
        RUN_GEMM(L, s, p, "gemm.patch");
    }
    int cont_slot = 0;
    int head_ct_slot = -1;                       // >= 0: head_prep also writes the last layer's logits into this slot
    int fused_ct = -1, fused_slot = 0;           // contrast layer whose logits the next layer's LayerNorm-2 will write
    bool direct_ct = false;                      // ... or the next layer's LayerNorm-1 (many-sequence frames, see below)
    bool modal_folded = false;                   // the last fc2 epilogue has added the next (fusion) layer's modal embedding
    Pending pend_v;                              // split-K slabs not yet folded into the residual stream (visual/joint rows)
    if (fold) {
        // ---- the LayerNorm-free layer walk: per ViT block QKV (folded norm1) -> attention -> proj (finishes x) -> fc1 (folded norm2) -> fc2 (finishes x), five
        //      launches; BERT layer i's five GEMM / attention kernels ride in ViT block i's launches of the same kind (block.py:29-32, bert_backbone.py:390-394)
        struct FinCtx { GemmParams a, b; bool has_b; };
        struct LnfCtx { GemmParams a, b; CtJob ct; bool has_b, has_ct; };
        auto cost = [](const GemmParams& p, double& fl, double& by, double& wb) {
            fl += 2.0 * p.M * p.N * p.K;
            wb += 2.0 * (double)p.N * p.K;
            by += 2.0 * ((double)p.M * p.K + (double)p.N * p.K) + (double)p.M * p.N * (p.epi == 1 ? (p.accumulate ? 8.0 : 4.0) + (p.xn ? 2.0 : 0.0) : 2.0);
        };
        auto run_fin = [&](const char* what, GemmParams& a, GemmParams* b) {
            a.tune = &m->tune;
            a.c_store = tune_get(&m->tune, &uvl_tuning::res_store, 0);
            if (b) { b->tune = &m->tune; b->c_store = a.c_store; }
            FinCtx c{a, b ? *b : a, b != nullptr};
            double fl = 0, by = 0, wb = 0;
            cost(a, fl, by, wb);
            if (b) cost(*b, fl, by, wb);
            L.wb_next = wb;
            L.run(s, what, fl, by, [](void* cc, hipStream_t q) { auto* x = (FinCtx*)cc; return launch_gemm_fin(x->a, x->has_b ? &x->b : nullptr, q); }, &c);
        };
        auto run_lnf = [&](const char* what, GemmParams& a, GemmParams* b, const CtJob* ct) {
            a.tune = &m->tune;
            LnfCtx c{a, b ? *b : a, ct ? *ct : CtJob(), b != nullptr, ct != nullptr};
            double fl = 0, by = 0, wb = 0;
            cost(a, fl, by, wb);
            if (b) cost(*b, fl, by, wb);
            L.wb_next = wb;
            L.run(s, what, fl, by, [](void* cc, hipStream_t q) { auto* x = (LnfCtx*)cc; return launch_gemm_lnf(x->a, x->has_b ? &x->b : nullptr, x->has_ct ? &x->ct : nullptr, q); }, &c);
        };
        const int tl = paired ? last_bert : -1;              // BERT layers [0, tl] ride on ViT blocks [0, tl]
        const int Mt = B * T;
        // partial statistics of the text rows' current bf16 copy: StT0 after the embedding and after every output GEMM (pre-norm u2), StT1 after attention.output (u1)
        CtJob ctj;
        bool have_ct = false;
        for (int i = 0; i < m->depth; ++i) {
            const bool joint = i >= m->nf;
            const int N = (joint && !skip) ? nj : nv;
            const int M = B * N;
            const VitBlockW& vw = m->vit[i];
            const bool last = (i == m->depth - 1) || (m->debug_stop_layer == i);
            const bool rider = i <= tl;
            const bool next_joint = !last && i + 1 >= m->nf;
            const int Nn = (next_joint && !skip) ? nj : nv;      // rows per sample of the NEXT block: the row pitch of the bf16 copy fc2 leaves for its QKV GEMM
            if (joint && i == m->nf && !skip) {
                // the text rows join (extractor.py:62-63): output.LayerNorm of the last BERT layer (or the rows a previous frame kept), snapshot for the logits / for
                // frames that reuse the branch, + modal_embed[1] (mae_vit.py:196), bf16 copy + partials beside the visual rows'
                L.cur = PART_V2;
                TextJoinParams tj;
                tj.x = w.X; tj.xbs = nj; tj.xro = nv; tj.B = B; tj.T = T; tj.D = D;
                tj.alt = reuse ? w.TxtSnap + (size_t)(m->nf - 1) * Mt * D : nullptr;
                tj.gamma = m->bert[m->nf - 1].ln2g; tj.beta = m->bert[m->nf - 1].ln2b; tj.eps = 1e-12f;
                tj.snap = reuse ? nullptr : w.TxtSnap + (size_t)(m->nf - 1) * Mt * D;
                tj.add = m->modal + D; tj.xn = w.Xn; tj.xn_bs = nj; tj.xn_ro = nv; tj.st = w.St; tj.st_rows = B * nj;
                L.run(s, "layernorm", 0, (double)Mt * D * 14, tramp<TextJoinParams, launch_text_join>, &tj);
            }
            const BertLayerW* bw = rider ? &m->bert[i] : nullptr;
            {   // attn.qkv on norm1(x) (block.py:30,49): norm1 folded into the weight, the rows' statistics from the partials
                GemmParams p;
                p.A = w.Xn; p.lda = D; p.W = vw.fqkv; p.ldw = D; p.bias = vw.fbqkv; p.colsum = vw.csqkv; p.st_in = w.St; p.ln_eps = 1e-6f; p.M = M; p.N = 3 * D; p.K = D;
                p.epi = 2; p.rpb = N; p.q = w.Q; p.k = w.K; p.vt = w.Vt; p.H = H; p.Npad = npad; p.D = D; p.q_scale = UVL_QSCALE;
                if (pfw) { p.pf = vw.wproj; p.pf_bytes = (uint32_t)((size_t)D * D * 2); }
                GemmParams t;
                if (rider) {
                    t.A = w.Tn; t.lda = D; t.W = bw->fqkv; t.ldw = D; t.bias = bw->fbqkv; t.colsum = bw->csqkv; t.st_in = w.StT0; t.ln_eps = 1e-12f; t.M = Mt; t.N = 3 * D; t.K = D;
                    t.epi = 2; t.rpb = T; t.q = w.Tq; t.k = w.Tk; t.vt = w.Tvt; t.H = H; t.Npad = 64; t.D = D; t.q_scale = UVL_QSCALE;
                    if (i < m->rider_pf) { t.pf2 = bw->wao; t.pf2_tiles = D / 32; t.pf2_lp = D / 2; }        // attention.output.dense: 32-row panels of K = D
                }
                run_lnf("gemm.qkv", p, rider ? &t : nullptr, have_ct ? &ctj : nullptr);
                have_ct = false;
            }
            {
                AttnParams p;
                p.q = w.Q; p.k = w.K; p.vt = w.Vt; p.key_add = w.key_add; p.key_add_stride = npad; p.o = w.O; p.B = B; p.H = H; p.N = N; p.Npad = npad; p.q_prescaled = 1; p.tune = &m->tune;
                const double fl = 4.0 * N * (double)N * D * B, by = 8.0 * M * D;
                if (rider) {
                    AttnPair ap;
                    ap.a = p;
                    ap.b.q = w.Tq; ap.b.k = w.Tk; ap.b.vt = w.Tvt; ap.b.key_add = w.bert_add; ap.b.key_add_stride = 64; ap.b.o = w.To; ap.b.B = B; ap.b.H = H; ap.b.N = T; ap.b.Npad = 64;
                    ap.b.q_prescaled = 1; ap.b.tune = &m->tune;
                    L.run(s, "attention", fl + 4.0 * T * (double)T * D * B, by + 8.0 * Mt * D, [](void* c, hipStream_t q) { auto* x = (AttnPair*)c; return launch_attention_pair(x->a, x->b, q); }, &ap);
                } else {
                    L.run(s, "attention", fl, by, tramp<AttnParams, launch_attention>, &p);
                }
            }
            {   // x += attn.proj(o) (block.py:29-30,44), finished in the launch; rider: u1 = LayerNorm_prev(u) + attention.output.dense(o) (bert_backbone.py:335-339)
                GemmParams p;
                p.A = w.O; p.lda = D; p.W = vw.wproj; p.ldw = D; p.bias = vw.bproj; p.M = M; p.N = D; p.K = D; p.epi = 1; p.C = w.X; p.ldc = D; p.accumulate = 1;
                p.rpb = N; p.obs = nj; p.oro = 0; p.xn = w.Xn; p.xn_bs = N; p.xn_ro = 0; p.st_out = w.St; p.st_rows = M;
                if (pfw) { p.pf = vw.ffc1; p.pf_bytes = (uint32_t)((size_t)Fn * D * 2); }
                GemmParams t;
                if (rider) {
                    t.A = w.To; t.lda = D; t.W = bw->wao; t.ldw = D; t.bias = bw->bao; t.M = Mt; t.N = D; t.K = D; t.epi = 1; t.C = w.X; t.ldc = D; t.accumulate = 1;
                    t.rpb = T; t.obs = nj; t.oro = nv; t.xn = w.Tn; t.xn_bs = T; t.xn_ro = 0; t.st_out = w.StT1; t.st_rows = Mt;
                    // the residual is the LayerNorm in front of this layer applied to the stored pre-norm rows: the embedding LayerNorm (layer 0) or output.LayerNorm of layer i - 1
                    t.res_st = w.StT0; t.res_g = i == 0 ? m->emb_g : m->bert[i - 1].ln2g; t.res_b = i == 0 ? m->emb_b : m->bert[i - 1].ln2b; t.res_eps = 1e-12f;
                    if (i > 0 && is_cont_layer(i - 1) && out->d_logits) t.res_copy = w.TxtSnap + (size_t)(i - 1) * Mt * D;      // layer i - 1's text rows, for frames that reuse the branch
                    if (i < m->rider_pf) { t.pf2 = bw->fi; t.pf2_tiles = Fn / 64; t.pf2_lp = D; }             // intermediate.dense: 64-row panels of K = D
                }
                run_fin("gemm.proj", p, rider ? &t : nullptr);
            }
            {   // mlp.fc1 on norm2(x) + GELU (block.py:31, backbones/utils.py:58-60); rider: intermediate.dense on attention.output.LayerNorm(u1) (bert_backbone.py:366)
                GemmParams p;
                p.A = w.Xn; p.lda = D; p.W = vw.ffc1; p.ldw = D; p.bias = vw.fbfc1; p.colsum = vw.csfc1; p.st_in = w.St; p.ln_eps = 1e-6f; p.M = M; p.N = Fn; p.K = D;
                p.epi = 0; p.C = w.Hb; p.ldc = Fn; p.act = 1;
                if (pfw) { p.pf = vw.wfc2; p.pf_bytes = (uint32_t)((size_t)Fn * D * 2); }
                GemmParams t;
                if (rider) {
                    t.A = w.Tn; t.lda = D; t.W = bw->fi; t.ldw = D; t.bias = bw->fbi; t.colsum = bw->csi; t.st_in = w.StT1; t.ln_eps = 1e-12f; t.M = Mt; t.N = Fn; t.K = D;
                    t.epi = 0; t.C = w.Th; t.ldc = Fn; t.act = 1;
                    if (i < m->rider_pf) { t.pf2 = bw->wo; t.pf2_tiles = D / 32; t.pf2_lp = Fn / 2; }         // output.dense: 32-row panels of K = 4 D
                }
                run_lnf("gemm.fc1", p, rider ? &t : nullptr, nullptr);
            }
            {   // x += mlp.fc2(h) (block.py:31-32), finished in the launch, + the NEXT fusion layer's modal embedding (mae_vit.py:196: a permanent change of the stream);
                // rider: u2 = attention.output.LayerNorm(u1) + output.dense(h) (bert_backbone.py:376-380)
                GemmParams p;
                p.A = w.Hb; p.lda = Fn; p.W = vw.wfc2; p.ldw = Fn; p.bias = vw.bfc2; p.M = M; p.N = D; p.K = Fn; p.epi = 1; p.C = w.X; p.ldc = D; p.accumulate = 1;
                p.rpb = N; p.obs = nj; p.oro = 0;
                if (!last) { p.xn = w.Xn; p.xn_bs = Nn; p.xn_ro = 0; p.st_out = w.St; p.st_rows = B * Nn; }
                if (next_joint) { p.addtab = m->modal; p.addtab_split = nv; }
                if (pfw) {
                    p.pf = (i + 1 < m->depth && !last) ? (const void*)m->vit[i + 1].fqkv : (const void*)m->conv[0].w;
                    p.pf_bytes = (uint32_t)((i + 1 < m->depth && !last) ? (size_t)3 * D * D * 2 : (size_t)4 * m->conv[0].cout * 9 * m->conv[0].cin * 2);
                }
                GemmParams t;
                if (rider) {
                    t.A = w.Th; t.lda = Fn; t.W = bw->wo; t.ldw = Fn; t.bias = bw->bo; t.M = Mt; t.N = D; t.K = Fn; t.epi = 1; t.C = w.X; t.ldc = D; t.accumulate = 1;
                    t.rpb = T; t.obs = nj; t.oro = nv; t.xn = w.Tn; t.xn_bs = T; t.xn_ro = 0; t.st_out = w.StT0; t.st_rows = Mt;
                    t.res_st = w.StT1; t.res_g = bw->ln1g; t.res_b = bw->ln1b; t.res_eps = 1e-12f;
                    if (i + 1 < m->rider_pf && i + 1 <= tl) { t.pf2 = m->bert[i + 1].fqkv; t.pf2_tiles = 3 * D / 64; t.pf2_lp = D; }      // the next layer's query / key / value
                }
                run_fin("gemm.fc2", p, rider ? &t : nullptr);
            }
            // ---- contrastive logits of this layer (extractor.py:64-65,85-93): on the next block's QKV launch, or in the head prologue for the last layer ----
            if (is_cont_layer(i)) {
                if (out->d_logits && last && joint && i == m->depth - 1 && m->debug_stop_layer < 0) {
                    head_ct_slot = cont_slot;
                } else if (out->d_logits && !last) {
                    ctj = CtJob();
                    ctj.x = w.X; ctj.xbs = nj; ctj.D = D; ctj.B = B; ctj.nz = nz; ctj.nv = nv; ctj.nx = nx; ctj.skip_text = skip;
                    if (next_joint) { ctj.sub_vis = m->modal; ctj.sub_txt = m->modal + D; }     // fc2 above has added them; this layer's output is without
                    if (!skip && !joint) {
                        if (i == m->nf - 1 || reuse) { ctj.txt = w.TxtSnap + (size_t)i * Mt * D; ctj.txt_bs = T; }       // normalised rows: the text join's / an earlier frame's snapshot
                        else {                 // the pre-norm rows the output GEMM has just left, normalised by the job (the same bits the snapshot will hold)
                            ctj.txt = w.X + (size_t)nv * D; ctj.txt_bs = nj; ctj.txt_g = m->bert[i].ln2g; ctj.txt_b = m->bert[i].ln2b; ctj.txt_eps = 1e-12f;
                            ctj.txt_st = w.StT0; ctj.txt_st_bs = T; ctj.txt_st_rows = Mt;
                        }
                    }
                    ctj.flag = in->d_flag; ctj.logit_scale = m->logit_scale_bb; ctj.logits = out->d_logits; ctj.slot = cont_slot; ctj.ncont = m->cfg.n_cont;
                    have_ct = true;
                } else if (out->d_logits) {        // a frame cut at this (fusion) layer: x is complete and carries no later layer's embedding -- the stand-alone kernel
                    ContrastParams p;
                    L.cur = PART_V2;
                    p.x = w.X; p.nj = nj; p.nz = nz; p.nx = nx; p.nv = nv; p.D = D; p.T = T; p.B = B;
                    p.text_mask = in->d_text_mask; p.flag = in->d_flag; p.logit_scale = m->logit_scale_bb;
                    p.mean_mode = 0; p.skip_text = skip; p.logits = out->d_logits; p.slot = cont_slot; p.n_cont = m->cfg.n_cont;
                    L.run(s, "contrast", 0, 0, tramp<ContrastParams, launch_contrast>, &p);
                }
                ++cont_slot;
            }
            if (m->debug_stop_layer == i) break;
        }
        if (have_ct) return fail(UVL_ESTATE, "internal: logits job left unlaunched");
    } else
    for (int i = 0; i < m->depth; ++i) {
        if (m->debug_stop_layer == -2) break;
        const bool joint = i >= m->nf;
        const int N = (joint && !skip) ? nj : nv;
        const int M = B * N;
        const VitBlockW& vw = m->vit[i];
        const bool last = (i == m->depth - 1) || (m->debug_stop_layer == i);
        if (joint && i == m->nf) {               // first fusion layer reads the text rows
            if (paired) flush_riders(s);         // the text branch's last LayerNorm has no visual partner before this point
            if (!skip) L.cur = PART_V2;
            if (fork && hipStreamWaitEvent(s, m->ev_join, 0) != hipSuccess) return fail(UVL_EHIP, "join failed");
        }
        // ---- ViT block (block.py:29-32) ----
        {
            LnParams p;
            p.x = w.X; p.M = M; p.D = D; p.rpb = N; p.xbs = nj; p.xro = 0;
            consume(p, pend_v);
            if (joint) {                             // forward_joint, mae_vit.py:196: img_feat + modal_embed[0], txt_feat + modal_embed[1]
                // modal_folded: the previous layer's fc2 epilogue has added it already (in-place residual form; the text rows of the FIRST
                // fusion layer come from the text branch and still take theirs here)
                p.pre_add0 = modal_folded ? nullptr : m->modal;
                p.pre_add1 = (modal_folded && i > m->nf) ? nullptr : m->modal + D;
                p.split = nv;
            }
            if (reuse && i == m->nf) { p.x_alt = w.TxtSnap + (size_t)(m->nf - 1) * B * T * D; p.x_alt_rows = T; }   // text rows kept from the last full frame
            p.gamma = vw.ln1g; p.beta = vw.ln1b; p.eps = 1e-6f; p.y_bf16 = w.Xn; p.y_wt = (m->bf16_store & 4) ? 1 : 0;
            if (fused_ct >= 0 && direct_ct) {
                // the logits of layer `fused_ct` ride on THIS launch: it leaves the rows the job reads untouched (no slabs, no pre-add on visual rows;
                // the text token of a pre-fusion layer comes from the text branch's snapshot), so no copy of the layer's output is kept
                if (!skip) L.cur = PART_V2;
                p.ct_x = w.X; p.ct_self = 1; p.ct_sub_vis = modal_folded ? m->modal : nullptr; p.ct_sub_txt = (modal_folded && fused_ct >= m->nf) ? m->modal + D : nullptr;
                p.ct_nz = nz; p.ct_nv = nv; p.ct_nx = nx; p.ct_T = T; p.ct_skip_text = skip;
                p.ct_slot = fused_slot; p.ct_ncont = m->cfg.n_cont;
                p.ct_flag = in->d_flag; p.ct_logit_scale = m->logit_scale_bb; p.ct_logits = out->d_logits;
                if (fused_ct < m->nf && !skip) p.ct_txt = w.TxtSnap + (size_t)fused_ct * B * T * D;
                fused_ct = -1;
            } else if (fused_ct >= 0) p.x_snap = w.XSnap;   // this fold completes layer `fused_ct`: keep its output for the logits
            direct_ct = false;
            modal_folded = false;
            run_ln(s, p, (double)M * D * 6, false);
        }
        {
            GemmParams p;
            p.A = w.Xn; p.lda = D; p.W = vw.wqkv; p.Wp = vw.pqkv; p.ldw = D; p.bias = vw.bqkv; p.M = M; p.N = 3 * D; p.K = D;
            p.epi = 2; p.rpb = N; p.q = w.Q; p.k = w.K; p.vt = w.Vt; p.H = H; p.Npad = npad; p.D = D; p.q_scale = UVL_QSCALE;
            // every GEMM launch of the small-tile kernels also requests the NEXT GEMM's weight (common.h::prefetch_issue)
            if (pfw) { p.pf = vw.wproj; p.pf_bytes = (uint32_t)((size_t)D * D * 2); }
            run_gemm(s, p, "gemm.qkv", false);
        }
        {
            AttnParams p;
            p.q = w.Q; p.k = w.K; p.vt = w.Vt; p.key_add = w.key_add; p.key_add_stride = npad; p.o = w.O; p.B = B; p.H = H; p.N = N; p.Npad = npad; p.q_prescaled = 1;
            run_attn(s, p, "attention", 4.0 * N * (double)N * D * B, 8.0 * M * D, false);
        }
        residual_gemm(s, "gemm.proj", w.O, D, vw.wproj, vw.bproj, M, D, N, 0, w.Part, pend_v, true, false, vw.pproj, nullptr, 0,
                      pfw ? vw.wfc1 : nullptr, (size_t)Fn * D * 2);
        {
            LnParams p;
            p.x = w.X; p.M = M; p.D = D; p.rpb = N; p.xbs = nj; p.xro = 0;
            consume(p, pend_v);
            p.gamma = vw.ln2g; p.beta = vw.ln2b; p.eps = 1e-6f; p.y_bf16 = w.Xn; p.y_wt = (m->bf16_store & 4) ? 1 : 0;
            if (fused_ct >= 0) {                 // logits of layer `fused_ct` ride on this launch (rows come from the snapshot)
                if (!skip) L.cur = PART_V2;
                p.ct_x = w.XSnap; p.ct_nz = nz; p.ct_nv = nv; p.ct_nx = nx; p.ct_T = T; p.ct_skip_text = skip;
                p.ct_slot = fused_slot; p.ct_ncont = m->cfg.n_cont;
                p.ct_flag = in->d_flag; p.ct_logit_scale = m->logit_scale_bb; p.ct_logits = out->d_logits;
                if (fused_ct < m->nf && !skip) {   // pre-fusion layer: its text token comes from the text branch's snapshot
                    if (fork && hipStreamWaitEvent(s, m->ev_bert[fused_ct], 0) != hipSuccess) return fail(UVL_EHIP, "bert event failed");
                    p.ct_txt = w.TxtSnap + (size_t)fused_ct * B * T * D;
                }
                fused_ct = -1;
            }
            run_ln(s, p, (double)M * D * 6, false);
        }
        {
            GemmParams p;
            p.A = w.Xn; p.lda = D; p.W = vw.wfc1; p.Wp = vw.pfc1; p.ldw = D; p.bias = vw.bfc1; p.M = M; p.N = Fn; p.K = D;
            p.epi = 0; p.C = w.Hb; p.ldc = Fn; p.act = 1;
            if (pfw) { p.pf = vw.wfc2; p.pf_bytes = (uint32_t)((size_t)Fn * D * 2); }
            run_gemm(s, p, "gemm.fc1", false);
        }
        {
            // the NEXT layer is a fusion layer: its "+ modal_embed" (a permanent change of the residual stream) is one more term of this epilogue
            // instead of a read-modify-write of every row in that layer's LayerNorm-1 (in-place residual form only: many-sequence frames)
            // (not where this layer's logits are computed by the stand-alone contrast kernel, which reads the residual stream as it is)
            const bool ct_alone = is_cont_layer(i) && out->d_logits && (m->cfg.txt_token_mean || !m->fuse_contrast);
            const bool next_joint = !last && i + 1 < m->depth && i + 1 >= m->nf && m->fold_modal && !ct_alone;
            // (the last layer's fc2 requests the first conv layer of the head towers)
            const void* nextw = !pfw ? nullptr : (i + 1 < m->depth && !last) ? (const void*)m->vit[i + 1].wqkv : (const void*)m->conv[0].w;
            const size_t nextb = (i + 1 < m->depth && !last) ? (size_t)3 * D * D * 2 : (size_t)4 * m->conv[0].cout * 9 * m->conv[0].cin * 2;
            modal_folded = residual_gemm(s, "gemm.fc2", w.Hb, Fn, vw.wfc2, vw.bfc2, M, Fn, N, 0, w.Part, pend_v, !last, false, vw.pfc2,
                                         next_joint ? m->modal : nullptr, nv, nextw, nextb);
        }
        if (i <= last_bert && !paired) { text_layer(i); if (text_err) return text_err; }
        // ---- contrastive logits (extractor.py:64-65,85-93) ----
        if (is_cont_layer(i)) {
            if (out->d_logits && last && joint && i == m->depth - 1 && m->fuse_contrast && m->debug_stop_layer < 0) {
                head_ct_slot = cont_slot;        // last layer: X is final, head_prep walks the same search rows anyway
            } else if (out->d_logits && !last && !m->cfg.txt_token_mean && m->fuse_contrast) {
                // 'cls' text token and a following layer: the next LayerNorm-1 leaves this layer's output in XSnap and the
                // next LayerNorm-2 computes the logits from it (no launch of its own)
                fused_ct = i;
                fused_slot = cont_slot;
                // direct form: the next LayerNorm-1 computes them itself from its own input rows -- possible where that launch does not modify the
                // rows (the modal embedding already added by fc2 above, no slabs pending) and the text token is there when it starts (fusion
                // layers: row nv of x; the last pre-fusion layer: the text branch's snapshot, joined before the first fusion layer)
                direct_ct = modal_folded && pend_v.nsplit == 0 && i + 1 >= m->nf;
            } else if (out->d_logits) {
                // a stand-alone contrast kernel of a pre-fusion layer reads that BERT layer's snapshot: its last LayerNorm may
                // still be waiting for a partner
                if (paired && !joint) flush_riders(s, i);
                ContrastParams p;
                if (!skip) L.cur = PART_V2;    // first consumer of text data on the visual side: everything from here is part V2
                if (!joint && !skip) {   // text token of THIS layer comes from the text branch's snapshot
                    if (fork && hipStreamWaitEvent(s, m->ev_bert[i], 0) != hipSuccess) return fail(UVL_EHIP, "bert event failed");
                    p.txt_snap = w.TxtSnap + (size_t)i * B * T * D;
                }
                p.x = w.X; p.nj = nj; p.nz = nz; p.nx = nx; p.nv = nv; p.D = D; p.T = T; p.B = B;
                p.text_mask = in->d_text_mask; p.flag = in->d_flag; p.logit_scale = m->logit_scale_bb;
                p.mean_mode = m->cfg.txt_token_mean; p.skip_text = skip; p.logits = out->d_logits; p.slot = cont_slot; p.n_cont = m->cfg.n_cont;
                p.part = pend_v.part; p.nsplit = pend_v.nsplit; p.part_rows = pend_v.rows; p.part_stride = pend_v.stride;   // read-only view
                L.run(s, "contrast", 0, 0, tramp<ContrastParams, launch_contrast>, &p);
            }
            ++cont_slot;
        }
        if (m->debug_stop_layer == i) {
            if (!joint && fork && hipStreamWaitEvent(s, m->ev_join, 0) != hipSuccess) return fail(UVL_EHIP, "join failed");
            break;
        }
    }
    if (m->nf >= m->depth && fork && m->debug_stop_layer < 0) {   // no fusion layer at all: still join the text branch
        if (hipStreamWaitEvent(s, m->ev_join, 0) != hipSuccess) return fail(UVL_EHIP, "join failed");
    }
    flush_ln(s);
    if (paired) flush_riders(s);
    if (pend_v.nsplit || pend_t.nsplit) return fail(UVL_ESTATE, "internal: split-K slabs left unconsumed");
    if (fused_ct >= 0) return fail(UVL_ESTATE, "internal: fused contrast job left unlaunched");

    if (!skip) L.cur = PART_V2;
    // ---- head (modality_adaptive_box_head.py:62-94) ----
    const int S = m->S, C = m->C;
    float* cont = out->d_cont_score ? out->d_cont_score : w.cont;
    float* bbox = out->d_bbox_map ? out->d_bbox_map : w.bbox;
    const int g0_ld = m->cfg.cls_tokenize ? 2 * D : D;
    {
        HeadPrepParams p;
        p.x = w.X; p.nj = nj; p.nv = nv; p.nz = nz; p.nx = nx; p.T = T; p.D = D; p.B = B;
        p.prompt = in->d_prompt; p.logit_scale = m->logit_scale_head; p.text_mask = in->d_text_mask; p.flag = in->d_flag;
        p.softmax_one = m->cfg.softmax_one; p.mean_mode = m->cfg.txt_token_mean; p.cls_tokenize = m->cfg.cls_tokenize; p.skip_text = skip;
        p.g0 = w.G0; p.g0_ld = g0_ld;
        p.o_search = out->d_search; p.o_template = out->d_template; p.o_text = out->d_text; p.o_vis = out->d_vis_token; p.o_txt = out->d_txt_token;
        p.o_cont = cont;
        if (head_ct_slot >= 0) { p.ct_logits = out->d_logits; p.ct_logit_scale = m->logit_scale_bb; p.ct_slot = head_ct_slot; p.ct_ncont = m->cfg.n_cont; }
        if (tb) { p.prompt = nullptr; p.o_cont = nullptr; }      // first pass: token outputs + head input only
        L.run(s, "head_prep", 0, 0, tramp<HeadPrepParams, launch_head_prep>, &p);
        if (tb && !L.err) {
            const int rc = run_prompter(m, w, B, out->d_template, out->d_search, out->d_vis_token, out->d_txt_token, in->d_flag,
                                        tb->template_mask, tb->context_mask, B / 2, tb->prompts_out, s);
            if (rc) return rc;
            HeadPrepParams c = p;
            c.cont_only = 1; c.train_cont = 1; c.prompt = tb->prompts_out; c.o_cont = cont; c.ct_logits = nullptr;
            L.run(s, "head_prep", 0, 0, tramp<HeadPrepParams, launch_head_prep>, &c);
        }
    }
    const bf16_t* cin[4] = {w.G0, w.G1, w.G2, w.G3};
    bf16_t* cout[4] = {w.G1, w.G2, w.G3, w.G4};
    const int in_ld[4] = {g0_ld, 4 * C, 2 * C, C};
    HeadFinParams hf;
    {
        HeadTailParams& p = hf.t;
        p.g4 = w.G4; p.ld = C / 2; p.c8 = C / 8; p.w1 = m->w1; p.b1 = m->b1; p.cont = cont; p.cont_ch = (m->cfg.softmax_one && !tb) ? 3 : 2;
        p.flag = in->d_flag; p.coord = m->coord; p.B = B; p.S = S; p.F = m->F; p.offset_sigmoid = m->cfg.offset_sigmoid; p.joint_cls = m->cfg.joint_cls;
        p.o_cls = out->d_cls_score; p.o_cls_test = out->d_cls_score_test; p.o_bbox_map = bbox; p.o_pred = out->d_pred_boxes; p.o_argmax = out->d_argmax;
        hf.g3 = w.G3; hf.g3_ld = C; hf.wf = m->head_wf; hf.bias3 = m->conv[3].b;
    }
    // the last tower layer + the tail as ONE launch where the geometry fits (head_fin.hip): any batch -- one workgroup per sample
    const bool fin_head = m->head_fin && head_fin_ok(hf) && B <= kDispatch.head_fin_max_batch;
    for (int l = 0; l < (fin_head ? 3 : 4); ++l) {
        const ConvLayerW& cw = m->conv[l];
        int goff[4];
        for (int g = 0; g < 4; ++g) goff[g] = (l == 0) ? ((g == 0 && m->cfg.cls_tokenize) ? D : 0) : g * cw.cin;
        const ConvLayerW* nx = (pfw && l + 1 < 4) ? &m->conv[l + 1] : nullptr;
        const void* nxw = nx ? (const void*)((fin_head && l == 2) ? m->head_wf : nx->w) : nullptr;
        run_conv_layer(L, s, l, cin[l], in_ld[l], goff, cw.w, cw.b, B, m->F, cw.cin, cw.cout, cout[l], w.ConvPart, &m->tune,
                       nxw, nx ? (size_t)4 * nx->cout * 9 * nx->cin * 2 : 0);
    }
    if (fin_head) {
        const double fl = 2.0 * B * S * 4.0 * (C / 8) * 9.0 * (C / 4);
        L.wb_next = 2.0 * 4.0 * (C / 8) * 9.0 * (C / 4);
        L.run(s, "head_fin", fl, 2.0 * ((double)B * S * C + 4.0 * (C / 8) * 9.0 * (C / 4)), tramp<HeadFinParams, launch_head_fin>, &hf);
    } else {
        L.run(s, "head_tail", 0, 0, tramp<HeadTailParams, launch_head_tail>, &hf.t);
    }
    return L.err;
}

extern "C" int uvl_forward_test(uvl_model_t* m, const uvl_inputs* in, const uvl_outputs* out, void* d_ws, size_t ws_bytes, void* stream) {
    return run_forward(m, in, out, d_ws, ws_bytes, (hipStream_t)stream, nullptr);
}

extern "C" int uvl_forward_test_profiled(uvl_model_t* m, const uvl_inputs* in, const uvl_outputs* out, void* d_ws, size_t ws_bytes, void* stream,
                                         float ms_per_family[UVL_NFAM], int launches_per_family[UVL_NFAM]) {
    Profiler prof;
    hipStream_t s = (hipStream_t)stream;
    int rc = run_forward(m, in, out, d_ws, ws_bytes, s, &prof);
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = fail(UVL_EHIP, "sync failed");
    std::map<std::string, ProfEntry> agg;
    if (ms_per_family) for (int i = 0; i < UVL_NFAM; ++i) { ms_per_family[i] = 0; if (launches_per_family) launches_per_family[i] = 0; }
    for (auto& r : prof.recs) {
        float ms = 0;
        hipEventElapsedTime(&ms, r.a, r.b);
        ProfEntry& e = agg[std::string(r.name) + "|" + r.kernel];      // one entry per (launch site, kernel instantiation)
        e.name = r.name; e.kernel = r.kernel; e.ms += ms; e.flops += r.flops; e.bytes += r.bytes; e.wbytes += r.wbytes; e.launches += 1;
        int fam = 4;
        if (!strncmp(r.name, "gemm", 4)) fam = 0;
        else if (!strncmp(r.name, "attention", 9)) fam = 1;
        else if (!strncmp(r.name, "layernorm", 9)) fam = 2;
        else if (!strncmp(r.name, "conv", 4)) fam = 3;
        if (ms_per_family) ms_per_family[fam] += ms;
        if (launches_per_family) launches_per_family[fam] += 1;
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    if (m) { m->prof.clear(); for (auto& kv : agg) m->prof.push_back(kv.second); }
    return rc;
}

extern "C" int uvl_profile_count(const uvl_model_t* m) { return m ? (int)m->prof.size() : 0; }
extern "C" int uvl_profile_entry(const uvl_model_t* m, int i, char* name, char* kernel, int name_cap, double* ms, double* flops, double* bytes, int* launches) {
    if (!m || i < 0 || i >= (int)m->prof.size()) return fail(UVL_EINVAL, "bad profile index");
    const ProfEntry& e = m->prof[i];
    if (name && name_cap > 0) { strncpy(name, e.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (kernel && name_cap > 0) { strncpy(kernel, e.kernel.c_str(), name_cap - 1); kernel[name_cap - 1] = 0; }
    if (ms) *ms = e.ms; if (flops) *flops = e.flops; if (bytes) *bytes = e.bytes; if (launches) *launches = e.launches;
    return UVL_OK;
}

extern "C" int uvl_profile_entry_weight_bytes(const uvl_model_t* m, int i, double* bytes) {
    if (!m || i < 0 || i >= (int)m->prof.size() || !bytes) return fail(UVL_EINVAL, "bad profile index");
    *bytes = m->prof[i].wbytes;
    return UVL_OK;
}

extern "C" int uvl_tune_set(uvl_model_t* m, const char* key, int value) {
    if (!m || !key) return fail(UVL_EINVAL, "uvl_tune_set: null argument");
    static const struct { const char* key; int32_t uvl_tuning::*field; } keys[] = {
        {"gemm_cfg", &uvl_tuning::gemm_cfg}, {"gemm_gm", &uvl_tuning::gemm_gm}, {"gemm_prod", &uvl_tuning::gemm_prod},
        {"gemm_big", &uvl_tuning::gemm_big}, {"gemm_kxcd", &uvl_tuning::gemm_kxcd}, {"attn_cfg", &uvl_tuning::attn_cfg},
        {"sk_k1", &uvl_tuning::sk_k1}, {"sk_k4", &uvl_tuning::sk_k4}, {"gemm_pipe", &uvl_tuning::gemm_pipe}, {"ring1", &uvl_tuning::ring1}, {"text_cfg", &uvl_tuning::text_cfg}, {"res_store", &uvl_tuning::res_store}, {"slab_store", &uvl_tuning::slab_store}, {"attn_wgs", &uvl_tuning::attn_wgs}, {"gemm_dr", &uvl_tuning::gemm_dr}, {"res_pre", &uvl_tuning::res_pre}, {"fin_w", &uvl_tuning::fin_w}, {"lnf_w", &uvl_tuning::lnf_w}};
    for (const auto& k : keys)
        if (!strcmp(key, k.key)) {
            // (cfg 36 on proj / fc2 too needs their weight images: made by the next frame on ITS stream -- run_forward -- not here on the null stream, where the
            // allocation + pack + synchronize would stall every blocking stream or break a capture in progress)
            m->tune.*(k.field) = value < 0 ? -1 : value;
            return UVL_OK;
        }
    if (!strcmp(key, "reset")) { uvl_tuning_init(&m->tune); return UVL_OK; }
    return fail(UVL_ENOTFOUND, "unknown tuning key '%s'", key);
}

extern "C" int uvl_debug_set(uvl_model_t* m, const char* key, int value) {
    if (!m || !key) return fail(UVL_EINVAL, "null argument");
    if (!strcmp(key, "stop_layer")) { m->debug_stop_layer = value; return UVL_OK; }
    if (!strcmp(key, "prefetch_w")) { m->prefetch_w = value < 0 ? 0 : (value > 2 ? 2 : value); return UVL_OK; }
    if (!strcmp(key, "fold_modal")) { m->fold_modal = value ? 1 : 0; return UVL_OK; }
    if (!strcmp(key, "rider_first")) { m->rider_first = value ? 1 : 0; return UVL_OK; }
    if (!strcmp(key, "text_nt")) { m->text_nt = value & 15; return UVL_OK; }
    if (!strcmp(key, "bf16_store")) { m->bf16_store = value; return UVL_OK; }
    if (!strcmp(key, "rider_sk")) { m->rider_sk = value > 1 ? 2 : 1; return UVL_OK; }
    if (!strcmp(key, "pair_text")) { m->pair_text = value < 0 ? 0 : (value > 3 ? 3 : value); return UVL_OK; }
    if (!strcmp(key, "fuse_contrast")) { m->fuse_contrast = value ? 1 : 0; return UVL_OK; }
    if (!strcmp(key, "fuse_ln")) { m->fuse_ln = value ? 1 : 0; return UVL_OK; }
    if (!strcmp(key, "fold_ln")) { m->fold_ln = value ? 1 : 0; return UVL_OK; }
    if (!strcmp(key, "head_fin")) { m->head_fin = value ? 1 : 0; return UVL_OK; }
    if (!strcmp(key, "rider_pf")) { m->rider_pf = value < 0 ? 0 : value; return UVL_OK; }
    if (!strcmp(key, "text_dr_res")) { m->text_dr_res = value ? 1 : 0; return UVL_OK; }      // (its weight images: the next frame, on its stream)
    if (!strcmp(key, "fork_text")) { m->fork_text = value ? 1 : 0; return UVL_OK; }   // 0: the text branch of multi-sequence frames runs on the caller's stream
    return fail(UVL_ENOTFOUND, "unknown debug key '%s'", key);
}

// ---- hipGraph --------------------------------------------------------------------------------------
// A multi-stream graph replays badly on ROCm 7.2 (the executor enqueues one branch after the other, so the second
// branch starts hundreds of microseconds late); single-stream graphs replay at GPU speed.  The frame is therefore
// recorded as up to three single-stream graphs: T (text branch) replayed on the library's second stream, V1 (visual
// stream up to the first consumer of text data) and V2 (the rest) replayed on the caller's stream, ordered by events.
static void graph_free(uvl_model* m) {
    for (int k = 0; k < 3; ++k) {
        if (m->graph_exec3[k]) { hipGraphExecDestroy(m->graph_exec3[k]); m->graph_exec3[k] = nullptr; }
        if (m->graph3[k]) { hipGraphDestroy(m->graph3[k]); m->graph3[k] = nullptr; }
    }
}

extern "C" int uvl_graph_release(uvl_model_t* m) {
    if (!m) return fail(UVL_EINVAL, "null model");
    graph_free(m);
    return UVL_OK;
}

extern "C" int uvl_graph_capture(uvl_model_t* m, const uvl_inputs* in, const uvl_outputs* out, void* d_ws, size_t ws_bytes) {
    if (!m || !in) return fail(UVL_EINVAL, "null argument");
    graph_free(m);
    if (!m->cap_stream) HIPCHK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    HIPCHK(hipDeviceSynchronize());
    // one sequence with paired text kernels is a single-stream frame: one graph, like the no-text case
    const bool text = !in->skip_text && !in->reuse_text && m->nf >= 0 && !text_rides(m, in->batch, 0, 0);
    const int part_of[3] = {PART_TEXT, PART_V1, PART_V2};
    for (int k = 0; k < 3; ++k) {
        if (!text && k != 1) continue;
        const int parts = text ? part_of[k] : PART_ALL;       // no text branch: one graph holds the whole frame
        HIPCHK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
        int rc = run_forward(m, in, out, d_ws, ws_bytes, m->cap_stream, nullptr, parts);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(m->cap_stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); graph_free(m); return rc; }
        if (e != hipSuccess) { graph_free(m); return fail(UVL_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
        m->graph3[k] = g;
        HIPCHK(hipGraphInstantiate(&m->graph_exec3[k], g, nullptr, nullptr, 0));
    }
    m->graph_has_text = text;
    return UVL_OK;
}

extern "C" int uvl_graph_launch(uvl_model_t* m, void* stream) {
    if (!m || !m->graph_exec3[1]) return fail(UVL_ESTATE, "no captured graph");
    hipStream_t s = (hipStream_t)stream;
    if (!m->graph_has_text) {
        HIPCHK(hipGraphLaunch(m->graph_exec3[1], s));
        return UVL_OK;
    }
    // the text graph may start once everything queued on the caller's stream so far (previous frame, input copies) is done
    HIPCHK(hipEventRecord(m->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(m->aux, m->ev_fork, 0));
    HIPCHK(hipGraphLaunch(m->graph_exec3[0], m->aux));
    HIPCHK(hipEventRecord(m->ev_join, m->aux));
    HIPCHK(hipGraphLaunch(m->graph_exec3[1], s));
    HIPCHK(hipStreamWaitEvent(s, m->ev_join, 0));
    HIPCHK(hipGraphLaunch(m->graph_exec3[2], s));
    return UVL_OK;
}

// ---- prompter (UVLTrack.forward_prompt, uvltrack.py:33-38 -> head:96-106 -> heads/utils.py:82-99) ---------------
extern "C" int uvl_forward_prompt(uvl_model_t* m, int batch, const float* d_template_tokens, const float* d_search_tokens,
                                  const float* d_vis_token, const float* d_txt_token, const int64_t* d_flag,
                                  const uint8_t* d_template_mask, const uint8_t* d_context_mask, float* d_prompt_out,
                                  void* d_ws, size_t ws_bytes, void* stream) {
    if (!m || !d_template_tokens || !d_search_tokens || !d_vis_token || !d_txt_token || !d_flag || !d_template_mask || !d_context_mask || !d_prompt_out)
        return fail(UVL_EINVAL, "uvl_forward_prompt: null argument");
    if (!m->finalized) return fail(UVL_ESTATE, "uvl_finalize_weights has not been called");
    if (!m->has_prompter) return fail(UVL_ESTATE, "prompter weights (box_head.prompter.{logit_scale,query_embed,mlp.*}) were not loaded");
    const int B = batch;
    if (B <= 0 || B > m->cfg.max_batch) return fail(UVL_EINVAL, "batch %d outside [1, %d]", B, m->cfg.max_batch);
    const Workspace w = carve(m, B, (char*)d_ws);
    if (!d_ws || ws_bytes < w.total || (uintptr_t)d_ws % 256) return fail(UVL_EINVAL, "bad workspace");
    return run_prompter(m, w, B, d_template_tokens, d_search_tokens, d_vis_token, d_txt_token, d_flag, d_template_mask, d_context_mask, 0,
                        d_prompt_out, (hipStream_t)stream);
}

extern "C" int uvl_anno2mask(const float* d_boxes_xywh, int batch, int size, uint8_t* d_mask, void* stream) {
    if (!d_boxes_xywh || !d_mask || batch <= 0 || size <= 0) return fail(UVL_EINVAL, "uvl_anno2mask: bad argument");
    HIPCHK(launch_anno2mask(d_boxes_xywh, batch, size, d_mask, (hipStream_t)stream));
    return UVL_OK;
}

// ---- UVLTrack.forward (uvltrack.py:18-24), eval mode: what the tracker's grounding() calls (tracker:45-62) -------------------------
extern "C" int uvl_forward(uvl_model_t* m, const uvl_inputs* in, const uint8_t* d_template_mask, const uint8_t* d_context_mask,
                           const uvl_outputs* out, float* d_prompts_out, void* d_ws, size_t ws_bytes, void* stream) {
    if (!m || !in || !out || !d_template_mask || !d_context_mask || !d_prompts_out) return fail(UVL_EINVAL, "uvl_forward: null argument");
    if (!m->finalized) return fail(UVL_ESTATE, "uvl_finalize_weights has not been called");
    if (!m->has_prompter) return fail(UVL_ESTATE, "prompter weights (box_head.prompter.{logit_scale,query_embed,mlp.*}) were not loaded");
    if (!out->d_search || !out->d_template || !out->d_vis_token || !out->d_txt_token)
        return fail(UVL_EINVAL, "uvl_forward: outputs search / template / vis_token / txt_token are required (the inline prompter reads them)");
    if (in->skip_text) return fail(UVL_EINVAL, "uvl_forward: skip_text is a forward_test option");
    TrainBranch tb{d_template_mask, d_context_mask, d_prompts_out};
    return run_forward(m, in, out, d_ws, ws_bytes, (hipStream_t)stream, nullptr, PART_ALL, &tb);
}

// ---- tracker decode (lib/test/tracker/uvltrack.py:116-125) ------------------------------------------------------
extern "C" int uvl_decode(uvl_model_t* m, int batch, const float* d_cls_score_test, const float* d_cont_score, const float* d_bbox_map,
                          const float* d_window, const float* d_state, const float* d_resize_factor, const float* d_image_hw,
                          float margin, float* d_new_state, float* d_score, float* d_box_net, int64_t* d_index, void* stream) {
    if (!m || !d_cls_score_test || !d_bbox_map || !d_window || !d_state || !d_resize_factor || !d_image_hw || !d_new_state || batch <= 0)
        return fail(UVL_EINVAL, "uvl_decode: bad argument");
    DecodeParams p;
    p.cls = d_cls_score_test; p.cont = d_cont_score; p.bbox_map = d_bbox_map; p.window = d_window; p.state = d_state;
    p.resize_factor = d_resize_factor; p.image_hw = d_image_hw; p.B = batch; p.S = m->S; p.cont_ch = m->cfg.softmax_one ? 3 : 2;
    p.search_size = (float)m->cfg.search_size; p.margin = margin;
    p.new_state = d_new_state; p.score = d_score; p.box_net = d_box_net; p.index = d_index;
    HIPCHK(launch_decode(p, (hipStream_t)stream));
    return UVL_OK;
}

// ---- pre-processing --------------------------------------------------------------------
// Crop geometry of sample_target (processing_utils.py:173-193): plain integer / double arithmetic on the host, Python's
// round() is round-half-to-even = nearbyint in the default rounding mode.
extern "C" int uvl_crop_geometry_of(const float box_xywh[4], float search_area_factor, int output_sz, int height, int width, uvl_crop_geometry* g) {
    if (!box_xywh || !g || height <= 0 || width <= 0) return fail(UVL_EINVAL, "uvl_crop_geometry_of: bad argument");
    const double x = box_xywh[0], y = box_xywh[1], w = box_xywh[2], h = box_xywh[3];
    const double c = std::ceil(std::sqrt(w * h) * (double)search_area_factor);
    if (!(c >= 1.0) || c > 1e8) return fail(UVL_EINVAL, "uvl_crop_geometry_of: too small (or absurd) bounding box");
    g->crop_sz = (int)c;
    g->x1 = (int)std::nearbyint(x + 0.5 * w - c * 0.5);
    g->y1 = (int)std::nearbyint(y + 0.5 * h - c * 0.5);
    const int x2 = g->x1 + g->crop_sz, y2 = g->y1 + g->crop_sz;
    g->x1_pad = std::max(0, -g->x1);
    g->x2_pad = std::max(x2 - width + 1, 0);
    g->y1_pad = std::max(0, -g->y1);
    g->y2_pad = std::max(y2 - height + 1, 0);
    g->resize_factor = output_sz > 0 ? (float)((double)output_sz / c) : 1.0f;
    if (g->x1 + g->x1_pad >= x2 - g->x2_pad || g->y1 + g->y1_pad >= y2 - g->y2_pad)
        return fail(UVL_EINVAL, "uvl_crop_geometry_of: the crop does not intersect the image");
    return UVL_OK;
}

static int sample_target_impl(const uint8_t* d_buf, int ox, int oy, int bw, int bh, int row_stride_bytes, int height, int width,
                              const float box_xywh[4], float search_area_factor, int output_sz, uint8_t* d_patch_hwc, float* d_norm_chw,
                              uint8_t* d_att_mask, uvl_crop_geometry* geometry_out, void* stream) {
    if (!d_buf || output_sz <= 0 || bw <= 0 || bh <= 0 || row_stride_bytes < 3 * bw) return fail(UVL_EINVAL, "uvl_sample_target: bad argument");
    if (!d_patch_hwc && !d_norm_chw && !d_att_mask) return fail(UVL_EINVAL, "uvl_sample_target: no output requested");
    uvl_crop_geometry g;
    const int rc = uvl_crop_geometry_of(box_xywh, search_area_factor, output_sz, height, width, &g);
    if (rc) return rc;
    // the kept part of the crop, [x1+x1_pad, x2-x2_pad) x [y1+y1_pad, y2-y2_pad), must lie inside the buffer
    const int kx0 = g.x1 + g.x1_pad, kx1 = g.x1 + g.crop_sz - g.x2_pad, ky0 = g.y1 + g.y1_pad, ky1 = g.y1 + g.crop_sz - g.y2_pad;
    if (kx0 < ox || ky0 < oy || kx1 > ox + bw || ky1 > oy + bh) return fail(UVL_EINVAL, "uvl_sample_target: the window does not cover the crop");
    PreprocParams p;
    p.img = d_buf; p.H = height; p.W = width; p.stride = row_stride_bytes; p.ox = ox; p.oy = oy;
    p.crop_sz = g.crop_sz; p.x1 = g.x1; p.y1 = g.y1; p.x1_pad = g.x1_pad; p.x2_pad = g.x2_pad; p.y1_pad = g.y1_pad; p.y2_pad = g.y2_pad;
    p.out = output_sz; p.patch = d_patch_hwc; p.norm = d_norm_chw; p.att = d_att_mask;
    HIPCHK(launch_preprocess(p, (hipStream_t)stream));
    if (geometry_out) *geometry_out = g;
    return UVL_OK;
}

extern "C" int uvl_sample_target(const uint8_t* d_image, int height, int width, int row_stride_bytes, const float box_xywh[4],
                                 float search_area_factor, int output_sz, uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask,
                                 uvl_crop_geometry* geometry_out, void* stream) {
    return sample_target_impl(d_image, 0, 0, width, height, row_stride_bytes, height, width, box_xywh, search_area_factor, output_sz,
                              d_patch_hwc, d_norm_chw, d_att_mask, geometry_out, stream);
}

extern "C" int uvl_sample_target_window(const uint8_t* d_window, int win_x0, int win_y0, int win_width, int win_height, int row_stride_bytes,
                                        int frame_height, int frame_width, const float box_xywh[4], float search_area_factor, int output_sz,
                                        uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask, uvl_crop_geometry* geometry_out,
                                        void* stream) {
    return sample_target_impl(d_window, win_x0, win_y0, win_width, win_height, row_stride_bytes, frame_height, frame_width, box_xywh,
                              search_area_factor, output_sz, d_patch_hwc, d_norm_chw, d_att_mask, geometry_out, stream);
}

extern "C" int uvl_sample_target_staged(const uint8_t* h_stage, uint8_t* d_stage, size_t header_bytes, int win_x0, int win_y0, int win_width,
                                        int win_height, int frame_height, int frame_width, const float box_xywh[4], float search_area_factor,
                                        int output_sz, uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask,
                                        uvl_crop_geometry* geometry_out, void* stream) {
    if (!h_stage || !d_stage || header_bytes % 16 != 0 || win_width <= 0 || win_height <= 0)
        return fail(UVL_EINVAL, "uvl_sample_target_staged: bad argument");
    const size_t nbytes = header_bytes + (size_t)win_width * win_height * 3;
    HIPCHK(hipMemcpyAsync(d_stage, h_stage, nbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return sample_target_impl(d_stage + header_bytes, win_x0, win_y0, win_width, win_height, win_width * 3, frame_height, frame_width, box_xywh,
                              search_area_factor, output_sz, d_patch_hwc, d_norm_chw, d_att_mask, geometry_out, stream);
}

// grounding_resize (processing_utils.py:77-104 for the integer geometry; Python int() truncates towards zero)
extern "C" int uvl_grounding_resize(const uint8_t* d_image, int height, int width, int row_stride_bytes, int output_sz,
                                    uint8_t* d_patch_hwc, float* d_norm_chw, uint8_t* d_att_mask, int32_t image_top_coords[4], void* stream) {
    if (!d_image || height <= 0 || width <= 0 || output_sz <= 0 || row_stride_bytes < 3 * width) return fail(UVL_EINVAL, "uvl_grounding_resize: bad argument");
    if (!d_patch_hwc && !d_norm_chw && !d_att_mask) return fail(UVL_EINVAL, "uvl_grounding_resize: no output requested");
    int ow, oh;
    if (width > height) { ow = output_sz; oh = (int)((double)output_sz * height / width); }
    else { oh = output_sz; ow = (int)((double)output_sz * width / height); }
    if (ow <= 0 || oh <= 0) return fail(UVL_EINVAL, "uvl_grounding_resize: degenerate aspect ratio");
    int y1 = (int)((output_sz - oh) / 2.0), y2 = y1, x1 = (int)((output_sz - ow) / 2.0), x2 = x1;
    if (y1 + y2 + oh != output_sz) y1 += 1;
    if (x1 + x2 + ow != output_sz) x1 += 1;
    GroundingParams p;
    p.img = d_image; p.H = height; p.W = width; p.stride = row_stride_bytes; p.new_w = ow; p.new_h = oh; p.x1_pad = x1; p.y1_pad = y1;
    p.out = output_sz; p.patch = d_patch_hwc; p.norm = d_norm_chw; p.att = d_att_mask;
    HIPCHK(launch_grounding_resize(p, (hipStream_t)stream));
    if (image_top_coords) { image_top_coords[0] = x1; image_top_coords[1] = y1; image_top_coords[2] = ow; image_top_coords[3] = oh; }
    return UVL_OK;
}

extern "C" int uvl_normalize_u8(const uint8_t* d_patch_hwc, int height, int width, float* d_norm_chw, void* stream) {
    if (!d_patch_hwc || !d_norm_chw || height <= 0 || width <= 0) return fail(UVL_EINVAL, "uvl_normalize_u8: bad argument");
    HIPCHK(launch_normalize_u8(d_patch_hwc, d_norm_chw, height * width, (hipStream_t)stream));
    return UVL_OK;
}

// ---- per-kernel entry points -----------------------------------------------------------------------
extern "C" int uvl_pack_weight(const void* d_w, void* d_w_packed, int N, int K, void* stream) {
    if (!d_w || !d_w_packed || N <= 0 || N % 16 != 0 || K <= 0 || K % 64 != 0) return fail(UVL_EINVAL, "uvl_pack_weight: need N %% 16 == 0 and K %% 64 == 0");
    HIPCHK(launch_pack_w_dr((const bf16_t*)d_w, (bf16_t*)d_w_packed, N, K, (hipStream_t)stream));
    return UVL_OK;
}
extern "C" int uvl_linear_pk(const void* d_x, const void* d_w, const void* d_w_packed, const float* d_bias, void* d_y, int M, int N, int K, int act, int out_f32,
                             int accumulate, const uvl_tuning* tune, void* stream) {
    if (!d_x || !d_w || !d_y || M <= 0 || N % 32 != 0 || K % 64 != 0) return fail(UVL_EINVAL, "uvl_linear: need N %% 32 == 0 and K %% 64 == 0");
    GemmParams p;
    p.A = (const bf16_t*)d_x; p.lda = K; p.W = (const bf16_t*)d_w; p.ldw = K; p.bias = d_bias; p.M = M; p.N = N; p.K = K;
    p.epi = out_f32 ? 1 : 0; p.C = d_y; p.ldc = N; p.act = act; p.accumulate = accumulate; p.tune = tune;
    p.c_store = tune_get(tune, &uvl_tuning::res_store, 0);       // cache policy / atomic form of the f32 stores (0 plain; the frame's default is its own)
    p.Wp = (const bf16_t*)d_w_packed;
    HIPCHK(launch_gemm(p, (hipStream_t)stream));
    return UVL_OK;
}
extern "C" int uvl_linear(const void* d_x, const void* d_w, const float* d_bias, void* d_y, int M, int N, int K, int act, int out_f32,
                          int accumulate, const uvl_tuning* tune, void* stream) {
    return uvl_linear_pk(d_x, d_w, nullptr, d_bias, d_y, M, N, K, act, out_f32, accumulate, tune, stream);
}

/* tuning entry: y[sk] (f32 slabs [splitk][M,N]) = partial sums over K-range sk (bias in slab 0) */
extern "C" int uvl_linear_splitk(const void* d_x, const void* d_w, const float* d_bias, float* d_slabs, int M, int N, int K, int splitk,
                                 const uvl_tuning* tune, void* stream) {
    if (!d_x || !d_w || !d_slabs || M <= 0) return fail(UVL_EINVAL, "uvl_linear_splitk: bad argument");
    GemmParams p;
    p.A = (const bf16_t*)d_x; p.lda = K; p.W = (const bf16_t*)d_w; p.ldw = K; p.bias = d_bias; p.M = M; p.N = N; p.K = K;
    p.epi = 1; p.C = d_slabs; p.ldc = N; p.splitk = splitk; p.part_stride = (size_t)M * N; p.tune = tune;
    HIPCHK(launch_gemm(p, (hipStream_t)stream));
    return UVL_OK;
}

/* ---- conv towers alone (parity test entry; same kernels the frame uses) ---- */
extern "C" int uvl_fold_conv_bn(const float* d_w, const float* d_b, const float* d_bn_w, const float* d_bn_b, const float* d_bn_mean,
                                const float* d_bn_var, void* d_w_packed, float* d_bias_folded, int cout, int cin, void* stream) {
    if (!d_w || !d_b || !d_bn_w || !d_bn_b || !d_bn_mean || !d_bn_var || !d_w_packed || !d_bias_folded || cout <= 0 || cin <= 0)
        return fail(UVL_EINVAL, "uvl_fold_conv_bn: bad argument");
    HIPCHK(launch_fold_conv_bn(d_w, d_b, d_bn_w, d_bn_b, d_bn_mean, d_bn_var, (bf16_t*)d_w_packed, d_bias_folded, cout, cin, (hipStream_t)stream));
    return UVL_OK;
}

extern "C" int uvl_conv_tower_layer(const void* d_x, int batch, int feat, int x_ld, const int32_t x_group_offset[4], int cin, int cout,
                                    const void* d_w_packed, const float* d_bias_folded, void* d_y, float* d_slabs, const uvl_tuning* tune, void* stream) {
    if (!d_x || !d_w_packed || !d_bias_folded || !d_y || !x_group_offset || batch <= 0 || feat <= 0) return fail(UVL_EINVAL, "uvl_conv_tower_layer: bad argument");
    if (cin % 64 != 0 || cout % 32 != 0) return fail(UVL_EINVAL, "uvl_conv_tower_layer: need cin %% 64 == 0 and cout %% 32 == 0");
    Launcher L{nullptr};
    int goff[4] = {x_group_offset[0], x_group_offset[1], x_group_offset[2], x_group_offset[3]};
    run_conv_layer(L, (hipStream_t)stream, 0, (const bf16_t*)d_x, x_ld, goff, (const bf16_t*)d_w_packed, d_bias_folded, batch, feat, cin, cout,
                   (bf16_t*)d_y, d_slabs, tune);
    return L.err;
}

extern "C" int uvl_contrast_logits(const float* d_x, int batch, int rows_per_sample, int dim, int nz, int nx, int text_row, int text_len, const uint8_t* d_text_mask,
                                   int mean_mode, int skip_text, const int64_t* d_flag, const float* d_logit_scale, float* d_logits, int slot, int n_cont, int form, void* stream) {
    if (!d_x || !d_flag || !d_logit_scale || !d_logits || batch <= 0 || nx <= 0 || nz < 0 || dim % 4 != 0 || dim > 1024 || slot < 0 || slot >= n_cont)
        return fail(UVL_EINVAL, "uvl_contrast_logits: bad argument");
    if (1 + nz + nx > rows_per_sample || (!skip_text && text_row + (mean_mode ? text_len : 1) > rows_per_sample)) return fail(UVL_EINVAL, "uvl_contrast_logits: rows out of range");
    if (form == 1) {
        if (mean_mode) return fail(UVL_EINVAL, "uvl_contrast_logits: the LayerNorm-free frame's job takes the 'cls' text token only");
        CtJob j;
        j.x = d_x; j.xbs = rows_per_sample; j.D = dim; j.B = batch; j.nz = nz; j.nv = text_row; j.nx = nx; j.skip_text = skip_text ? 1 : 0;
        j.flag = d_flag; j.logit_scale = d_logit_scale; j.logits = d_logits; j.slot = slot; j.ncont = n_cont;
        HIPCHK(launch_ct_job(j, (hipStream_t)stream));
        return UVL_OK;
    }
    if (mean_mode && !d_text_mask) return fail(UVL_EINVAL, "uvl_contrast_logits: the 'mean' text token needs the text mask");
    ContrastParams p;
    p.x = d_x; p.nj = rows_per_sample; p.nz = nz; p.nx = nx; p.nv = text_row; p.D = dim; p.T = text_len; p.B = batch;
    p.text_mask = d_text_mask; p.flag = d_flag; p.logit_scale = d_logit_scale; p.mean_mode = mean_mode ? 1 : 0; p.skip_text = skip_text ? 1 : 0;
    p.logits = d_logits; p.slot = slot; p.n_cont = n_cont;
    HIPCHK(launch_contrast(p, (hipStream_t)stream));
    return UVL_OK;
}

extern "C" int uvl_head_end(const void* d_g3, int batch, int feat, int cin, const void* d_w_packed, const float* d_bias_folded, const float* d_w1, const float* d_b1,
                            const float* d_cont_score, int cont_channels, const int64_t* d_flag, const float* d_coord, int offset_sigmoid, int joint_cls, int form, void* d_scratch,
                            float* d_cls_score, float* d_cls_score_test, float* d_bbox_map, float* d_pred_boxes, int64_t* d_argmax, void* stream) {
    if (!d_g3 || !d_w_packed || !d_bias_folded || !d_w1 || !d_b1 || !d_cont_score || !d_flag || !d_coord || !d_scratch || batch <= 0 || feat <= 0)
        return fail(UVL_EINVAL, "uvl_head_end: bad argument");
    if (cin % 64 != 0 || cont_channels < 1 || cont_channels > 3) return fail(UVL_EINVAL, "uvl_head_end: need cin %% 64 == 0 and 1..3 cont_score channels");
    hipStream_t s = (hipStream_t)stream;
    const int cout = cin / 2;
    HeadFinParams hf;
    HeadTailParams& p = hf.t;
    p.g4 = (const bf16_t*)d_scratch; p.ld = 4 * cout; p.c8 = cout; p.w1 = d_w1; p.b1 = d_b1; p.cont = d_cont_score; p.cont_ch = cont_channels;
    p.flag = d_flag; p.coord = d_coord; p.B = batch; p.S = feat * feat; p.F = feat; p.offset_sigmoid = offset_sigmoid ? 1 : 0; p.joint_cls = joint_cls ? 1 : 0;
    p.o_cls = d_cls_score; p.o_cls_test = d_cls_score_test; p.o_bbox_map = d_bbox_map; p.o_pred = d_pred_boxes; p.o_argmax = d_argmax;
    hf.g3 = (const bf16_t*)d_g3; hf.g3_ld = 4 * cin; hf.wf = (const bf16_t*)d_scratch; hf.bias3 = d_bias_folded;
    if (form == 1) {
        if (!head_fin_ok(hf)) return fail(UVL_EINVAL, "uvl_head_end: the one-launch form is written for 16 x 16 features and 4 x 64 -> 4 x 32 channels");
        HIPCHK(launch_head_fin_pack((const bf16_t*)d_w_packed, (bf16_t*)d_scratch, s));
        HIPCHK(launch_head_fin(hf, s));
        return UVL_OK;
    }
    Launcher L{nullptr};
    int goff[4] = {0, cin, 2 * cin, 3 * cin};
    uvl_tuning tune;
    uvl_tuning_init(&tune);
    tune.fin_w = 2;                                          // (the plain launch: this entry's form 0 is the two-launch path of every geometry)
    run_conv_layer(L, s, 3, (const bf16_t*)d_g3, 4 * cin, goff, (const bf16_t*)d_w_packed, d_bias_folded, batch, feat, cin, cout, (bf16_t*)d_scratch, nullptr, &tune);
    if (L.err) return L.err;
    HIPCHK(launch_head_tail(p, s));
    return UVL_OK;
}

extern "C" int uvl_attention(const void* d_q, const void* d_k, const void* d_vt, const float* d_key_add, void* d_o, int B, int H, int N, int Npad, int q_prescaled,
                             const uvl_tuning* tune, void* stream) {
    if (!d_q || !d_k || !d_vt || !d_key_add || !d_o) return fail(UVL_EINVAL, "uvl_attention: null pointer");
    AttnParams p;
    p.q = (const bf16_t*)d_q; p.k = (const bf16_t*)d_k; p.vt = (const bf16_t*)d_vt; p.key_add = d_key_add; p.key_add_stride = Npad;
    p.o = (bf16_t*)d_o; p.B = B; p.H = H; p.N = N; p.Npad = Npad; p.q_prescaled = q_prescaled ? 1 : 0; p.tune = tune;
    HIPCHK(launch_attention(p, (hipStream_t)stream));
    return UVL_OK;
}

extern "C" int uvl_qkv_project_pk(const void* d_x, const void* d_w, const void* d_w_packed, const float* d_bias, void* d_q, void* d_k, void* d_vt, int B, int N, int Npad, int D, float q_scale,
                                  const uvl_tuning* tune, void* stream) {
    if (!d_x || !d_w || !d_q || !d_k || !d_vt || D % 64 != 0) return fail(UVL_EINVAL, "uvl_qkv_project: bad argument");
    GemmParams p;
    p.A = (const bf16_t*)d_x; p.lda = D; p.W = (const bf16_t*)d_w; p.ldw = D; p.bias = d_bias; p.M = B * N; p.N = 3 * D; p.K = D;
    p.epi = 2; p.rpb = N; p.q = (bf16_t*)d_q; p.k = (bf16_t*)d_k; p.vt = (bf16_t*)d_vt; p.H = D / 64; p.Npad = Npad; p.D = D; p.q_scale = q_scale; p.tune = tune;
    p.Wp = (const bf16_t*)d_w_packed;
    HIPCHK(launch_gemm(p, (hipStream_t)stream));
    return UVL_OK;
}
extern "C" int uvl_qkv_project(const void* d_x, const void* d_w, const float* d_bias, void* d_q, void* d_k, void* d_vt, int B, int N, int Npad, int D, float q_scale,
                               const uvl_tuning* tune, void* stream) {
    return uvl_qkv_project_pk(d_x, d_w, nullptr, d_bias, d_q, d_k, d_vt, B, N, Npad, D, q_scale, tune, stream);
}

/* ---- LayerNorm-free forms (round 6; the kernels of one- / two-sequence frames, fold.h / gemm_fin.hip) ---- */
extern "C" int uvl_fold_ln_linear(const float* d_w, const float* d_bias, const float* d_gamma, const float* d_beta, void* d_w_folded, float* d_bias_folded, float* d_colsum,
                                  int N, int K, void* stream) {
    if (!d_w || !d_gamma || !d_beta || !d_w_folded || !d_bias_folded || !d_colsum || N <= 0 || K <= 0 || K % 4 != 0) return fail(UVL_EINVAL, "uvl_fold_ln_linear: bad argument");
    HIPCHK(launch_fold_ln_linear(d_w, d_bias, d_gamma, d_beta, (bf16_t*)d_w_folded, d_bias_folded, d_colsum, N, K, (hipStream_t)stream));
    return UVL_OK;
}
extern "C" int uvl_linear_fin(const void* d_a, const void* d_w, const float* d_bias, float* d_x, void* d_xn, float* d_stats, int M, int N, int K, int accumulate,
                              const float* d_res_stats, const float* d_res_gamma, const float* d_res_beta, float res_eps, float* d_res_copy, const uvl_tuning* tune, void* stream) {
    if (!d_a || !d_w || !d_x || M <= 0) return fail(UVL_EINVAL, "uvl_linear_fin: null pointer");
    GemmParams p;
    p.tune = tune;
    p.A = (const bf16_t*)d_a; p.lda = K; p.W = (const bf16_t*)d_w; p.ldw = K; p.bias = d_bias; p.M = M; p.N = N; p.K = K; p.epi = 1; p.C = d_x; p.ldc = N; p.accumulate = accumulate ? 1 : 0;
    p.xn = (bf16_t*)d_xn; p.xn_bs = 0; p.xn_ro = 0; p.st_out = d_stats; p.st_rows = M;
    p.res_st = d_res_stats; p.res_g = d_res_gamma; p.res_b = d_res_beta; p.res_eps = res_eps; p.res_copy = d_res_copy;
    if (!gemm_fin_ok(p)) return fail(UVL_EINVAL, "uvl_linear_fin: need N %% 64 == 0, K %% 128 == 0 (and gamma / beta / accumulate with d_res_stats)");
    HIPCHK(launch_gemm_fin(p, nullptr, (hipStream_t)stream));
    return UVL_OK;
}
extern "C" int uvl_linear_lnf(const void* d_a, const float* d_stats, const void* d_w_folded, const float* d_bias_folded, const float* d_colsum, float eps, void* d_y,
                              int M, int N, int K, int act, const uvl_tuning* tune, void* stream) {
    if (!d_a || !d_stats || !d_w_folded || !d_bias_folded || !d_colsum || !d_y) return fail(UVL_EINVAL, "uvl_linear_lnf: null pointer");
    GemmParams p;
    p.tune = tune;
    p.A = (const bf16_t*)d_a; p.lda = K; p.W = (const bf16_t*)d_w_folded; p.ldw = K; p.bias = d_bias_folded; p.colsum = d_colsum; p.st_in = d_stats; p.ln_eps = eps;
    p.M = M; p.N = N; p.K = K; p.epi = 0; p.C = d_y; p.ldc = N; p.act = act;
    if (launch_gemm_lnf(p, nullptr, nullptr, (hipStream_t)stream) != hipSuccess) return fail(UVL_EINVAL, "uvl_linear_lnf: need N %% 64 == 0, K %% 128 == 0, K <= 1024 (or the launch failed: %s)", hipGetErrorString(hipGetLastError()));
    return UVL_OK;
}
extern "C" int uvl_qkv_project_lnf(const void* d_a, const float* d_stats, const void* d_w_folded, const float* d_bias_folded, const float* d_colsum, float eps,
                                   void* d_q, void* d_k, void* d_vt, int B, int N, int Npad, int D, float q_scale, const uvl_tuning* tune, void* stream) {
    if (!d_a || !d_stats || !d_w_folded || !d_bias_folded || !d_colsum || !d_q || !d_k || !d_vt || D % 64 != 0) return fail(UVL_EINVAL, "uvl_qkv_project_lnf: bad argument");
    GemmParams p;
    p.tune = tune;
    p.A = (const bf16_t*)d_a; p.lda = D; p.W = (const bf16_t*)d_w_folded; p.ldw = D; p.bias = d_bias_folded; p.colsum = d_colsum; p.st_in = d_stats; p.ln_eps = eps;
    p.M = B * N; p.N = 3 * D; p.K = D; p.epi = 2; p.rpb = N; p.q = (bf16_t*)d_q; p.k = (bf16_t*)d_k; p.vt = (bf16_t*)d_vt; p.H = D / 64; p.Npad = Npad; p.D = D; p.q_scale = q_scale;
    if (launch_gemm_lnf(p, nullptr, nullptr, (hipStream_t)stream) != hipSuccess) return fail(UVL_EINVAL, "uvl_qkv_project_lnf: need D %% 128 == 0, D <= 1024 (or the launch failed: %s)", hipGetErrorString(hipGetLastError()));
    return UVL_OK;
}

extern "C" int uvl_layernorm(const float* d_x, const float* d_gamma, const float* d_beta, float eps, void* d_y_bf16, float* d_y_f32, int M, int D, void* stream) {
    if (!d_x || !d_gamma || !d_beta) return fail(UVL_EINVAL, "uvl_layernorm: null pointer");
    LnParams p;
    p.x = d_x; p.M = M; p.D = D; p.gamma = d_gamma; p.beta = d_beta; p.eps = eps; p.y_bf16 = (bf16_t*)d_y_bf16; p.y_f32 = d_y_f32;
    HIPCHK(launch_layernorm(p, (hipStream_t)stream));
    return UVL_OK;
}

extern "C" int uvl_f32_to_bf16(const float* d_in, void* d_out, size_t n, void* stream) {
    HIPCHK(launch_f32_to_bf16(d_in, (bf16_t*)d_out, n, (hipStream_t)stream));
    return UVL_OK;
}
