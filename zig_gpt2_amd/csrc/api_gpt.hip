// api_gpt.hip — model tier of the C ABI: GPTConfig/State/Block/GPT/generate of the reference's
// src/main.zig as one device-resident object.  Weights, per-sequence KV caches and all scratch
// live in a single arena allocated by zg_gpt_create (the State.init / load_gpt moment of the
// reference); a decode step is a fixed chain of kernels captured once into a hipGraph whose
// position, tokens and argmax all live in device memory, so a whole greedy generation is enqueued
// without a host round trip per token.
//
// HBM layout (one hipMalloc, 256-B aligned sub-buffers):
//   [ weights: wte | wpe | ln_f | per layer: c_attn_w c_proj_w c_fc_w mlp_proj_w + fp32 vectors ]
//   [ KV cache: layer x {K,V} x batch x head x ctx x 64 ]   head-major: one head's keys are one
//                                                            contiguous [ctx, 64] slab, so no
//                                                            per-step transpose (ops.zig:153,158)
//   [ scratch: x q h4 attention partials logits argmax partials, control block, token buffers ]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "zg_runtime.h"

using namespace zg;

struct zg_layer {
    void *c_attn_w, *c_proj_w, *c_fc_w, *mlp_proj_w;
    // ZG_GPT_WEIGHTS_F32 handles only: the same matrices split exactly into bf16 planes [out][3 in] = [hi | mid | lo]
    // (filled when the tensor is loaded) — the B operand of the whole-prompt GEMMs
    bf16_t *c_attn_p, *c_proj_p, *c_fc_p, *mlp_proj_p;
    // LayerNorm folded out of the two LayerNorm-fed Linears (gemv_ksplit.hip, gemv_lnk_kernel): c2 = W g, c3 = W b + bias
    float *c_attn_c2, *c_attn_c3, *c_fc_c2, *c_fc_c3;
    float *ln_1_g, *ln_1_b, *c_attn_b, *c_proj_b, *ln_2_g, *ln_2_b, *c_fc_b, *mlp_proj_b;
    void *k_cache, *v_cache;
};

struct zg_gpt {
    zg_gpt_config cfg;
    size_t batch;
    unsigned flags;
    int wt;        // WT_BF16 / WT_F32
    int kv_mode;
    size_t wbytes;  // bytes per matrix element
    char* arena;
    size_t arena_bytes, weight_region_bytes, state_bytes;  // arena = [weight region (absent when borrowed) | state]
    size_t kv_region_bytes;  // the KV caches of all layers: one contiguous stretch of the arena from layers[0].k_cache
    void *wte, *wpe;
    float *ln_f_g, *ln_f_b;
    float *lm_c2, *lm_c3;  // ln_f folded out of lm_head (batched decode): wte g, wte b
    std::vector<zg_layer> layers;
    // state
    StepCtrl* ctrl;
    float *x, *q, *h4, *part, *logits, *part_val;
    int *part_idx, *prompt, *prompt_len, *forced, *cur_token, *out_tokens;
    // split-K combine area of the batched wide Linears (mlp c_proj): [sk_tiles][4][128] floats + one counter per tile
    float* sk_ws;
    int* sk_cnt;
    int sk_tiles;
    // whole-prompt (prefill) scratch, rows = batch * ctx: x fp32 [E], qkv fp32 [3E], split bf16 [kSplit E] and [kSplit 4E]
    float *pf_x, *pf_qkv, *pf_ws;
    size_t pf_ws_floats;
    unsigned* sk_flags;   // stream-K hand-over of the c_attn GEMM (PrefillQkv.sk_*): 512 flag words, zeroed at create
    unsigned sk_epoch;
    bool sk_used;         // a stream-K launch since the last check of gemm_s4_fault
    bf16_t *pf_a, *pf_h;
    // lock-step batch with bf16 weights: activation planes between the kernels of a Block (GemvArgs.pl_in / pl_out):
    // xp = planes of g * x for the next LayerNorm-fed Linear [48 E bytes], hp = planes of gelu(c_fc) [48 * 4E bytes]
    // ap = planes of the merged attention output [48 E bytes], written by the last split of every (sequence, head)
    // (AttnArgs.pl_out; attn_cnt = its arrival counters)
    bf16_t *xp, *hp, *ap;
    int* attn_cnt;
    // tagged hand-overs (GemvArgs.sk_tag, AttnArgs.part_tag): step counter advanced by the embed kernel, (value, tag) words
    unsigned* epoch;
    unsigned spin_limit;  // polls before a poller of a tagged hand-over gives up
    unsigned* fault;  // set by a poller of a tagged hand-over whose bounded wait ran out; checked wherever a call drains the stream
    unsigned long long *sk_tag, *part_tag;
    size_t sk_tag_bytes, part_tag_bytes;
    size_t epochs_since_clear;  // steps enqueued since the tagged words were last zeroed (note_steps)
    bool tags_on;
    // LayerNorm statistics of x by 16-column tile, written by the producers of x (GemvArgs.st_out / st_in)
    float* xst;
    bool st_on;
    bool pl_on;
    int max_splits, lm_grid;
    // pinned host mirrors for small control traffic
    StepCtrl* h_ctrl;
    int* h_ints;  // [batch * ctx] staging for prompts / tokens
    // graphs: one per (64-position bucket of seq_len, with/without lm_head), captured on first use.
    // The bucket's upper bound t_hi is baked into the attention / merge kernels so that their loads
    // do not wait for the exact seq_len (which lives in device memory).
    std::vector<hipGraphExec_t> graphs;
    std::vector<hipGraphExec_t> graphs_k;  // per bucket: graph_steps consecutive steps with lm_head in one graph (generate loop)
    // generate with the sampler (zg_gpt_generate_sample_*): per bucket one step / graph_steps steps with lm_head AND the sampler
    // node behind it; captured at create with ZG_GPT_SAMPLED_GENERATE, otherwise on the first sampled generation
    std::vector<hipGraphExec_t> graphs_s, graphs_ks;
    int* sampled;             // [batch]: the sampler's draw from the last step's logits
    float* samp_ws;           // segment sums of the sampler (sample_workspace_floats)
    SampleParams* samp;       // device: temperature and seed of the generation in flight
    SampleParams* h_samp;     // pinned mirror
    bool gen_sampled;         // the generation in flight draws its tokens
    size_t graph_steps;
    hipStream_t graph_stream;
    size_t steps_enqueued;
    bool ln_folded;  // c2 / c3 of every layer match the weights currently in the arena
    // side-stream L2 prefetcher of the decode chain (prefetch.hip); runs during zg_gpt_generate_enqueue only
    PfCtl* pf_ctl;
    PfJob* pf_jobs;  // [pf_njobs]: embed, 5 per layer, lm_head
    int pf_njobs;
    bool pf_on;
    bool pf_ran;      // a prefetcher was launched by the last generate call
    bool pf_stalled;  // one left on its idle limit (no concurrency with the decode stream, or a host hiccup): sitting out
    int pf_strikes;   // idle-limit exits so far; the third one is final
    int pf_sit_out;   // generate calls left before a stalled prefetcher is tried again
    hipStream_t pf_stream;
    hipEvent_t pf_ev_main, pf_ev_side;
    // independent prompt groups on one GPU (zg_gpt_create_ex): a private stream, so that the decode chains of several handles
    // overlap on the chip, and a weight region borrowed from another handle of the same model
    hipStream_t stream;  // nullptr: the library stream of the moment (zg_set_stream)
    zg_gpt* parent;      // owner of the weight region this handle reads (nullptr: its own)
    int n_children;      // handles borrowing this one's weight region
    char* wbase;         // the weight region: arena, or the parent's
    // a generation in flight between gen_begin and gen_end (zg_gpt_generate_enqueue / _many)
    size_t gen_pos, gen_n, gen_min_prompt, gen_since_sync;
    bool gen_open;
};

static inline hipStream_t gs(const zg_gpt* g) { return g->stream ? g->stream : ctx().stream; }
static inline zg_gpt* root(zg_gpt* g) { return g->parent ? g->parent : g; }

namespace {

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        const size_t o = off;
        off += bytes;
        return o;
    }
};

// One pass computes sizes (both bases nullptr) or assigns pointers: the weight region from wbase (the handle's own arena or the
// one it borrows), everything else — KV caches, scratch, control — from sbase.
void carve(zg_gpt* g, char* wbase, char* sbase) {
    const zg_gpt_config& c = g->cfg;
    const size_t E = c.n_embed, V = c.vocab_size, C = c.context_size, L = c.n_layer, B = g->batch;
    const size_t wb = g->wbytes, kvb = g->kv_mode == 1 ? 2 : g->kv_mode == 2 ? 3 : 4;  // B24: a bf16 plane, then a byte plane
    Carver cv;
    char* base = wbase;
    auto P = [&](size_t bytes) -> char* {
        const size_t o = cv.take(bytes);
        return base ? base + o : nullptr;
    };
    g->wte = P(V * E * wb);
    g->wpe = P(C * E * wb);
    g->ln_f_g = (float*)P(E * 4);
    g->ln_f_b = (float*)P(E * 4);
    g->lm_c2 = (float*)P(V * 4);
    g->lm_c3 = (float*)P(V * 4);
    g->layers.resize(L);
    for (size_t l = 0; l < L; ++l) {
        zg_layer& y = g->layers[l];
        y.c_attn_w = P(3 * E * E * wb);
        y.c_proj_w = P(E * E * wb);
        y.c_fc_w = P(4 * E * E * wb);
        y.mlp_proj_w = P(4 * E * E * wb);
        y.ln_1_g = (float*)P(E * 4);
        y.ln_1_b = (float*)P(E * 4);
        y.c_attn_b = (float*)P(3 * E * 4);
        y.c_proj_b = (float*)P(E * 4);
        y.ln_2_g = (float*)P(E * 4);
        y.ln_2_b = (float*)P(E * 4);
        y.c_fc_b = (float*)P(4 * E * 4);
        y.mlp_proj_b = (float*)P(E * 4);
        y.c_attn_c2 = (float*)P(3 * E * 4);
        y.c_attn_c3 = (float*)P(3 * E * 4);
        y.c_fc_c2 = (float*)P(4 * E * 4);
        y.c_fc_c3 = (float*)P(4 * E * 4);
        y.c_attn_p = y.c_proj_p = y.c_fc_p = y.mlp_proj_p = nullptr;
        if (g->wt == WT_F32 && !(g->flags & ZG_GPT_NO_PREFILL)) {  // inside the weight region: broadcast with the weights
            y.c_attn_p = (bf16_t*)P(3 * E * E * kSplit * 2);
            y.c_proj_p = (bf16_t*)P(E * E * kSplit * 2);
            y.c_fc_p = (bf16_t*)P(4 * E * E * kSplit * 2);
            y.mlp_proj_p = (bf16_t*)P(4 * E * E * kSplit * 2);
        }
    }
    g->weight_region_bytes = (cv.off + 255) & ~(size_t)255;
    cv = Carver{};
    base = sbase;
    for (size_t l = 0; l < L; ++l) {
        g->layers[l].k_cache = P(B * C * E * kvb);
        g->layers[l].v_cache = P(B * C * E * kvb);
    }
    g->ctrl = (StepCtrl*)P(sizeof(StepCtrl));
    g->kv_region_bytes = L ? (size_t)(reinterpret_cast<char*>(g->ctrl) - reinterpret_cast<char*>(g->layers[0].k_cache)) : 0;
    g->x = (float*)P(B * E * 4);
    g->q = (float*)P(B * E * 4);
    g->h4 = (float*)P(B * 4 * E * 4);
    g->part = (float*)P(B * c.n_heads * g->max_splits * kPartStride * 4);
    g->logits = (float*)P(B * V * 4);
    g->part_val = (float*)P(B * 4096 * 4);
    g->part_idx = (int*)P(B * 4096 * 4);
    g->prompt = (int*)P(B * C * 4);
    g->prompt_len = (int*)P(B * 4);
    g->forced = (int*)P(B * 4);
    g->cur_token = (int*)P(B * 4);
    g->out_tokens = (int*)P(B * C * 4);
    g->sampled = (int*)P(B * 4);
    g->samp = (SampleParams*)P(sizeof(SampleParams));
    g->samp_ws = (float*)P(sample_workspace_floats((int)B) * 4);
    g->xp = (bf16_t*)P(E * 48);
    g->hp = (bf16_t*)P(4 * E * 48);
    g->ap = (bf16_t*)P(E * 48);
    g->attn_cnt = (int*)P(8 * c.n_heads * 4);
    g->epoch = (unsigned*)P(256);
    g->xst = (float*)P(((E + 15) / 16) * 8 * 2 * 4);
    g->sk_tag_bytes = ((E + 15) / 16) * 4 * 128 * 8;
    g->part_tag_bytes = 8 * c.n_heads * g->max_splits * kPartStride * 8;
    g->sk_tag = (unsigned long long*)P(g->sk_tag_bytes);
    g->part_tag = (unsigned long long*)P(g->part_tag_bytes);
    g->sk_tiles = (int)((E + 15) / 16);
    g->sk_ws = (float*)P((size_t)g->sk_tiles * 4 * 128 * 4);
    g->sk_cnt = (int*)P((size_t)g->sk_tiles * 4 * 4);  // [tile][4]: the four-wave plane-fed kernel takes one ticket per wave
    g->pf_njobs = (int)(2 + 5 * L);
    g->pf_ctl = (PfCtl*)P(sizeof(PfCtl));
    g->pf_jobs = (PfJob*)P((size_t)g->pf_njobs * sizeof(PfJob));
    g->pf_x = g->pf_qkv = g->pf_ws = nullptr;
    g->sk_flags = nullptr;
    g->sk_epoch = 0;
    g->sk_used = false;
    g->pf_a = g->pf_h = nullptr;
    g->pf_ws_floats = 0;
    if (!(g->flags & ZG_GPT_NO_PREFILL)) {
        g->pf_x = (float*)P(B * C * E * 4);
        g->pf_qkv = (float*)P(B * C * 3 * E * 4);
        g->pf_a = (bf16_t*)P(B * C * kSplit * E * 2);
        g->pf_h = (bf16_t*)P(B * C * kSplit * 4 * E * 2);
        // bf16 weights: 64 MiB of split-K partial sums; fp32 weights: a full fp32 [rows][4 E] GEMM output
        // split-K partials of the prompt GEMMs; fp32 weights: three weight-plane passes of [B ctx, 4 E] at the least
        g->pf_ws_floats = g->wt == WT_BF16 ? (size_t)(16u << 20) : std::max((size_t)(16u << 20), 3 * B * C * 4 * E);
        g->pf_ws = (float*)P(g->pf_ws_floats * 4);
        g->sk_flags = (unsigned*)P(2048);
    }
    g->state_bytes = (cv.off + 255) & ~(size_t)255;
}

int bucket_t_hi(const zg_gpt* g, size_t seq_len) {
    const size_t hi = ((seq_len + 63) / 64) * 64;
    return (int)(hi < g->cfg.context_size ? hi : g->cfg.context_size);
}

GemvArgs base_gemv(const zg_gpt* g, const void* W, const float* bias, size_t N, size_t K, int t_hi) {
    GemvArgs a{};
    a.t_hi = t_hi;
    a.dbg = ctx().dbg;
    a.zero = ctx().d_zero;
    a.W = W;
    a.bias = bias;
    a.N = (int)N;
    a.K = (int)K;
    a.M = (int)g->batch;
    a.eps = 1e-5f;  // LayerNorm.eps default, ops.zig:76
    a.ctrl = g->ctrl;
    a.n_heads = (int)g->cfg.n_heads;
    a.head_dim = 64;
    a.max_splits = g->max_splits;
    a.ctx = (int)g->cfg.context_size;
    a.kv_mode = g->kv_mode;
    a.kv_lo = g->batch * g->cfg.context_size * g->cfg.n_embed * 2;
    a.sk_ws = g->sk_ws;
    a.sk_cnt = g->sk_cnt;
    a.sk_tiles = g->sk_tiles;
    a.progress = g->pf_on ? &g->pf_ctl->progress : nullptr;
    return a;
}

EmbedArgs embed_args(const zg_gpt* g, int finish_only) {
    EmbedArgs e{};
    e.ctrl = g->ctrl;
    e.wte = g->wte;
    e.wpe = g->wpe;
    e.weight_type = g->wt;
    e.n_embed = (int)g->cfg.n_embed;
    e.batch = (int)g->batch;
    e.vocab = (int)g->cfg.vocab_size;
    e.prompt = g->prompt;
    e.prompt_stride = (int)g->cfg.context_size;
    e.prompt_len = g->prompt_len;
    e.forced = g->forced;
    e.cur_token = g->cur_token;
    e.out_tokens = g->out_tokens;
    e.out_stride = (int)g->cfg.context_size;
    e.part_val = g->part_val;
    e.part_idx = g->part_idx;
    e.part_stride = g->lm_grid;  // the lm_head GEMV writes partials [batch][gridDim.x]
    e.n_partials = g->lm_grid;
    e.x = g->x;
    e.pl_out = g->pl_on ? g->xp : nullptr;
    e.pl_g = g->layers[0].ln_1_g;
    e.epoch = (g->pl_on && g->tags_on && finish_only != 1 && finish_only != 2) ? g->epoch : nullptr;
    e.st_out = g->st_on ? g->xst : nullptr;
    e.finish_only = finish_only;
    e.progress = g->pf_on ? &g->pf_ctl->progress : nullptr;
    e.sampled = g->sampled;
    return e;
}

// A planned decode launch: enqueue it, or (rec != nullptr: building the prefetcher's job table at create) describe
// the weight tiles its workgroups read.
int emit_gemv(const zg_gpt* g, const GemvArgs& a, int grid, hipStream_t s, std::vector<PfJob>* rec, unsigned cls) {
    if (!rec) return launch_gemv(a, g->wt, grid, s);
    PfJob j{};
    j.cls = cls;
    const int rows = gemv_rows_per_wg(a, g->wt);
    if (rows > 0) {
        j.kind = PF_WEIGHTS;
        j.base = reinterpret_cast<const char*>(a.W);
        j.total_bytes = (size_t)a.N * a.K * g->wbytes;
        j.wg_bytes = (unsigned)((size_t)rows * a.K * g->wbytes);
        j.n_wg = (unsigned)grid;
        j.touch_bytes = j.wg_bytes;
        if (a.M == 1) {
            const bool lnk = a.prologue == PRO_LAYERNORM && a.ln_c2 != nullptr;
            const float* v[3] = {lnk ? a.ln_g : nullptr, lnk ? a.ln_c2 : a.bias, lnk ? a.ln_c3 : nullptr};
            const size_t n[3] = {(size_t)a.K * 4, (size_t)a.N * 4, (size_t)a.N * 4};
            for (int i = 0; i < 3; ++i) {
                j.aux[i] = reinterpret_cast<const char*>(v[i]);
                j.aux_bytes[i] = v[i] ? (unsigned)n[i] : 0u;
            }
        }
        // a matrix far larger than the L2s (lm_head): only the head of every tile, about 16 MiB in all
        const size_t cap = (size_t)16 << 20;
        if (j.total_bytes > cap) j.touch_bytes = (unsigned)(((size_t)j.wg_bytes * cap / j.total_bytes + 127) & ~(size_t)127);
    }
    rec->push_back(j);
    return ZG_OK;
}

int enqueue_lm_head(zg_gpt* g, hipStream_t s, std::vector<PfJob>* rec = nullptr) {
    const size_t E = g->cfg.n_embed, V = g->cfg.vocab_size;
    // ln_f (main.zig:189) + lm_head = wte, no bias (main.zig:192-194, :312) + greedy partial argmax
    GemvArgs a = base_gemv(g, g->wte, nullptr, V, E, 0);
    a.prologue = PRO_LAYERNORM;
    a.x = g->x;
    a.x_stride = (int)E;
    a.ln_g = g->ln_f_g;
    a.ln_b = g->ln_f_b;
    a.ln_c2 = g->lm_c2;
    a.ln_c3 = g->lm_c3;
    a.epilogue = EPI_ARGMAX;
    a.logits = g->logits;
    a.logits_stride = (int)V;
    a.part_val = g->part_val;
    a.part_idx = g->part_idx;
    const int grid = gemv_plan(a, g->wt);
    ZG_REQUIRE(grid == g->lm_grid, ZG_ERR_ARG, "lm_head grid changed");
    return emit_gemv(g, a, grid, s, rec, 6);
}

// Optional per-kernel event recorder (zg_gpt_profile_step only).
struct StepProf {
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;  // kernel class of the interval ENDING at ev[i]
    size_t n = 0;
};
inline int prof_mark(StepProf* p, int cls, hipStream_t s) {
    if (!p) return ZG_OK;
    if (p->n == p->ev.size()) {
        hipEvent_t e;
        ZG_HIP(hipEventCreate(&e));
        p->ev.push_back(e);
        p->cls.push_back(cls);
    }
    p->cls[p->n] = cls;
    ZG_HIP(hipEventRecord(p->ev[p->n++], s));
    return ZG_OK;
}

// (Re)derive the folded-LayerNorm vectors from the weights in the arena: after loading, or on ranks that received
// the weight region by broadcast.  A handful of small launches outside any graph.
int ensure_ln_folded(zg_gpt* g, hipStream_t s) {
    zg_gpt* r = root(g);  // the vectors live in the weight region: one flag per region, kept by its owner
    if (r->ln_folded) return ZG_OK;
    const int E = (int)g->cfg.n_embed;
    for (const zg_layer& y : g->layers) {
        ZG_TRY(launch_ln_fold(y.c_attn_w, g->wt, y.ln_1_g, y.ln_1_b, y.c_attn_b, 3 * E, E, y.c_attn_c2, y.c_attn_c3, s));
        ZG_TRY(launch_ln_fold(y.c_fc_w, g->wt, y.ln_2_g, y.ln_2_b, y.c_fc_b, 4 * E, E, y.c_fc_c2, y.c_fc_c3, s));
    }
    ZG_TRY(launch_ln_fold(g->wte, g->wt, g->ln_f_g, g->ln_f_b, nullptr, (int)g->cfg.vocab_size, E, g->lm_c2, g->lm_c3, s));
    // handles that share the region run on other streams: the vectors must be complete before any of them reads the flag
    if (r->n_children > 0 || g->parent) ZG_HIP(hipStreamSynchronize(s));
    r->ln_folded = true;
    return ZG_OK;
}

// One decode step = GPT.forward (main.zig:178-195) for all sequences.
// `only` >= 0 (measurement): launch just that kernel class of layer `only_layer`.
// rec != nullptr: nothing is launched; the step's launches are described for the prefetcher instead (emit_gemv).
// salt >= 0 (measurement chains of one kernel class): launch ids of the tagged hand-overs by chain position instead of by
// layer, so that consecutive launches of the chain never find each other's tags.
int env_int(const char* name, int dflt);

int enqueue_step(zg_gpt* g, bool with_logits, int t_hi, hipStream_t s, StepProf* prof = nullptr, int only = -1, size_t only_layer = 0,
                 std::vector<PfJob>* rec = nullptr, int salt = -1, bool with_sampler = false) {
    const size_t E = g->cfg.n_embed;
    auto launch_id = [&](size_t l, int k) { return (unsigned)(salt >= 0 ? 1 + (2 * salt + k) % 254 : 2 * (int)l + 1 + k); };
    ZG_TRY(prof_mark(prof, -1, s));
    if (rec) rec->push_back(PfJob{});
    else if (only < 0 || only == 0) {
        EmbedArgs e = embed_args(g, only == 0 ? 3 : 0);  // main.zig:179-183
        ZG_TRY(launch_embed_step(e, s));
    }
    ZG_TRY(prof_mark(prof, 0, s));
    for (size_t l = (only < 0 ? 0 : only_layer); l < (only < 0 ? g->cfg.n_layer : only_layer + 1); ++l) {
        const zg_layer& y = g->layers[l];
        if (only < 0 || only == 1) {   // ln_1 + c_attn + split_qkv + cache append: main.zig:121-123, ops.zig:143-157
            GemvArgs a = base_gemv(g, y.c_attn_w, y.c_attn_b, 3 * E, E, t_hi);
            a.prologue = PRO_LAYERNORM;
            a.x = g->x;
            a.x_stride = (int)E;
            a.ln_g = y.ln_1_g;
            a.ln_b = y.ln_1_b;
            a.ln_c2 = y.c_attn_c2;
            a.ln_c3 = y.c_attn_c3;
            a.epilogue = EPI_QKV;
            a.pl_in = g->pl_on ? g->xp : nullptr;
            a.st_in = g->st_on ? g->xst : nullptr;
            a.q = g->q;
            a.k_cache = y.k_cache;
            a.v_cache = y.v_cache;
            const int grid = gemv_plan(a, g->wt);
            ZG_TRY(emit_gemv(g, a, grid, s, rec, 1));
            ZG_TRY(prof_mark(prof, 1, s));
        }
        if (only < 0 || only == 2) {   // scaled_dot_product_attention over the cache: ops.zig:160 -> :249-307
            AttnArgs a{};
            a.q = g->q;
            a.k = y.k_cache;
            a.v = y.v_cache;
            a.stride_b = (long)(g->cfg.context_size * E);
            a.stride_h = (long)(g->cfg.context_size * 64);
            a.stride_t = 64;
            a.kv_mode = g->kv_mode;
            a.kv_lo = g->batch * g->cfg.context_size * E * 2;
            a.n_heads = (int)g->cfg.n_heads;
            a.head_dim = 64;
            a.batch = (int)g->batch;
            a.ctrl = g->ctrl;
            a.t_hi = t_hi;
            a.max_splits = g->max_splits;
            a.part = g->part;
            a.progress = g->pf_on ? &g->pf_ctl->progress : nullptr;
            if (g->pl_on) {
                a.pl_out = g->ap;
                a.merge_cnt = g->attn_cnt;
                if (g->tags_on && 2 * l + 2 <= 255) {
                    a.epoch = g->epoch;
                    a.launch_id = launch_id(l, 0);
                    a.part_tag = g->part_tag;
                    a.fault = g->fault;
                    a.spin_limit = g->spin_limit;
                }
            }
            if (rec) {  // the K and V rows of earlier positions, laid out for this grid
                PfJob j{};
                j.kind = PF_KV;
                j.cls = 2;
                j.base = reinterpret_cast<const char*>(y.k_cache);
                j.base2 = reinterpret_cast<const char*>(y.v_cache);
                j.n_heads = (unsigned)g->cfg.n_heads;
                j.ctx = (unsigned)g->cfg.context_size;
                j.batch = (unsigned)g->batch;
                j.row_bytes = g->kv_mode ? 128u : 256u;  // (B24: the bf16 plane; its byte plane is left to the kernel)
                rec->push_back(j);
            } else
                ZG_TRY(launch_attn_decode(a, s));
            ZG_TRY(prof_mark(prof, 2, s));
        }
        if (only < 0 || only == 3) {   // merge heads + attn c_proj + residual: ops.zig:171-172, main.zig:136-139
            GemvArgs a = base_gemv(g, y.c_proj_w, y.c_proj_b, E, E, t_hi);
            a.prologue = PRO_ATTN_MERGE;
            a.part = g->part;
            a.epilogue = EPI_RESIDUAL;
            a.y = g->x;
            a.y_stride = (int)E;
            a.resid = g->x;
            a.resid_stride = (int)E;
            if (g->pl_on) {  // the heads arrive merged, as planes
                a.prologue = PRO_NONE;
                a.pl_in = g->ap;
                a.pl_out = g->xp;
                a.pl_g = y.ln_2_g;
                a.st_out = g->st_on ? g->xst : nullptr;
            }
            const int grid = gemv_plan(a, g->wt);
            ZG_TRY(emit_gemv(g, a, grid, s, rec, 3));
            ZG_TRY(prof_mark(prof, 3, s));
        }
        if (only < 0 || only == 4) {   // ln_2 + c_fc + gelu: main.zig:140, :79-80
            GemvArgs a = base_gemv(g, y.c_fc_w, y.c_fc_b, 4 * E, E, t_hi);
            a.prologue = PRO_LAYERNORM;
            a.x = g->x;
            a.x_stride = (int)E;
            a.ln_g = y.ln_2_g;
            a.ln_b = y.ln_2_b;
            a.ln_c2 = y.c_fc_c2;
            a.ln_c3 = y.c_fc_c3;
            a.epilogue = EPI_GELU;
            a.y = g->h4;
            a.y_stride = (int)(4 * E);
            if (g->pl_on) {  // gelu(c_fc) leaves as planes only
                a.st_in = g->st_on ? g->xst : nullptr;
                a.pl_in = g->xp;
                a.pl_out = g->hp;
                a.y = nullptr;
            }
            const int grid = gemv_plan(a, g->wt);
            ZG_TRY(emit_gemv(g, a, grid, s, rec, 4));
            ZG_TRY(prof_mark(prof, 4, s));
        }
        if (only < 0 || only == 5) {   // mlp c_proj + residual: main.zig:81, :142-145
            GemvArgs a = base_gemv(g, y.mlp_proj_w, y.mlp_proj_b, E, 4 * E, t_hi);
            a.prologue = PRO_NONE;
            a.x = g->h4;
            a.x_stride = (int)(4 * E);
            a.epilogue = EPI_RESIDUAL;
            a.y = g->x;
            a.y_stride = (int)E;
            a.resid = g->x;
            a.resid_stride = (int)E;
            if (g->pl_on) {
                if (g->tags_on && 2 * l + 2 <= 255) {
                    a.epoch = g->epoch;
                    a.launch_id = launch_id(l, 1);
                    a.sk_tag = g->sk_tag;
                    a.fault = g->fault;
                    a.spin_limit = g->spin_limit;
                }
                a.pl_in = g->hp;
                if (l + 1 < g->cfg.n_layer) {  // the next Block's ln_1 + c_attn (ln_f + lm_head reads x itself)
                    a.pl_out = g->xp;
                    a.pl_g = g->layers[l + 1].ln_1_g;
                    a.st_out = g->st_on ? g->xst : nullptr;
                }
            }
            const int grid = gemv_plan(a, g->wt);
            ZG_TRY(emit_gemv(g, a, grid, s, rec, 5));
            ZG_TRY(prof_mark(prof, 5, s));
        }
    }
    if (with_logits && (only < 0 || only == 6)) {
        ZG_TRY(enqueue_lm_head(g, s, rec));
        ZG_TRY(prof_mark(prof, 6, s));
    }
    // GPT.sample's tail (main.zig:200-206) on the logits of this step: the next step's embed kernel feeds what it draws (mode 2)
    if (with_sampler && with_logits && only < 0 && !rec)
        ZG_TRY(launch_sample_step(g->logits, (int)g->batch, (int)g->cfg.vocab_size, g->samp, g->ctrl, g->part_val, g->lm_grid, g->lm_grid, g->samp_ws,
                                  g->sampled, s));
    return ZG_OK;
}

// Whole-prompt forward of positions 0..P-1 of every sequence (tokens in g->prompt): fills the KV caches
// exactly as P calls of GPT.forward would (main.zig:331-334) and leaves the residual stream of all rows in
// pf_x.  With `last_block_full` false the last Block stops after its cache append: nothing downstream of
// it is needed when generation re-feeds the last prompt token (main.zig:337).
// fp32 weights (ZG_GPT_WEIGHTS_F32): the same pass; the weight operand is then the exact three-term bf16 split of the fp32
// matrix (plane-major, made when the tensor is loaded) and every GEMM runs as three partial passes of the same kernel — the
// six plane products above 2^-24 of the leading one, fp32-sgemm grade (prefill.hip launch_prefill_gemm_wp).
// (Measured and dropped in round 4: replaying this pass from a hipGraph per prompt length.  0.844 against 0.749 ms at 64
// tokens, 1.645 against 1.553 ms at 1023 — the ~10 us a launch costs here is the kernels' own latency at these sizes, not host
// overhead, and a graph launch adds its own ~10 us; profiles/round4_prefill_graph.jsonl.)
int enqueue_prefill(zg_gpt* g, size_t P, bool last_block_full, hipStream_t s) {
    const bool f32w = g->wt == WT_F32;
    const int np = f32w ? kWeightPlanes : (g->flags & ZG_GPT_PREFILL_2PLANE) ? 2 : kSplit;
    const size_t E = g->cfg.n_embed, L = g->cfg.n_layer, H = g->cfg.n_heads, C = g->cfg.context_size;
    const int B = (int)g->batch, M = (int)(g->batch * P), iE = (int)E;
    ZG_TRY(launch_embed_prefill(g->prompt, (int)C, B, (int)P, g->wte, g->wpe, g->wt, iE, g->pf_x, s));
    ZG_TRY(launch_ln_split(g->pf_x, M, iE, g->layers[0].ln_1_g, g->layers[0].ln_1_b, 1e-5f, g->pf_a, s));
    for (size_t l = 0; l < L; ++l) {
        const zg_layer& y = g->layers[l];
        // pf_a holds split(ln_1(x)) here: from the line above or from the tail of the previous Block's last GEMM
        // c_attn with the cache append of ops.zig:152-157 in its epilogue
        PrefillQkv qa{(int)P, iE, (int)H, (int)C, g->kv_mode, y.k_cache, y.v_cache, g->batch * C * E * 2};
        if (!f32w && g->sk_flags != nullptr) {  // (the persistent GEMM may hand half tiles over between workgroups: gemm_s4.hip SK)
            qa.sk_ws = g->pf_ws;
            qa.sk_ws_bytes = g->pf_ws_floats * 4;
            qa.sk_flags = g->sk_flags;
            qa.sk_flags_words = 512;  // (carve: 2048 bytes)
            qa.sk_epoch = ++g->sk_epoch;
            g->sk_used = true;
        }
        ZG_TRY(launch_prefill_gemm(g->pf_a, f32w ? y.c_attn_p : (const bf16_t*)y.c_attn_w, y.c_attn_b, g->pf_qkv, M, 3 * iE, iE, 3 * iE, PF_QKV,
                                   g->pf_ws, g->pf_ws_floats, nullptr, s, &qa, np));
        if (l + 1 == L && !last_block_full) break;
        // (an fp32 cache is read directly: the c_attn epilogue then need not store the k / v columns of qkv)
        ZG_TRY(launch_attn_prefill(g->pf_qkv, g->pf_a, B, (int)P, iE, (int)H, g->pf_ws, g->pf_ws_floats, g->kv_mode == 0 ? (const float*)y.k_cache : nullptr,
                                       g->kv_mode == 0 ? (const float*)y.v_cache : nullptr, (int)C, s));
        const PrefillLn ln2{y.ln_2_g, y.ln_2_b, 1e-5f, g->pf_a};
        ZG_TRY(launch_prefill_gemm(g->pf_a, f32w ? y.c_proj_p : (const bf16_t*)y.c_proj_w, y.c_proj_b, g->pf_x, M, iE, iE, iE, PF_RESID, g->pf_ws,
                                   g->pf_ws_floats, &ln2, s, nullptr, np));
        ZG_TRY(launch_prefill_gemm(g->pf_a, f32w ? y.c_fc_p : (const bf16_t*)y.c_fc_w, y.c_fc_b, g->pf_h, M, 4 * iE, iE, 0, PF_GELU_SPLIT, g->pf_ws,
                                   g->pf_ws_floats, nullptr, s, nullptr, np));
        const bool more = l + 1 < L;
        const PrefillLn ln1{more ? g->layers[l + 1].ln_1_g : nullptr, more ? g->layers[l + 1].ln_1_b : nullptr, 1e-5f, g->pf_a};
        ZG_TRY(launch_prefill_gemm(g->pf_h, f32w ? y.mlp_proj_p : (const bf16_t*)y.mlp_proj_w, y.mlp_proj_b, g->pf_x, M, iE, 4 * iE, iE, PF_RESID, g->pf_ws,
                                   g->pf_ws_floats, more ? &ln1 : nullptr, s, nullptr, np));
    }
    return ZG_OK;
}

// The tagged hand-overs of the lock-step batch poll with a bound; a poller that ran into it raised the fault word.  The
// results of such a step are wrong, so every entry point that drains the stream fails (and clears the word for the next
// call).  The word lives in pinned host memory the kernels store to directly: reading it costs no copy and no second
// synchronisation.  PRECONDITION: the stream has been drained since the steps in question.
int check_fault(zg_gpt* g) {
    // both words are read and cleared before anything is reported: a tag fault left latched behind a stream-K fault of the same
    // drain would be charged to a later, healthy call
    unsigned sk = 0, tag = 0;
    if (g->sk_used) {  // a whole-prompt pass may have handed half tiles over inside gemm_s4: did a consumer give up waiting?
        g->sk_used = false;
        ZG_TRY(gemm_s4_fault(&sk));
    }
    if (g->tags_on) {
        volatile unsigned* f = g->fault;
        tag = *f;
        if (tag) *f = 0;
    }
    if (sk) {
        set_error("a half-tile hand-over of the whole-prompt c_attn GEMM timed out%s: results discarded", tag ? " (and a tagged hand-over of the decode step)" : "");
        return ZG_ERR_HIP;
    }
    if (tag) {
        set_error("a tagged hand-over of the decode step timed out (a workgroup waited %u polls for its writers): results discarded", g->spin_limit);
        return ZG_ERR_HIP;
    }
    return ZG_OK;
}

// A tag is (epoch << 8 | launch id) in 32 bits: 24 bits of the step counter survive, and a slot that is only written at long
// contexts (the high attention splits) could meet its own tag again 2^24 steps later — about an hour of decoding.  So the
// host counts the steps it enqueues and zeroes the tagged words (tag 0 is never valid: launch ids start at 1) on the stream
// every 2^23 of them, in front of the steps of the call that crosses the mark.
int note_steps(zg_gpt* g, size_t n, hipStream_t s) {
    if (!g->tags_on) return ZG_OK;
    g->epochs_since_clear += n;
    if (g->epochs_since_clear < ((size_t)1 << 23)) return ZG_OK;
    g->epochs_since_clear = n;
    ZG_HIP(hipMemsetAsync(g->sk_tag, 0, g->sk_tag_bytes, s));
    ZG_HIP(hipMemsetAsync(g->part_tag, 0, g->part_tag_bytes, s));
    return ZG_OK;
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Side-stream prefetcher (prefetch.hip): job table, control block, low-priority stream.  Called from zg_gpt_create
// BEFORE the graphs are captured (the decode kernels get the progress counter as an argument).
int setup_prefetcher(zg_gpt* g) {
    g->pf_on = g->pf_ran = g->pf_stalled = false;
    g->pf_strikes = g->pf_sit_out = 0;
    g->pf_stream = nullptr;
    g->pf_ev_main = g->pf_ev_side = nullptr;
    // Default: only where it was measured to pay — one sequence and Linears of a few MB (GPT-2 124M: 241 -> 224 us per
    // token; GPT-2 XL's 20 MB matrices cannot be fetched a launch ahead, 8 prompts gain < 1 %).  ZGPT2_PREFETCH=1 / 0 forces.
    // (not beside co-running handles: a handle with a private stream is one of several chains on the chip — §3.3 of DESIGN.md — and
    // the prefetcher's per-XCD placement follows ONE queue's block order; each would also park 96 polling workgroups)
    const bool small = g->stream == nullptr && g->batch == 1 && 4 * g->cfg.n_embed * g->cfg.n_embed * g->wbytes <= ((size_t)6 << 20);
    const int want = env_int("ZGPT2_PREFETCH", small ? 1 : 0);
    if ((g->flags & ZG_GPT_NO_PREFETCH) || !want || gs(g) == nullptr) return ZG_OK;
    if (g->pf_njobs > 255) return ZG_OK;  // the progress word counts launches in 8 bits (n_layer >= 51): no prefetcher, not an error
    std::vector<PfJob> jobs;
    ZG_TRY(enqueue_step(g, true, (int)g->cfg.context_size, nullptr, nullptr, -1, 0, &jobs));
    ZG_REQUIRE((int)jobs.size() == g->pf_njobs, ZG_ERR_ARG, "prefetcher: %zu launches per step", jobs.size());
    ZG_HIP(hipMemcpy(g->pf_jobs, jobs.data(), jobs.size() * sizeof(PfJob), hipMemcpyHostToDevice));
    int least = 0, greatest = 0;
    ZG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    ZG_HIP(hipStreamCreateWithPriority(&g->pf_stream, hipStreamNonBlocking, least));
    ZG_HIP(hipEventCreateWithFlags(&g->pf_ev_main, hipEventDisableTiming));
    ZG_HIP(hipEventCreateWithFlags(&g->pf_ev_side, hipEventDisableTiming));
    g->pf_on = true;
    return ZG_OK;
}

void drop_prefetcher(zg_gpt* g) {
    if (g->pf_stream) {
        (void)hipStreamSynchronize(g->pf_stream);
        (void)hipStreamDestroy(g->pf_stream);
    }
    if (g->pf_ev_main) (void)hipEventDestroy(g->pf_ev_main);
    if (g->pf_ev_side) (void)hipEventDestroy(g->pf_ev_side);
    g->pf_stream = nullptr;
    g->pf_ev_main = g->pf_ev_side = nullptr;
    g->pf_on = false;
}

// Start the prefetcher for a run of decode steps ending at sequence length last_T: the control block is cleared on
// the side stream (behind the previous run's prefetcher), the decode stream waits for that, and the prefetcher
// starts once the decode stream reaches this point.  pf_stop() goes behind the last step.
int pf_start(zg_gpt* g, size_t last_T, hipStream_t s) {
    if (!g->pf_on || s == nullptr) return ZG_OK;
    if (g->pf_ran) {  // how did the previous one leave?  (the decode stream was synchronised by the caller)
        ZG_HIP(hipStreamSynchronize(g->pf_stream));
        unsigned why[8];
        ZG_HIP(hipMemcpy(why, g->pf_ctl->exit_reason, sizeof(why), hipMemcpyDeviceToHost));
        bool idle_exit = false;
        for (unsigned w : why) idle_exit |= w == 2u;
        g->pf_ran = false;
        if (idle_exit) {  // a strike, not a verdict: one slow host moment inside a generate loop produces the same exit
            g->pf_stalled = true;
            ++g->pf_strikes;
            g->pf_sit_out = 8;  // generate calls without it before the next try
        }
    }
    if (g->pf_stalled) {
        if (g->pf_strikes >= 3 || g->pf_sit_out-- > 0) return ZG_OK;
        g->pf_stalled = false;  // try again
    }
    ZG_HIP(hipMemsetAsync(g->pf_ctl, 0, sizeof(PfCtl), g->pf_stream));
    ZG_HIP(hipEventRecord(g->pf_ev_side, g->pf_stream));
    ZG_HIP(hipStreamWaitEvent(s, g->pf_ev_side, 0));
    ZG_HIP(hipEventRecord(g->pf_ev_main, s));
    ZG_HIP(hipStreamWaitEvent(g->pf_stream, g->pf_ev_main, 0));
    PfArgs a{};
    a.ctl = g->pf_ctl;
    a.jobs = g->pf_jobs;
    a.njobs = g->pf_njobs;
    // the measured optimum of the round-2 / round-3 sweeps (profiles/NOTEBOOK.md), constants since round 5
    a.lead = 2;    // launches ahead of the running one
    a.nsub = 12;   // prefetcher workgroups per XCD
    a.max_T = (int)last_T;
    a.idle_limit = (unsigned)env_int("ZGPT2_PF_IDLE", 100000);  // polls without progress (~0.1 s) before it gives up (0: the test of that exit)
    a.sleep = 1;
    a.cls_mask = 0x3e;  // every class but lm_head (its head start measured a net loss)
    a.line_shift = 7;   // one touch per 128-byte line
    a.cap_bytes = 0;
    g->pf_ran = true;
    return launch_prefetcher(a, g->pf_stream);
}

int pf_stop(zg_gpt* g, hipStream_t s) {
    if (!g->pf_on || s == nullptr) return ZG_OK;
    ZG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&g->pf_ctl->progress), (int)PF_STOP, 1, s));
    return ZG_OK;
}

size_t prefill_min() { return 4; }  // shorter prompts go through the decode chain (measured: the whole-prompt pass pays from 4 tokens up)

void drop_graphs(zg_gpt* g) {
    for (auto* v : {&g->graphs, &g->graphs_k, &g->graphs_s, &g->graphs_ks})
        for (auto& e : *v)
            if (e) {
                (void)hipGraphExecDestroy(e);
                e = nullptr;
            }
}

// Run one decode step at sequence length seq_len: replay the graph of its bucket (capturing it on
// first use), or launch eagerly when graphs are disabled / the stream cannot be captured.
// n_steps consecutive steps captured on stream cs into *out
int capture_steps(zg_gpt* g, hipGraphExec_t* out, bool with_logits, int t_hi, size_t n_steps, hipStream_t cs, bool with_sampler = false) {
    hipGraph_t graph = nullptr;
    ZG_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    int st = ZG_OK;
    for (size_t i = 0; i < n_steps && st == ZG_OK; ++i) st = enqueue_step(g, with_logits, t_hi, cs, nullptr, -1, 0, nullptr, -1, with_sampler);
    hipError_t e = hipStreamEndCapture(cs, &graph);
    if (st != ZG_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return st;
    }
    ZG_HIP(e);
    ZG_HIP(hipGraphInstantiate(out, graph, nullptr, nullptr, 0));
    ZG_HIP(hipGraphDestroy(graph));
    return ZG_OK;
}

int capture_bucket(zg_gpt* g, size_t idx, hipStream_t s) {
    const size_t seq_len = (idx / 2 + 1) * 64;  // any length of the bucket: only its upper bound is baked in
    const bool with_logits = idx & 1;
    if (g->graphs.size() <= idx) g->graphs.resize(idx + 1, nullptr);
    if (g->graphs[idx]) return ZG_OK;
    const int t_hi = bucket_t_hi(g, seq_len);
    return capture_steps(g, &g->graphs[idx], with_logits, t_hi, 1, s);
}

// graph_steps consecutive decode steps (all with lm_head, all in 64-position bucket b) as ONE graph: the position lives
// in device memory, so the same kernels simply repeat; saves the gap between graph launches in the generate loop.
int capture_multi(zg_gpt* g, size_t b, hipStream_t s) {
    if (g->graphs_k.size() <= b) g->graphs_k.resize(b + 1, nullptr);
    if (g->graphs_k[b]) return ZG_OK;
    const int t_hi = bucket_t_hi(g, (b + 1) * 64);
    return capture_steps(g, &g->graphs_k[b], true, t_hi, g->graph_steps, s);
}

// The sampled twins of a bucket's graphs (one step / graph_steps steps, each with lm_head and the sampler node).
int capture_sampled(zg_gpt* g, size_t b, bool multi, hipStream_t s) {
    auto& v = multi ? g->graphs_ks : g->graphs_s;
    if (v.size() <= b) v.resize(b + 1, nullptr);
    if (v[b]) return ZG_OK;
    return capture_steps(g, &v[b], true, bucket_t_hi(g, (b + 1) * 64), multi ? g->graph_steps : 1, s, true);
}

// All decode graphs of a handle (two per 64-position bucket: with / without lm_head) for stream s.  Called from
// zg_gpt_create — the State.init moment (main.zig:46-64) — so that no forward allocates; a later zg_set_stream
// re-captures them on the first call that sees the new stream.
int capture_all(zg_gpt* g, hipStream_t s) {
    if ((g->flags & ZG_GPT_NO_GRAPH) || s == nullptr) return ZG_OK;
    if (g->graph_stream != s) {
        drop_graphs(g);
        g->graph_stream = s;
    }
    const size_t n = ((g->cfg.context_size + 63) / 64) * 2;
    for (size_t idx = 0; idx < n; ++idx) ZG_TRY(capture_bucket(g, idx, s));
    if (g->graph_steps > 1)
        for (size_t b = 0; b < n / 2; ++b) ZG_TRY(capture_multi(g, b, s));
    if (g->flags & ZG_GPT_SAMPLED_GENERATE)
        for (size_t b = 0; b < n / 2; ++b) {
            ZG_TRY(capture_sampled(g, b, false, s));
            if (g->graph_steps > 1) ZG_TRY(capture_sampled(g, b, true, s));
        }
    return ZG_OK;
}

// Run one decode step at sequence length seq_len: replay the graph of its bucket, or launch eagerly when graphs
// are disabled / the stream cannot be captured.
int run_step(zg_gpt* g, bool with_logits, size_t seq_len, hipStream_t s, bool with_sampler = false) {
    ZG_TRY(ensure_ln_folded(g, s));
    const int t_hi = bucket_t_hi(g, seq_len);
    with_sampler = with_sampler && with_logits;
    if ((g->flags & ZG_GPT_NO_GRAPH) || s == nullptr) return enqueue_step(g, with_logits, t_hi, s, nullptr, -1, 0, nullptr, -1, with_sampler);
    if (g->graph_stream != s) ZG_TRY(capture_all(g, s));  // the caller switched streams after zg_gpt_create
    if (with_sampler) {
        const size_t b = (seq_len + 63) / 64 - 1;
        ZG_TRY(capture_sampled(g, b, false, s));
        ZG_HIP(hipGraphLaunch(g->graphs_s[b], s));
        return ZG_OK;
    }
    const size_t idx = ((seq_len + 63) / 64 - 1) * 2 + (with_logits ? 1 : 0);
    ZG_TRY(capture_bucket(g, idx, s));  // no-op: captured at create
    ZG_HIP(hipGraphLaunch(g->graphs[idx], s));
    return ZG_OK;
}

int upload_f32(const float* src, size_t n, void* dst, bool as_bf16, hipStream_t s) {
    // src may be host or device; matrices are converted on the device.
    Ctx& c = ctx();
    const float* dsrc = src;
    if (!is_device_ptr(src)) {
        if (!as_bf16) {
            ZG_HIP(hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, s));
            ZG_HIP(hipStreamSynchronize(s));
            return ZG_OK;
        }
        // chunk through the staging arena
        const size_t cap = c.stage_cap / 4;
        ZG_REQUIRE(cap > 0, ZG_ERR_STAGING, "no staging arena");
        for (size_t o = 0; o < n; o += cap) {
            const size_t m = (n - o < cap) ? (n - o) : cap;
            ZG_HIP(hipMemcpyAsync(c.stage, src + o, m * 4, hipMemcpyHostToDevice, s));
            ZG_TRY(launch_f32_to_bf16(reinterpret_cast<const float*>(c.stage), reinterpret_cast<bf16_t*>(dst) + o, m, s));
            ZG_HIP(hipStreamSynchronize(s));
        }
        return ZG_OK;
    }
    if (as_bf16) ZG_TRY(launch_f32_to_bf16(dsrc, reinterpret_cast<bf16_t*>(dst), n, s));
    else ZG_HIP(hipMemcpyAsync(dst, dsrc, n * 4, hipMemcpyDeviceToDevice, s));
    ZG_HIP(hipStreamSynchronize(s));
    return ZG_OK;
}

}  // namespace

extern "C" {

int zg_gpt_create(zg_gpt** out, const zg_gpt_config* config, size_t batch, unsigned flags) {
    return zg_gpt_create_ex(out, config, batch, flags, nullptr);
}

int zg_gpt_create_ex(zg_gpt** out, const zg_gpt_config* config, size_t batch, unsigned flags, const zg_gpt_options* opt) {
    ZG_TRY(require_init());
    ZG_REQUIRE(out && config, ZG_ERR_ARG, "zg_gpt_create: null argument");
    const zg_gpt_config& c = *config;
    zg_gpt* parent = opt ? opt->share_weights_with : nullptr;
    if (parent) {
        ZG_REQUIRE(parent->parent == nullptr, ZG_ERR_ARG, "zg_gpt_create_ex: share_weights_with must own its weights");
        ZG_REQUIRE(memcmp(&parent->cfg, config, sizeof(zg_gpt_config)) == 0, ZG_ERR_SHAPE, "zg_gpt_create_ex: a handle shares weights with one of the same config only");
        const unsigned same = ZG_GPT_WEIGHTS_F32 | ZG_GPT_NO_PREFILL;  // what decides the layout of the weight region
        ZG_REQUIRE((parent->flags & same) == (flags & same), ZG_ERR_ARG, "zg_gpt_create_ex: weight type / prefill flags differ from the weight owner's");
    }
    ZG_REQUIRE(c.n_heads > 0 && c.n_embed % c.n_heads == 0 && c.n_embed / c.n_heads == 64, ZG_ERR_UNSUPPORTED,
               "head_dim %zu != 64 (GPT-2 family only)", c.n_heads ? c.n_embed / c.n_heads : (size_t)0);
    ZG_REQUIRE(c.n_embed % 8 == 0 && c.n_embed * 4 <= 8192, ZG_ERR_UNSUPPORTED, "n_embed %zu unsupported", c.n_embed);
    ZG_REQUIRE(batch >= 1 && batch <= 8, ZG_ERR_UNSUPPORTED, "batch %zu outside 1..8", batch);
    ZG_REQUIRE(c.vocab_size > 0 && c.context_size > 0 && c.n_layer > 0, ZG_ERR_ARG, "empty config");
    ZG_REQUIRE(!(flags & ZG_GPT_KV_F16) || !(flags & ZG_GPT_KV_B24), ZG_ERR_ARG, "ZG_GPT_KV_F16 and ZG_GPT_KV_B24 exclude each other");
    zg_gpt* g = new zg_gpt();
    g->cfg = c;
    g->batch = batch;
    g->flags = flags;
    g->wt = (flags & ZG_GPT_WEIGHTS_F32) ? WT_F32 : WT_BF16;
    g->wbytes = g->wt == WT_BF16 ? 2 : 4;
    g->kv_mode = (flags & ZG_GPT_KV_F16) ? 1 : (flags & ZG_GPT_KV_B24) ? 2 : 0;
    g->max_splits = (int)((c.context_size + kAttnChunk - 1) / kAttnChunk);
    g->graph_stream = nullptr;
    g->stream = nullptr;
    g->parent = parent;
    g->n_children = 0;
    g->gen_open = false;
    g->pf_stream = nullptr;
    g->pf_ev_main = g->pf_ev_side = nullptr;
    carve(g, nullptr, nullptr);
    g->arena_bytes = (parent ? 0 : g->weight_region_bytes) + g->state_bytes;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&g->arena), g->arena_bytes);
    if (e != hipSuccess) {
        delete g;
        return hip_fail(e, "hipMalloc(model arena)", __FILE__, __LINE__);
    }
    g->wbase = parent ? parent->wbase : g->arena;
    char* const sbase = parent ? g->arena : g->arena + g->weight_region_bytes;
    carve(g, g->wbase, sbase);
    (void)hipMemset(sbase, 0, g->state_bytes);
    if (opt && opt->own_stream) {  // a private stream: its priority decides the hardware queue it shares (profiles/NOTEBOOK.md §5)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);  // numerically: greatest <= 0 <= least
        const int pr = opt->stream_priority > 0 ? greatest : opt->stream_priority < 0 ? least : 0;
        e = hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, pr);
        if (e != hipSuccess) {
            (void)hipFree(g->arena);
            delete g;
            return hip_fail(e, "hipStreamCreateWithPriority(handle stream)", __FILE__, __LINE__);
        }
    }
    {
        GemvArgs a = base_gemv(g, g->wte, nullptr, c.vocab_size, c.n_embed, 0);
        a.prologue = PRO_LAYERNORM;
        a.epilogue = EPI_ARGMAX;  // the plan depends on the epilogue (workgroup width)
        g->lm_grid = gemv_plan(a, g->wt);
    }
    {   // the widest input of a Block (mlp c_proj: 4 E floats per sequence) must fit the batched kernels' LDS
        GemvArgs a = base_gemv(g, g->layers[0].mlp_proj_w, nullptr, c.n_embed, 4 * c.n_embed, 0);
        if (!gemv_supported(a, g->wt)) {
            if (g->stream) (void)hipStreamDestroy(g->stream);
            (void)hipFree(g->arena);
            delete g;
            set_error("batch %zu with n_embed %zu: %zu input rows of 4*n_embed floats do not fit the LDS (use a smaller batch)",
                      batch, c.n_embed, batch);
            return ZG_ERR_UNSUPPORTED;
        }
    }
    g->pl_on = false;
    if (g->wt == WT_BF16 && batch >= 2 && !(decode_paths_off() & 1)) {  // all three plane-fed Linears on the matrix-core path?
        const zg_layer& y = g->layers[0];
        GemvArgs a1 = base_gemv(g, y.c_attn_w, y.c_attn_b, 3 * c.n_embed, c.n_embed, 0);
        a1.prologue = PRO_LAYERNORM;
        a1.ln_c2 = y.c_attn_c2;
        a1.ln_c3 = y.c_attn_c3;
        a1.epilogue = EPI_QKV;
        GemvArgs a4 = base_gemv(g, y.c_fc_w, y.c_fc_b, 4 * c.n_embed, c.n_embed, 0);
        a4.prologue = PRO_LAYERNORM;
        a4.ln_c2 = y.c_fc_c2;
        a4.ln_c3 = y.c_fc_c3;
        a4.epilogue = EPI_GELU;
        GemvArgs a5 = base_gemv(g, y.mlp_proj_w, y.mlp_proj_b, c.n_embed, 4 * c.n_embed, 0);
        a5.prologue = PRO_NONE;
        a5.epilogue = EPI_RESIDUAL;
        GemvArgs a3 = base_gemv(g, y.c_proj_w, y.c_proj_b, c.n_embed, c.n_embed, 0);
        a3.prologue = PRO_NONE;
        a3.epilogue = EPI_RESIDUAL;
        g->pl_on = c.n_embed % 32 == 0 && gemv_planes_ok(a1, g->wt) && gemv_planes_ok(a3, g->wt) && gemv_planes_ok(a4, g->wt) &&
                   gemv_planes_ok(a5, g->wt);
    }
    g->tags_on = g->pl_on && !(decode_paths_off() & 4);
    g->spin_limit = (unsigned)env_int("ZGPT2_TAG_SPIN_LIMIT", 1 << 20);
    g->st_on = false;
    if (g->pl_on && !(decode_paths_off() & 8) && c.n_embed % 16 == 0 && c.n_embed / 16 <= 128) {
        // every producer and consumer of x must be the four-wave kernel
        const zg_layer& y = g->layers[0];
        GemvArgs a1 = base_gemv(g, y.c_attn_w, y.c_attn_b, 3 * c.n_embed, c.n_embed, 0);
        a1.prologue = PRO_LAYERNORM;
        a1.x_stride = (int)c.n_embed;
        a1.ln_c2 = y.c_attn_c2;
        a1.ln_c3 = y.c_attn_c3;
        a1.epilogue = EPI_QKV;
        GemvArgs a4 = a1;
        a4.N = (int)(4 * c.n_embed);
        a4.epilogue = EPI_GELU;
        GemvArgs a3 = base_gemv(g, y.c_proj_w, y.c_proj_b, c.n_embed, c.n_embed, 0);
        a3.prologue = PRO_NONE;
        a3.epilogue = EPI_RESIDUAL;
        a3.resid_stride = (int)c.n_embed;
        GemvArgs a5 = base_gemv(g, y.mlp_proj_w, y.mlp_proj_b, c.n_embed, 4 * c.n_embed, 0);
        a5.prologue = PRO_NONE;
        a5.epilogue = EPI_RESIDUAL;
        a5.resid_stride = (int)c.n_embed;
        g->st_on = gemv_pl4_ok(a1, g->wt) && gemv_pl4_ok(a3, g->wt) && gemv_pl4_ok(a4, g->wt) && gemv_pl4_ok(a5, g->wt);
    }
    if (g->lm_grid > 4096) {
        if (g->stream) (void)hipStreamDestroy(g->stream);
        (void)hipFree(g->arena);
        delete g;
        set_error("lm_head grid %d exceeds the argmax partial buffer", g->lm_grid);
        return ZG_ERR_UNSUPPORTED;
    }
    g->h_ctrl = nullptr;
    g->h_ints = nullptr;
    // (the control mirror and, 256 bytes behind it, the fault word of the tagged hand-overs: pinned, written by the kernels)
    hipError_t he = hipHostMalloc(reinterpret_cast<void**>(&g->h_ctrl), sizeof(StepCtrl) + 512, hipHostMallocDefault);
    if (he == hipSuccess)
        he = hipHostMalloc(reinterpret_cast<void**>(&g->h_ints), (batch * c.context_size + batch) * sizeof(int), hipHostMallocDefault);
    if (he != hipSuccess) {
        if (g->h_ctrl) (void)hipHostFree(g->h_ctrl);
        if (g->stream) (void)hipStreamDestroy(g->stream);
        (void)hipFree(g->arena);
        delete g;
        return hip_fail(he, "hipHostMalloc(control mirrors)", __FILE__, __LINE__);
    }
    static_assert(sizeof(StepCtrl) + sizeof(SampleParams) <= 256, "the fault word sits 256 bytes behind the control mirror");
    g->h_samp = reinterpret_cast<SampleParams*>(reinterpret_cast<char*>(g->h_ctrl) + 128);  // (same pinned block)
    g->gen_sampled = false;
    g->fault = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(g->h_ctrl) + 256);
    *g->fault = 0;
    g->steps_enqueued = 0;
    g->epochs_since_clear = 0;
    g->ln_folded = false;
    {   // decode steps per graph in the generate loop: a graph launch costs ~7 us of idle queue (124M: 224.8 us per token
        // with 1 step per graph, 220.4 with 2 / 4, 218.3 with 8, 219.5 with 16)
        const int k = env_int("ZGPT2_GRAPH_STEPS", 8);
        g->graph_steps = (k == 2 || k == 4 || k == 8 || k == 16 || k == 32 || k == 64) ? (size_t)k : 1;
    }
    {   // every decode graph is captured and instantiated here, not on the first forward that needs it
        int st = setup_prefetcher(g);
        if (st == ZG_OK) st = capture_all(g, gs(g));
        if (st != ZG_OK) {
            drop_prefetcher(g);
            drop_graphs(g);
            (void)hipHostFree(g->h_ctrl);
            (void)hipHostFree(g->h_ints);
            if (g->stream) (void)hipStreamDestroy(g->stream);
            (void)hipFree(g->arena);
            delete g;
            return st;
        }
    }
    if (parent) ++parent->n_children;
    *out = g;
    return ZG_OK;
}

int zg_gpt_destroy(zg_gpt* g) {
    if (!g) return ZG_OK;
    ZG_REQUIRE(g->n_children == 0, ZG_ERR_ARG, "zg_gpt_destroy: %d handle(s) still borrow this one's weights (destroy them first)", g->n_children);
    (void)hipStreamSynchronize(gs(g));
    drop_prefetcher(g);
    drop_graphs(g);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    if (g->parent) --g->parent->n_children;
    (void)hipFree(g->arena);
    (void)hipHostFree(g->h_ctrl);
    (void)hipHostFree(g->h_ints);
    delete g;
    return ZG_OK;
}

int zg_gpt_stream(zg_gpt* g, void** hip_stream_out) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && hip_stream_out, ZG_ERR_ARG, "zg_gpt_stream: null argument");
    *hip_stream_out = gs(g);
    return ZG_OK;
}

int zg_gpt_load_block_tensor(zg_gpt* g, size_t layer, int slot, const float* src, size_t len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && src && layer < g->cfg.n_layer, ZG_ERR_ARG, "load_block_tensor: bad argument");
    ZG_REQUIRE(g->parent == nullptr, ZG_ERR_ARG, "load_block_tensor: this handle borrows its weights (load them into their owner)");
    const size_t E = g->cfg.n_embed;
    zg_layer& y = g->layers[layer];
    void* dst = nullptr;
    size_t n = 0;
    bool mat = false;
    switch (slot) {
        case ZG_LN_1_G: dst = y.ln_1_g; n = E; break;
        case ZG_LN_1_B: dst = y.ln_1_b; n = E; break;
        case ZG_C_ATTN_W: dst = y.c_attn_w; n = 3 * E * E; mat = true; break;
        case ZG_C_ATTN_B: dst = y.c_attn_b; n = 3 * E; break;
        case ZG_C_PROJ_W: dst = y.c_proj_w; n = E * E; mat = true; break;
        case ZG_C_PROJ_B: dst = y.c_proj_b; n = E; break;
        case ZG_LN_2_G: dst = y.ln_2_g; n = E; break;
        case ZG_LN_2_B: dst = y.ln_2_b; n = E; break;
        case ZG_C_FC_W: dst = y.c_fc_w; n = 4 * E * E; mat = true; break;
        case ZG_C_FC_B: dst = y.c_fc_b; n = 4 * E; break;
        case ZG_MLP_PROJ_W: dst = y.mlp_proj_w; n = 4 * E * E; mat = true; break;
        case ZG_MLP_PROJ_B: dst = y.mlp_proj_b; n = E; break;
        default: ZG_REQUIRE(false, ZG_ERR_ARG, "unknown block slot %d", slot);
    }
    ZG_REQUIRE(len == n, ZG_ERR_SHAPE, "block slot %d expects %zu elements, got %zu", slot, n, len);
    g->ln_folded = false;
    ZG_TRY(upload_f32(src, n, dst, mat && g->wt == WT_BF16, gs(g)));
    if (mat && g->wt == WT_F32 && y.c_attn_p) {  // exact bf16 planes of the fp32 matrix for the whole-prompt GEMMs
        bf16_t* pl = slot == ZG_C_ATTN_W ? y.c_attn_p : slot == ZG_C_PROJ_W ? y.c_proj_p : slot == ZG_C_FC_W ? y.c_fc_p : y.mlp_proj_p;
        // plane-major [3][out][in]: the matrix as ONE row of out * in elements
        ZG_REQUIRE(n <= (size_t)2147483647 / 3, ZG_ERR_SHAPE, "weight matrix of %zu elements", n);  // (split3_kernel indexes 3 n in int)
        ZG_TRY(launch_split3(reinterpret_cast<const float*>(dst), 1, (int)n, pl, gs(g)));
        ZG_HIP(hipStreamSynchronize(gs(g)));
    }
    return ZG_OK;
}

int zg_gpt_load_tensor(zg_gpt* g, int slot, const float* src, size_t len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && src, ZG_ERR_ARG, "load_tensor: bad argument");
    ZG_REQUIRE(g->parent == nullptr, ZG_ERR_ARG, "load_tensor: this handle borrows its weights (load them into their owner)");
    const size_t E = g->cfg.n_embed;
    void* dst = nullptr;
    size_t n = 0;
    bool mat = false;
    switch (slot) {
        case ZG_WTE: dst = g->wte; n = g->cfg.vocab_size * E; mat = true; break;
        case ZG_WPE: dst = g->wpe; n = g->cfg.context_size * E; mat = true; break;
        case ZG_LN_F_G: dst = g->ln_f_g; n = E; break;
        case ZG_LN_F_B: dst = g->ln_f_b; n = E; break;
        default: ZG_REQUIRE(false, ZG_ERR_ARG, "unknown slot %d", slot);
    }
    ZG_REQUIRE(len == n, ZG_ERR_SHAPE, "slot %d expects %zu elements, got %zu", slot, n, len);
    g->ln_folded = false;
    return upload_f32(src, n, dst, mat && g->wt == WT_BF16, gs(g));
}

int zg_gpt_weight_arena(zg_gpt* g, void** device_ptr, size_t* bytes) {
    ZG_REQUIRE(g && device_ptr && bytes, ZG_ERR_ARG, "weight_arena: null argument");
    // The folded-LayerNorm vectors (c2 / c3) live inside the region handed out here.  A sender has to hold valid
    // ones before its bytes are copied, and a receiver must re-derive them from whatever lands in the region — also a
    // receiver that has run before (its flag would still say "folded").  So: fold now if needed (sender side: a few
    // small launches, drained), and mark the vectors stale for the next forward (receiver side: one re-fold).
    if (!root(g)->ln_folded) {
        ZG_TRY(require_init());
        ZG_TRY(ensure_ln_folded(g, gs(g)));
        ZG_HIP(hipStreamSynchronize(gs(g)));
    }
    root(g)->ln_folded = false;
    *device_ptr = g->wbase;
    *bytes = g->weight_region_bytes;
    return ZG_OK;
}

// load_gpt (src/main.zig:304-314) on ONE GPU of the node, then this: the weight region goes to every other rank's arena in one
// RCCL broadcast on the library's stream (dist.hip).  ms_out (optional): device time of the broadcast.
int zg_gpt_broadcast_weights(zg_gpt* g, int root, float* ms_out) {
    ZG_TRY(require_init());
    void* p = nullptr;
    size_t n = 0;
    ZG_TRY(zg_gpt_weight_arena(g, &p, &n));  // (sender: folded vectors valid; receiver: re-fold at the next forward)
    hipStream_t s = gs(g);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms_out) {
        ZG_HIP(hipEventCreate(&e0));
        ZG_HIP(hipEventCreate(&e1));
        ZG_HIP(hipEventRecord(e0, s));
    }
    const int st = zg::dist_broadcast(p, n, root, s);
    if (ms_out && st == ZG_OK) ZG_HIP(hipEventRecord(e1, s));
    const hipError_t he = hipStreamSynchronize(s);
    if (ms_out) {
        if (st == ZG_OK && he == hipSuccess) (void)hipEventElapsedTime(ms_out, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    ZG_TRY(st);
    ZG_HIP(he);
    return ZG_OK;
}

// A new sequence starts (position 1 of the token-at-a-time loop, a whole-prompt pass, a generation): the caches are cleared.  The
// decode attention loads the rows of its whole 64-position bucket and gives the ones at or behind the sequence length a
// probability of exactly 0 — which silences any finite leftover of an earlier sequence, but 0 x NaN is NaN: once a sequence had
// overflowed (or a checkpoint held a NaN) every later sequence on the handle came out NaN as well, repaired weights or not
// (tools/fuzz_errors_gpt.py).  124M, one sequence: 75 MB, ~20 us per generation of 217 ms.
// from_row > 0 (a whole-prompt pass writes rows 0 .. from_row - 1 itself): only the rows behind it, strip by strip — eight
// 1023-token prompts would otherwise pay a 600 MB memset (0.12 ms of a 5.4 ms pass) for one stale row per head.
static int clear_kv(zg_gpt* g, hipStream_t s, size_t from_row = 0) {
    if (g->kv_region_bytes == 0 || g->arena == nullptr) return ZG_OK;
    const size_t C = g->cfg.context_size, E = g->cfg.n_embed, B = g->batch, L = g->cfg.n_layer;
    if (from_row == 0 || from_row * 2 < C) {
        ZG_HIP(hipMemsetAsync(g->layers[0].k_cache, 0, g->kv_region_bytes, s));
        return ZG_OK;
    }
    const size_t elems = B * C * E, kvb = g->kv_mode == 1 ? 2 : g->kv_mode == 2 ? 3 : 4;
    const size_t stride = L > 1 ? (size_t)(reinterpret_cast<char*>(g->layers[1].k_cache) - reinterpret_cast<char*>(g->layers[0].k_cache)) / 2
                                : (size_t)(reinterpret_cast<char*>(g->layers[0].v_cache) - reinterpret_cast<char*>(g->layers[0].k_cache));
    ZG_REQUIRE(stride >= elems * kvb && reinterpret_cast<char*>(g->layers[0].v_cache) - reinterpret_cast<char*>(g->layers[0].k_cache) == (ptrdiff_t)stride,
               ZG_ERR_UNSUPPORTED, "kv clear: cache stride");
    const int strips = (int)(B * g->cfg.n_heads);
    ZG_TRY(launch_kv_clear_tail(g->layers[0].k_cache, (int)(2 * L), stride, 0, g->kv_mode == 0 ? 256 : 128, strips, (int)C, (int)from_row, s));
    if (g->kv_mode == 2) ZG_TRY(launch_kv_clear_tail(g->layers[0].k_cache, (int)(2 * L), stride, elems * 2, 64, strips, (int)C, (int)from_row, s));
    return ZG_OK;
}

int zg_gpt_step_bytes(zg_gpt* g, size_t seq_len, size_t* weight_bytes, size_t* kv_bytes) {
    ZG_REQUIRE(g, ZG_ERR_ARG, "step_bytes: null argument");
    const size_t E = g->cfg.n_embed;
    // wbytes * (sum_layers in*out + V*E) + kvbytes * 2 * T * E * L (per sequence)
    if (weight_bytes) *weight_bytes = g->wbytes * (g->cfg.n_layer * 12 * E * E + g->cfg.vocab_size * E);
    if (kv_bytes) *kv_bytes = (g->kv_mode == 1 ? 2 : g->kv_mode == 2 ? 3 : 4) * 2 * seq_len * E * g->cfg.n_layer * g->batch;
    return ZG_OK;
}

// GPT.forward enqueued on the handle's stream, nothing drained (zg_gpt_forward drains; zg_gpt_sample puts its sampler behind it first)
static int forward_enqueue(zg_gpt* g, size_t seq_len, const size_t* tokens, size_t n_tokens, int compute_logits, float* logits_out, size_t logits_len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && tokens, ZG_ERR_ARG, "gpt_forward: null argument");
    ZG_REQUIRE(n_tokens == g->batch, ZG_ERR_SHAPE, "gpt_forward: %zu tokens for batch %zu", n_tokens, g->batch);
    ZG_REQUIRE(seq_len >= 1 && seq_len <= g->cfg.context_size, ZG_ERR_SHAPE, "gpt_forward: seq_len %zu outside 1..%zu",
               seq_len, g->cfg.context_size);
    const size_t V = g->cfg.vocab_size;
    ZG_REQUIRE(!logits_out || (compute_logits && logits_len >= g->batch * V), ZG_ERR_SHAPE,
               "gpt_forward: logits_out needs compute_logits and %zu elements", g->batch * V);
    hipStream_t s = gs(g);
    for (size_t b = 0; b < g->batch; ++b) {
        ZG_REQUIRE(tokens[b] < V, ZG_ERR_SHAPE, "gpt_forward: token %zu >= vocab %zu", tokens[b], V);
        g->h_ints[b] = (int)tokens[b];
    }
    g->h_ctrl->step = (int)seq_len - 1;
    g->h_ctrl->seq_len = (int)seq_len;
    g->h_ctrl->mode = 1;
    g->h_ctrl->n_partials = g->lm_grid;
    ZG_HIP(hipMemcpyAsync(g->forced, g->h_ints, g->batch * sizeof(int), hipMemcpyHostToDevice, s));
    ZG_HIP(hipMemcpyAsync(g->ctrl, g->h_ctrl, sizeof(StepCtrl), hipMemcpyHostToDevice, s));
    if (seq_len == 1) ZG_TRY(clear_kv(g, s));
    ZG_TRY(note_steps(g, 1, s));
    ZG_TRY(ensure_ln_folded(g, s));
    ZG_TRY(run_step(g, compute_logits != 0, seq_len, s));
    if (logits_out) {
        ZG_HIP(hipMemcpyAsync(logits_out, g->logits, g->batch * V * sizeof(float),
                              is_device_ptr(logits_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    }
    return ZG_OK;
}

int zg_gpt_forward(zg_gpt* g, size_t seq_len, const size_t* tokens, size_t n_tokens, int compute_logits,
                   float* logits_out, size_t logits_len) {
    ZG_TRY(forward_enqueue(g, seq_len, tokens, n_tokens, compute_logits, logits_out, logits_len));
    // h_ints / h_ctrl are reused by the next call: drain before returning.
    ZG_HIP(hipStreamSynchronize(gs(g)));
    return check_fault(g);
}

int zg_gpt_prefill(zg_gpt* g, const size_t* tokens, size_t token_stride, size_t n_tokens, int compute_logits,
                   float* logits_out, size_t logits_len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && tokens, ZG_ERR_ARG, "gpt_prefill: null argument");
    ZG_REQUIRE(g->pf_x != nullptr, ZG_ERR_UNSUPPORTED, "gpt_prefill: the handle was created with ZG_GPT_NO_PREFILL");
    const size_t C = g->cfg.context_size, V = g->cfg.vocab_size, B = g->batch, E = g->cfg.n_embed;
    ZG_REQUIRE(n_tokens >= 1 && n_tokens <= C && n_tokens <= token_stride, ZG_ERR_SHAPE,
               "gpt_prefill: n_tokens %zu outside 1..%zu (stride %zu)", n_tokens, C, token_stride);
    ZG_REQUIRE(!logits_out || (compute_logits && logits_len >= B * V), ZG_ERR_SHAPE,
               "gpt_prefill: logits_out needs compute_logits and %zu elements", B * V);
    hipStream_t s = gs(g);
    ZG_HIP(hipStreamSynchronize(s));
    for (size_t b = 0; b < B; ++b)
        for (size_t i = 0; i < n_tokens; ++i) {
            const size_t t = tokens[b * token_stride + i];
            ZG_REQUIRE(t < V, ZG_ERR_SHAPE, "gpt_prefill: token %zu >= vocab %zu", t, V);
            g->h_ints[b * C + i] = (int)t;
        }
    ZG_HIP(hipMemcpyAsync(g->prompt, g->h_ints, B * C * sizeof(int), hipMemcpyHostToDevice, s));
    ZG_TRY(clear_kv(g, s, n_tokens));
    ZG_TRY(ensure_ln_folded(g, s));
    ZG_TRY(enqueue_prefill(g, n_tokens, compute_logits != 0, s));
    if (compute_logits) {  // ln_f + lm_head of each sequence's last position through the decode kernels
        ZG_HIP(hipMemcpy2DAsync(g->x, E * 4, g->pf_x + (n_tokens - 1) * E, n_tokens * E * 4, E * 4, B,
                                hipMemcpyDeviceToDevice, s));
        g->h_ctrl->step = (int)n_tokens - 1;
        g->h_ctrl->seq_len = (int)n_tokens;
        g->h_ctrl->mode = 1;
        g->h_ctrl->n_partials = g->lm_grid;
        ZG_HIP(hipMemcpyAsync(g->ctrl, g->h_ctrl, sizeof(StepCtrl), hipMemcpyHostToDevice, s));
        ZG_TRY(enqueue_lm_head(g, s));
        if (logits_out)
            ZG_HIP(hipMemcpyAsync(logits_out, g->logits, B * V * sizeof(float),
                                  is_device_ptr(logits_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    }
    ZG_HIP(hipStreamSynchronize(s));
    return check_fault(g);
}

int zg_gpt_argmax(zg_gpt* g, size_t* tokens_out, size_t n_tokens) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && tokens_out && n_tokens == g->batch, ZG_ERR_ARG, "gpt_argmax: bad argument");
    hipStream_t s = gs(g);
    ZG_TRY(launch_embed_step(embed_args(g, 2), s));
    ZG_HIP(hipMemcpyAsync(g->h_ints, g->cur_token, g->batch * sizeof(int), hipMemcpyDeviceToHost, s));
    ZG_HIP(hipStreamSynchronize(s));
    for (size_t b = 0; b < g->batch; ++b) tokens_out[b] = (size_t)g->h_ints[b];
    return ZG_OK;
}

int zg_gpt_sample(zg_gpt* g, size_t seq_len, const size_t* tokens, size_t n_tokens, float temp, const float* uniforms,
                  uint64_t seed, size_t* tokens_out, float* probs_out, size_t probs_len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && tokens && tokens_out && temp > 0.0f, ZG_ERR_ARG, "gpt_sample: bad argument");
    const size_t V = g->cfg.vocab_size, B = g->batch;
    ZG_REQUIRE(!probs_out || probs_len >= B * V, ZG_ERR_SHAPE, "gpt_sample: probs_out needs %zu elements", B * V);
    for (size_t b = 0; uniforms && b < B; ++b)  // (checked before anything is enqueued)
        ZG_REQUIRE(uniforms[b] >= 0.0f && uniforms[b] < 1.0f, ZG_ERR_ARG, "gpt_sample: uniform %f outside [0,1)", uniforms[b]);
    ZG_TRY(forward_enqueue(g, seq_len, tokens, n_tokens, 1, nullptr, 0));  // main.zig:199 (not drained: the sampler goes behind it)
    hipStream_t s = gs(g);
    // (pinned, 320 bytes into the control block: h_ints[0 .. B) is still being read by the forward's token upload)
    float* h_u = reinterpret_cast<float*>(reinterpret_cast<char*>(g->h_ctrl) + 320);
    for (size_t b = 0; b < B; ++b) {
        if (uniforms) {
            h_u[b] = uniforms[b];
        } else {  // counter PRNG (splitmix64 finaliser, 24 random bits), same construction as the synthetic weights
            uint64_t z = seed * 0x9E3779B97F4A7C15ULL + (uint64_t)seq_len * 0xD1B54A32D192ED03ULL + b + 1;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            h_u[b] = (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;
        }
    }
    float* d_u = g->q;  // scratch: q is dead after the forward
    ZG_HIP(hipMemcpyAsync(d_u, h_u, B * sizeof(float), hipMemcpyHostToDevice, s));
    ZG_TRY(launch_sample(g->logits, (int)B, (int)V, temp, d_u, g->part_val, g->lm_grid, g->lm_grid, g->samp_ws, g->cur_token, probs_out != nullptr,
                         s));  // main.zig:200-206
    ZG_HIP(hipMemcpyAsync(g->h_ints + B, g->cur_token, B * sizeof(int), hipMemcpyDeviceToHost, s));
    if (probs_out)
        ZG_HIP(hipMemcpyAsync(probs_out, g->logits, B * V * sizeof(float),
                              is_device_ptr(probs_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    ZG_HIP(hipStreamSynchronize(s));
    ZG_TRY(check_fault(g));
    for (size_t b = 0; b < B; ++b) tokens_out[b] = (size_t)g->h_ints[B + b];
    return ZG_OK;
}

int zg_gpt_hidden(zg_gpt* g, float* x_out, size_t len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && x_out && len >= g->batch * g->cfg.n_embed, ZG_ERR_ARG, "gpt_hidden: bad argument");
    hipStream_t s = gs(g);
    ZG_HIP(hipMemcpyAsync(x_out, g->x, g->batch * g->cfg.n_embed * sizeof(float),
                          is_device_ptr(x_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    ZG_HIP(hipStreamSynchronize(s));
    return check_fault(g);
}

// A generation in three parts, so that several handles' generations can be fed to their streams turn by turn
// (zg_gpt_generate_enqueue_many): gen_begin — prompts, cache clearing, the whole-prompt pass, the prefetcher's start;
// gen_pump — ONE graph launch (graph_steps decode steps) or one single step, false when nothing is left; gen_end — the
// prefetcher's stop word and the record of the last pick.  After a successful gen_begin, gen_end must run (also on failure:
// the prefetcher must not wait for steps that never come).
static int gen_begin(zg_gpt* g, const size_t* prompts, size_t prompt_stride, const size_t* prompt_lens, size_t n_steps, bool sampled = false,
                     float temp = 1.0f, uint64_t seed = 0) {
    ZG_REQUIRE(g && prompts && prompt_lens, ZG_ERR_ARG, "generate: null argument");
    const size_t C = g->cfg.context_size, V = g->cfg.vocab_size, B = g->batch;
    ZG_REQUIRE(n_steps >= 1 && n_steps <= C, ZG_ERR_SHAPE, "generate: n_steps %zu outside 1..%zu", n_steps, C);
    hipStream_t s = gs(g);
    ZG_HIP(hipStreamSynchronize(s));  // pinned staging below is shared with earlier calls
    size_t min_prompt = C;
    memset(g->h_ints, 0, (B * C + B) * sizeof(int));
    for (size_t b = 0; b < B; ++b) {
        const size_t np = prompt_lens[b];
        ZG_REQUIRE(np >= 1 && np <= C && np <= prompt_stride, ZG_ERR_SHAPE, "generate: prompt %zu has length %zu", b, np);
        for (size_t i = 0; i < np; ++i) {
            const size_t t = prompts[b * prompt_stride + i];
            ZG_REQUIRE(t < V, ZG_ERR_SHAPE, "generate: token %zu >= vocab %zu", t, V);
            g->h_ints[b * C + i] = (int)t;
        }
        g->h_ints[B * C + b] = (int)np;
        if (np < min_prompt) min_prompt = np;
    }
    // The positions every sequence has a prompt token for go through the Blocks together (prefill); the
    // rest of the loop is main.zig:330-338 one position at a time.
    size_t first = 0;
    if (g->pf_x != nullptr && min_prompt >= prefill_min())
        first = min_prompt < n_steps ? min_prompt : n_steps;
    g->h_ctrl->step = (int)first;
    g->h_ctrl->seq_len = (int)first;
    g->h_ctrl->mode = sampled ? 2 : 0;
    g->h_ctrl->n_partials = g->lm_grid;
    g->gen_sampled = sampled;
    if (sampled) {
        g->h_samp->inv_temp = 1.0f / temp;
        g->h_samp->pad = 0;
        g->h_samp->seed = seed;
        ZG_HIP(hipMemcpyAsync(g->samp, g->h_samp, sizeof(SampleParams), hipMemcpyHostToDevice, s));
    }
    ZG_HIP(hipMemcpyAsync(g->prompt, g->h_ints, B * C * sizeof(int), hipMemcpyHostToDevice, s));
    ZG_HIP(hipMemcpyAsync(g->prompt_len, g->h_ints + B * C, B * sizeof(int), hipMemcpyHostToDevice, s));
    ZG_HIP(hipMemcpyAsync(g->ctrl, g->h_ctrl, sizeof(StepCtrl), hipMemcpyHostToDevice, s));
    ZG_TRY(clear_kv(g, s, first));
    ZG_TRY(ensure_ln_folded(g, s));
    if (first > 0) {
        ZG_HIP(hipMemcpyAsync(g->out_tokens, g->prompt, B * C * sizeof(int), hipMemcpyDeviceToDevice, s));
        ZG_TRY(enqueue_prefill(g, first, false, s));
    }
    if (!(g->flags & ZG_GPT_NO_GRAPH) && s != nullptr && g->graph_stream != s) ZG_TRY(capture_all(g, s));  // before the prefetcher starts its idle clock
    ZG_TRY(note_steps(g, n_steps, s));
    ZG_TRY(pf_start(g, n_steps, s));
    g->gen_pos = first;
    g->gen_n = n_steps;
    g->gen_min_prompt = min_prompt;
    g->gen_since_sync = 0;
    g->gen_open = true;
    return ZG_OK;
}

static int gen_pump(zg_gpt* g, bool* more) {
    hipStream_t s = gs(g);
    const size_t C = g->cfg.context_size, n_steps = g->gen_n, st = g->gen_pos;
    *more = false;
    if (st >= n_steps) return ZG_OK;
    const size_t K = ((g->flags & ZG_GPT_NO_GRAPH) || s == nullptr) ? 1 : g->graph_steps;
    // ZGPT2_SYNC_EVERY=n (profiling only): drain the stream every n steps — rocprofv3's counter collection has crashed
    // on this stack when tens of thousands of dispatches were queued ahead of it
    static const int sync_every = env_int("ZGPT2_SYNC_EVERY", 0);
    if (sync_every > 0 && ++g->gen_since_sync >= (size_t)sync_every) {
        g->gen_since_sync = 0;
        (void)hipStreamSynchronize(s);
    }
    if (K > 1 && st >= g->gen_min_prompt && st % K == 0 && st + K <= n_steps && st + K <= C) {
        if (g->graph_stream != s) ZG_TRY(capture_all(g, s));
        const size_t b = st / 64;  // sequence lengths st + 1 .. st + K share a bucket (K divides 64)
        if (g->gen_sampled) {
            ZG_TRY(capture_sampled(g, b, true, s));
            ZG_HIP(hipGraphLaunch(g->graphs_ks[b], s));
        } else {
            ZG_TRY(capture_multi(g, b, s));
            ZG_HIP(hipGraphLaunch(g->graphs_k[b], s));
        }
        g->gen_pos = st + K;
    } else {
        ZG_TRY(run_step(g, st >= g->gen_min_prompt, st + 1, s, g->gen_sampled));
        g->gen_pos = st + 1;
    }
    *more = g->gen_pos < n_steps;
    return ZG_OK;
}

static int gen_end(zg_gpt* g, int rs) {
    hipStream_t s = gs(g);
    g->gen_open = false;
    ZG_TRY(pf_stop(g, s));  // also after a failed launch: the prefetcher must not wait for steps that never come
    ZG_TRY(rs);
    ZG_TRY(launch_embed_step(embed_args(g, 1), s));  // record the pick of the last step
    g->steps_enqueued = g->gen_n;
    return ZG_OK;
}

int zg_gpt_generate_enqueue(zg_gpt* g, const size_t* prompts, size_t prompt_stride, const size_t* prompt_lens,
                            size_t n_steps) {
    ZG_TRY(require_init());
    ZG_TRY(gen_begin(g, prompts, prompt_stride, prompt_lens, n_steps));
    int rs = ZG_OK;
    for (bool more = true; more && rs == ZG_OK;) rs = gen_pump(g, &more);
    return gen_end(g, rs);
}

// generate (src/main.zig:322-342) AS THE REFERENCE RUNS IT: every token behind the prompt is drawn by GPT.sample
// (main.zig:198-207, :336-338) — softmax(logits / temp), then the first index whose running sum exceeds u x total — with the
// whole loop on the device: the sampler is a node of the captured decode step (sample_step_kernel) and the next step's embed
// kernel feeds its draw.  The uniform of (sequence b, position T) is the counter PRNG of (seed, T, b) that zg_gpt_sample uses when
// it is given no uniforms, so this call returns exactly the tokens of a host loop `tok = zg_gpt_sample(g, T, tok, temp, NULL,
// seed, ...)` — without a host round trip per token.  Results through zg_gpt_generate_fetch.
int zg_gpt_generate_sample_enqueue(zg_gpt* g, const size_t* prompts, size_t prompt_stride, const size_t* prompt_lens, size_t n_steps,
                                   float temp, uint64_t seed) {
    ZG_TRY(require_init());
    ZG_REQUIRE(temp > 0.0f, ZG_ERR_ARG, "generate_sample: temperature %f", temp);
    ZG_TRY(gen_begin(g, prompts, prompt_stride, prompt_lens, n_steps, true, temp, seed));
    int rs = ZG_OK;
    for (bool more = true; more && rs == ZG_OK;) rs = gen_pump(g, &more);
    return gen_end(g, rs);
}

int zg_gpt_generate_sample(zg_gpt* g, const size_t* prompts, size_t prompt_stride, const size_t* prompt_lens, size_t n_steps, float temp,
                           uint64_t seed, size_t* out_tokens, size_t out_len) {
    ZG_REQUIRE(g && out_tokens && out_len >= g->batch * n_steps, ZG_ERR_SHAPE, "generate_sample: out_tokens too short");
    ZG_TRY(zg_gpt_generate_sample_enqueue(g, prompts, prompt_stride, prompt_lens, n_steps, temp, seed));
    return zg_gpt_generate_fetch(g, n_steps, out_tokens, out_len);
}

// generate (src/main.zig:322-342) for the prompts of SEVERAL handles at once: independent sequences need not run in lock step
// (the reference's batch restriction, ops.zig:126-128, lifted the other way) — every handle decodes its own prompts on its own
// stream (zg_gpt_create_ex: own_stream) and the chip overlaps the chains, each of which leaves it idle across every one of its
// launch boundaries.  The handles' graph launches are enqueued turn by turn: a hardware queue holds a fraction of a
// generation's dispatches, and a host that fed one handle to the end first would block on that queue while the others idle.
int zg_gpt_generate_enqueue_many(zg_gpt* const* handles, size_t n_handles, const size_t* prompts, size_t prompt_stride,
                                 const size_t* prompt_lens, size_t n_steps) {
    ZG_TRY(require_init());
    ZG_REQUIRE(handles && n_handles >= 1 && n_handles <= 64 && prompts && prompt_lens, ZG_ERR_ARG, "generate_many: bad argument");
    for (size_t i = 0; i < n_handles; ++i) {
        ZG_REQUIRE(handles[i] != nullptr, ZG_ERR_ARG, "generate_many: handle %zu is null", i);
        for (size_t j = 0; j < i; ++j) {
            ZG_REQUIRE(handles[i] != handles[j], ZG_ERR_ARG, "generate_many: handle %zu is handle %zu", i, j);
            ZG_REQUIRE(n_handles == 1 || gs(handles[i]) != gs(handles[j]), ZG_ERR_ARG,
                       "generate_many: handles %zu and %zu share a stream (create them with own_stream)", j, i);
        }
    }
    size_t begun = 0, row = 0;
    int rs = ZG_OK;
    for (; begun < n_handles && rs == ZG_OK; ++begun) {
        rs = gen_begin(handles[begun], prompts + row * prompt_stride, prompt_stride, prompt_lens + row, n_steps);
        if (rs != ZG_OK) break;  // (its message is picked up below)
        row += handles[begun]->batch;
    }
    char msg[512] = "";
    static const int feeders = env_int("ZGPT2_MANY_THREADS", 1);
    if (rs == ZG_OK && feeders && begun > 1) {
        // one feeder thread per handle: a hipGraphLaunch returns only when its hardware queue has room for the graph's
        // dispatches, so one thread feeding all queues in turn stands still whenever the slowest chain's queue is full
        std::vector<std::thread> th;
        std::vector<int> res(begun, ZG_OK);
        std::vector<std::string> errs(begun);
        const int dev = ctx().device;
        for (size_t i = 0; i < begun; ++i)
            th.emplace_back([&, i]() {
                int r = hipSetDevice(dev) == hipSuccess ? ZG_OK : ZG_ERR_HIP;
                for (bool more = true; more && r == ZG_OK;) r = gen_pump(handles[i], &more);
                res[i] = r;
                if (r != ZG_OK) errs[i] = zg_last_error();  // (the message is thread-local)
            });
        for (auto& t : th) t.join();
        for (size_t i = 0; i < begun && rs == ZG_OK; ++i)
            if (res[i] != ZG_OK) {
                rs = res[i];
                snprintf(msg, sizeof msg, "%s", errs[i].c_str());
            }
    } else {
        bool any = rs == ZG_OK;
        while (any && rs == ZG_OK) {
            any = false;
            for (size_t i = 0; i < begun && rs == ZG_OK; ++i) {
                bool more = false;
                if (handles[i]->gen_pos < handles[i]->gen_n) rs = gen_pump(handles[i], &more);
                any |= more;
            }
        }
        if (rs != ZG_OK) snprintf(msg, sizeof msg, "%s", zg_last_error());
    }
    int first_err = rs;
    for (size_t i = 0; i < begun; ++i) {
        const int e = gen_end(handles[i], rs);
        if (first_err == ZG_OK && e != ZG_OK) {
            first_err = e;
            snprintf(msg, sizeof msg, "%s", zg_last_error());
        }
    }
    if (first_err != ZG_OK) set_error("%s", msg);
    return first_err;
}

int zg_gpt_generate_fetch_many(zg_gpt* const* handles, size_t n_handles, size_t n_steps, size_t* out_tokens, size_t out_len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(handles && n_handles >= 1 && out_tokens, ZG_ERR_ARG, "generate_fetch_many: null argument");
    size_t rows = 0;
    for (size_t i = 0; i < n_handles; ++i) {
        ZG_REQUIRE(handles[i] != nullptr, ZG_ERR_ARG, "generate_fetch_many: handle %zu is null", i);
        rows += handles[i]->batch;
    }
    ZG_REQUIRE(out_len >= rows * n_steps, ZG_ERR_SHAPE, "generate_fetch_many: out_tokens too short");
    size_t row = 0;
    int first_err = ZG_OK;
    char msg[512];
    for (size_t i = 0; i < n_handles; ++i) {  // every handle is drained, also behind a failed one
        const int e = zg_gpt_generate_fetch(handles[i], n_steps, out_tokens + row * n_steps, handles[i]->batch * n_steps);
        if (first_err == ZG_OK && e != ZG_OK) {
            first_err = e;
            snprintf(msg, sizeof msg, "%s", zg_last_error());
        }
        row += handles[i]->batch;
    }
    if (first_err != ZG_OK) set_error("%s", msg);
    return first_err;
}

int zg_gpt_generate_fetch(zg_gpt* g, size_t n_steps, size_t* out_tokens, size_t out_len) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && out_tokens, ZG_ERR_ARG, "generate_fetch: null argument");
    const size_t C = g->cfg.context_size, B = g->batch;
    ZG_REQUIRE(n_steps <= C && out_len >= B * n_steps, ZG_ERR_SHAPE, "generate_fetch: out_tokens too short");
    hipStream_t s = gs(g);
    ZG_HIP(hipMemcpyAsync(g->h_ints, g->out_tokens, B * C * sizeof(int), hipMemcpyDeviceToHost, s));
    ZG_HIP(hipStreamSynchronize(s));
    ZG_TRY(check_fault(g));
    for (size_t b = 0; b < B; ++b)
        for (size_t i = 0; i < n_steps; ++i) out_tokens[b * n_steps + i] = (size_t)g->h_ints[b * C + i];
    return ZG_OK;
}

int zg_gpt_generate_greedy(zg_gpt* g, const size_t* prompts, size_t prompt_stride, const size_t* prompt_lens,
                           size_t n_steps, size_t* out_tokens, size_t out_len) {
    ZG_REQUIRE(g && out_tokens && out_len >= g->batch * n_steps, ZG_ERR_SHAPE, "generate: out_tokens too short");
    ZG_TRY(zg_gpt_generate_enqueue(g, prompts, prompt_stride, prompt_lens, n_steps));
    return zg_gpt_generate_fetch(g, n_steps, out_tokens, out_len);
}

int zg_gpt_profile_step(zg_gpt* g, size_t seq_len, int iters, float* us_out, size_t n_out) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && us_out && n_out >= 8 && iters > 0, ZG_ERR_ARG, "profile_step: bad argument");
    ZG_REQUIRE(seq_len >= 1 && seq_len + (size_t)iters - 1 <= g->cfg.context_size, ZG_ERR_SHAPE,
               "profile_step: positions %zu..%zu outside the context", seq_len, seq_len + iters - 1);
    hipStream_t s = gs(g);
    ZG_HIP(hipStreamSynchronize(s));
    ZG_TRY(ensure_ln_folded(g, s));
    ZG_TRY(note_steps(g, (size_t)iters, s));
    for (size_t b = 0; b < g->batch; ++b) g->h_ints[b] = (int)(b % g->cfg.vocab_size);
    g->h_ctrl->step = (int)seq_len - 1;
    g->h_ctrl->seq_len = (int)seq_len;
    g->h_ctrl->mode = 1;
    g->h_ctrl->n_partials = g->lm_grid;
    ZG_HIP(hipMemcpyAsync(g->forced, g->h_ints, g->batch * sizeof(int), hipMemcpyHostToDevice, s));
    ZG_HIP(hipMemcpyAsync(g->ctrl, g->h_ctrl, sizeof(StepCtrl), hipMemcpyHostToDevice, s));
    static StepProf prof;  // events are created on first use and reused
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // All iterations are enqueued before the single synchronisation so that the queue stays ahead of
    // the GPU (an interval then is kernel + launch boundary, not kernel + host launch latency).
    prof.n = 0;
    double null_us = 0.0;
    for (int it = 0; it < iters; ++it) {
        ZG_TRY(enqueue_step(g, true, bucket_t_hi(g, seq_len + it), s, &prof));  // eager; the embed kernel advances the position
        // calibration: a one-element copy recorded the same way = launch boundary + event overhead
        ZG_TRY(launch_copy_f32(g->q, g->q + 4, 1, s));
        ZG_TRY(prof_mark(&prof, 8, s));
    }
    ZG_HIP(hipStreamSynchronize(s));
    for (size_t i = 1; i < prof.n; ++i) {
        if (prof.cls[i] < 0) continue;  // interval between two steps
        float ms = 0.0f;
        ZG_HIP(hipEventElapsedTime(&ms, prof.ev[i - 1], prof.ev[i]));
        if (prof.cls[i] == 8) {
            null_us += ms * 1000.0;
            continue;
        }
        acc[prof.cls[i]] += ms * 1000.0;
        acc[7] += ms * 1000.0;
    }
    for (int i = 0; i < 8; ++i) us_out[i] = (float)(acc[i] / iters);
    if (n_out >= 9) us_out[8] = (float)(null_us / iters);
    return check_fault(g);  // (every event above was synchronised: timings of a faulted step are not reported)
}

int zg_gpt_time_kernel(zg_gpt* g, int which_and_options, int iters, float* avg_us, size_t* algorithmic_bytes) {
    ZG_TRY(require_init());
    const int which = which_and_options & 0xff;
    const bool cycle = (which_and_options & ZG_TIME_WALK_LAYERS) != 0;  // walk the layers, so that no launch finds its weights in the L2s
    const size_t t_opt = (size_t)((unsigned)which_and_options >> 16);    // another position for the attention kernel (0: mid-context)
    // (ZG_TIME_AT(t) occupies bits 16..30: t up to 32767 — every GPT-2 context; a larger t would set the sign bit)
    ZG_REQUIRE(g && avg_us && iters > 0 && which_and_options >= 0 && which <= 6, ZG_ERR_ARG, "time_kernel: bad argument (ZG_TIME_AT takes t < 32768)");
    hipStream_t s = gs(g);
    ZG_REQUIRE(s != nullptr, ZG_ERR_UNSUPPORTED, "time_kernel needs a capturable stream");
    const size_t E = g->cfg.n_embed, wb = g->wbytes;
    const size_t bytes_tab[7] = {0, 3 * E * E * wb, 0, E * E * wb, 4 * E * E * wb, 4 * E * E * wb, g->cfg.vocab_size * E * wb};
    // control block: a mid-context position so that the attention kernel has work
    size_t T = g->cfg.context_size / 2 > 0 ? g->cfg.context_size / 2 : 1;
    if (t_opt >= 1 && t_opt <= g->cfg.context_size) T = t_opt;
    ZG_HIP(hipStreamSynchronize(s));
    ZG_TRY(ensure_ln_folded(g, s));
    ZG_TRY(note_steps(g, (size_t)iters + 8, s));  // (one epoch per replay of the chain)
    g->h_ctrl->step = (int)T - 1;
    g->h_ctrl->seq_len = (int)T;
    g->h_ctrl->mode = 1;
    g->h_ctrl->n_partials = g->lm_grid;
    ZG_HIP(hipMemcpyAsync(g->ctrl, g->h_ctrl, sizeof(StepCtrl), hipMemcpyHostToDevice, s));
    const int chain = 64;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ZG_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int st = ZG_OK;
    // tagged hand-overs: a new epoch per replay and a launch id per chain position, so that every merging split / slice
    // waits for ITS writers as in a real step (one more tiny launch per 64)
    if (g->tags_on && which != 0) st = launch_epoch_bump(g->epoch, s);
    for (int i = 0; i < chain && st == ZG_OK; ++i)
        st = enqueue_step(g, true, bucket_t_hi(g, T), s, nullptr, which, cycle ? (size_t)i % g->cfg.n_layer : 0, nullptr, i);
    hipError_t ce = hipStreamEndCapture(s, &graph);
    if (st != ZG_OK) return st;
    ZG_HIP(ce);
    ZG_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    ZG_HIP(hipEventCreate(&e0));
    ZG_HIP(hipEventCreate(&e1));
    ZG_HIP(hipGraphLaunch(exec, s));
    ZG_HIP(hipStreamSynchronize(s));
    const int reps = (iters + chain - 1) / chain;
    ZG_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) ZG_HIP(hipGraphLaunch(exec, s));
    ZG_HIP(hipEventRecord(e1, s));
    ZG_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    ZG_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    *avg_us = ms * 1000.0f / (float)(reps * chain);
    if (algorithmic_bytes) *algorithmic_bytes = bytes_tab[which];
    return check_fault(g);
}

int zg_debug_prefetch_stats(zg_gpt* g, unsigned* out, size_t n_out) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g && out && n_out >= 25, ZG_ERR_ARG, "prefetch_stats: need 25 words");
    memset(out, 0, n_out * sizeof(unsigned));
    out[0] = g->pf_on ? (g->pf_stalled ? 2u : 1u) : 0u;
    if (!g->pf_on) return ZG_OK;
    ZG_HIP(hipStreamSynchronize(gs(g)));
    ZG_HIP(hipStreamSynchronize(g->pf_stream));
    PfCtl h;
    ZG_HIP(hipMemcpy(&h, g->pf_ctl, sizeof(PfCtl), hipMemcpyDeviceToHost));
    for (int x = 0; x < 8; ++x) {
        out[1 + x] = h.ticket[x];
        out[9 + x] = h.exit_reason[x];
        out[17 + x] = h.jobs_done[x];
    }
    for (size_t i = 0; i < 256 && 25 + i < n_out; ++i) out[25 + i] = h.xcd_log[i];  // diagnostic (-DZG_STAMPS) builds only
    return ZG_OK;
}

#ifdef ZG_STAMPS
// Diagnostic build only: (re)arm the timestamp buffer / read it back (count, then 10 words per record).
int zg_debug_stamps_begin(void) {
    ZG_TRY(require_init());
    Ctx& c = ctx();
    const size_t bytes = (16 + 10 * 4096) * sizeof(unsigned long long);
    if (!c.dbg) ZG_HIP(hipMalloc(reinterpret_cast<void**>(&c.dbg), bytes));
    ZG_HIP(hipMemset(c.dbg, 0, bytes));
    return ZG_OK;
}
int zg_debug_stamps_read(unsigned long long* out, size_t n_words) {
    ZG_TRY(require_init());
    ZG_HIP(hipDeviceSynchronize());
    ZG_HIP(hipMemcpy(out, ctx().dbg, n_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return ZG_OK;
}
#endif

}  // extern "C"
