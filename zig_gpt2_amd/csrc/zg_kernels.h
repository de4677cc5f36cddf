// zg_kernels.h — host-callable launchers of the HIP kernels (internal to libzgpt2_hip).
#pragma once
#include "zg_common.h"

namespace zg {

// Device-side decode control block (lives in the model arena; read by kernels so that one
// captured hipGraph replays for every position).
struct StepCtrl {
    int step;      // 0-based decode step s; sequence length T = s + 1 while a step runs
    int seq_len;   // T of the step in flight (written by the embed kernel)
    int mode;      // 0: generate loop (tokens from prompts / previous argmax), 1: forced tokens
    int n_partials;  // number of lm_head argmax partials per sequence (grid of the lm_head GEMV)
};

// ------------------------------------------------------------------------------------ GEMV
enum GemvPrologue { PRO_NONE = 0, PRO_LAYERNORM = 1, PRO_ATTN_MERGE = 2 };
enum GemvEpilogue { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_GELU = 2, EPI_QKV = 3, EPI_ARGMAX = 4 };
enum WeightType { WT_BF16 = 0, WT_F32 = 1 };

constexpr int kAttnChunk = 256;       // KV positions per attention workgroup (4 waves x 64)
constexpr int kPartStride = 64 + 2;   // floats per attention partial: o[hd<=64], m, l

struct GemvArgs {
    // y[m][n] = epilogue( sum_k prologue(x)[m][k] * W[n][k] + bias[n] ),  m < M, n < N
    const void* W;  // [N][K], bf16 bits or fp32, K contiguous (ops.Linear.weight layout)
    const float* bias;  // [N] or nullptr
    int N, K, M;
    int rows_per_wave;
    int waves_per_wg;  // filled by gemv_plan (M == 1 kernels may run as 1..4-wave workgroups)
    int prologue, epilogue;
    // PRO_NONE / PRO_LAYERNORM input
    const float* x;  // [M][x_stride]
    int x_stride;
    const float* ln_g;
    const float* ln_b;
    float eps;
    // PRO_LAYERNORM, M == 1: the LayerNorm folded out of the dot product (launch_ln_fold): c2[n] = sum_k W g,
    // c3[n] = sum_k W b + bias[n]; null = use the kernel that normalises the input first
    const float* ln_c2;
    const float* ln_c3;
    // PRO_ATTN_MERGE input: partials [M][H][max_splits][kPartStride]
    const float* part;
    int n_heads, head_dim, max_splits;
    const StepCtrl* ctrl;  // seq_len for the merge / KV scatter position
    int t_hi;  // launch-time upper bound of seq_len (same 256-position chunk as seq_len); 0 = read ctrl
    // EPI_STORE / EPI_RESIDUAL / EPI_GELU
    float* y;
    int y_stride;
    const float* resid;  // may alias y
    int resid_stride;
    // EPI_QKV: q [M][E]; caches [M][H][ctx][hd]
    float* q;
    void* k_cache;
    void* v_cache;
    int ctx;
    int kv_mode;   // 0 fp32, 1 fp16, 2 B24 (bf16 plane + byte plane `kv_lo` bytes further on)
    size_t kv_lo;
    // EPI_ARGMAX: optional logits [M][logits_stride]; partials [M][gridDim.x]
    float* logits;
    int logits_stride;
    float* part_val;
    int* part_idx;
    // split-K over workgroups (batched wide Linears): combine workspace [sk_tiles][4][128] floats and FOUR counters per
    // tile (zero between launches: the last arriver resets its counter; the 16-wave kernel uses counter [tile], the
    // four-wave plane-fed kernel one per wave, [4 tile + wave]); kslices is filled by gemv_plan
    float* sk_ws;
    int* sk_cnt;
    int sk_tiles;
    int kslices;
    // lock-step batch on the matrix cores: activation planes in global memory (zg_common.h plane_elem).  pl_in: the
    // input rows arrive as planes written by the previous kernel (PRO_LAYERNORM: planes of g * x, x itself is read for
    // the row statistics only; PRO_NONE: planes of x).  pl_out: this kernel's output rows (after residual / GELU) are
    // also written as planes, scaled by pl_g[n] when given (the next LayerNorm's gain); EPI_GELU with y == nullptr
    // writes only the planes.
    const bf16_t* pl_in;
    bf16_t* pl_out;
    const float* pl_g;
    // K slices of the four-wave plane-fed kernel meet by TAGGED DATA instead of tickets: slices 1.. store (value, tag) as one
    // 8-byte word, slice 0 polls until the tags are this launch's — one memory-side round trip instead of three.
    // tag = *epoch << 8 | launch_id: epoch is advanced by the embed kernel at every step, launch_id (1..255) is unique
    // among the launches of a step that share sk_tag [sk_tiles][4][128].
    const unsigned* epoch;
    unsigned launch_id;
    unsigned long long* sk_tag;
    unsigned* fault;  // set to 1 by a poller whose bounded wait ran out (the host fails the call loudly: api_gpt.hip check_fault)
    unsigned spin_limit;  // polls before a poller gives up (2^20; ZGPT2_TAG_SPIN_LIMIT for the test of the failure path)
    // LayerNorm statistics by tile: a producer of x (four-wave plane-fed kernel, embed kernel) also writes, per 16-column tile
    // and batch row, the sum and the sum of squares of its 16 outputs, st_out [8][N / 16][2]; the LayerNorm-fed consumer
    // adds the tiles (st_in) instead of reading x again — a third of its vector-memory traffic
    float* st_out;
    const float* st_in;
    const float* zero;        // device pointer to a few zero floats (stand-in for absent bias / residual)
    unsigned long long* dbg;  // diagnostic timestamps (only read by -DZG_STAMPS builds)
    unsigned* progress;       // launch counter followed by the side-stream prefetcher (prefetch.hip); null = not counted
};

// Whether launch_gemv can run this M x K at all (batched kernels keep the M input rows in LDS).
bool gemv_supported(const GemvArgs& a, int weight_type);
// Whether this launch may take its input rows as planes in global memory (GemvArgs.pl_in).
bool gemv_planes_ok(const GemvArgs& a, int weight_type);
// ... and whether it can write its output rows as planes (GemvArgs.pl_out).
bool gemv_planes_producer_ok(const GemvArgs& a, int weight_type);
// Whether a planned plane-fed launch runs as the four-wave kernel (the one that writes / reads the tile statistics).
bool gemv_pl4_ok(const GemvArgs& a, int weight_type);
// Fills rows_per_wave and returns the grid size for the given problem.
int gemv_plan(GemvArgs& a, int weight_type = WT_F32);
int gemv_kslices(const GemvArgs& a);
// Rows of W that one workgroup of the planned launch reads (block b: rows b * r .. (b + 1) * r - 1); 0 = another mapping
int gemv_rows_per_wg(const GemvArgs& a, int weight_type);
int launch_ln_fold(const void* W, int weight_type, const float* g, const float* b, const float* bias, int N, int K, float* c2,
                   float* c3, hipStream_t s);
int launch_gemv(const GemvArgs& a, int weight_type, int grid, hipStream_t s);

// ------------------------------------------------------------------------------------ attention
struct AttnArgs {
    const float* q;  // [B][H*hd]
    const void* k;   // element (b,h,t,d) at b*stride_b + h*stride_h + t*stride_t + d
    const void* v;
    long stride_b, stride_h, stride_t;
    int kv_mode;   // 0 fp32, 1 fp16, 2 B24: bf16 plane at k / v, byte plane (same element strides) kv_lo bytes further on
    size_t kv_lo;
    int n_heads, head_dim, batch;
    const StepCtrl* ctrl;  // seq_len read from ctrl->seq_len when non-null
    int seq_len;           // used when ctrl == nullptr
    int t_hi;              // launch-time upper bound of seq_len in the same 64-position bucket (>= seq_len)
    int max_splits;
    float* part;  // [B][H][max_splits][kPartStride]
    unsigned* progress;  // see GemvArgs
    // lock-step batch with activation planes: the last split of (b, h) to arrive merges them and writes the head's output as
    // planes [3][8][H * 64] (plane_elem) for the c_proj Linear; merge_cnt = one zeroed counter per (b, h)
    bf16_t* pl_out;
    int* merge_cnt;
    // ... or, with part_tag [B][H][max_splits][66] given, split 0 merges: the other splits store (value, tag) words and
    // split 0 polls them (see GemvArgs.sk_tag for the tag)
    const unsigned* epoch;
    unsigned launch_id;
    unsigned long long* part_tag;
    unsigned* fault;  // see GemvArgs.fault
    unsigned spin_limit;
};
int launch_attn_decode(const AttnArgs& a, hipStream_t s);
// the op tier's general path for head_dim != 64 (fp32, one workgroup per (sequence, head))
int launch_attn_any_dim(const float* q, const float* k, const float* v, long stride_b, long stride_h, long stride_t, int batch, int n_heads,
                        int head_dim, int seq_len, float* out, hipStream_t s);
// Standalone merge (op tier): out[b][h*hd+d] = sum_s w_s o_s / sum_s w_s l_s
int launch_attn_merge(const float* part, int batch, int n_heads, int head_dim, int max_splits,
                      int seq_len, float* out, hipStream_t s);

// ------------------------------------------------------------------------------------ elementwise
int launch_epoch_bump(unsigned* epoch, hipStream_t s);  // measurement chains: advance the tag epoch (api_gpt.hip zg_gpt_time_kernel)
int launch_layernorm(float* x, int rows, int n, const float* g, const float* b, float eps, hipStream_t s, unsigned* done_flag = nullptr,
                     unsigned done_seq = 0, bool* announced = nullptr, float* shadow = nullptr);
int launch_gelu(float* x, size_t n, hipStream_t s, unsigned* done_flag = nullptr, unsigned done_seq = 0, bool* announced = nullptr, float* shadow = nullptr);
int launch_softmax(float* x, size_t n, hipStream_t s, unsigned* done_flag = nullptr, unsigned done_seq = 0, bool* announced = nullptr);
int launch_done_flag(unsigned* flag, unsigned seq, hipStream_t s);
struct CopySegs {  // up to four device -> pinned-host segments (16-byte aligned, multiples of 4 bytes)
    const void* src[4];
    void* dst[4];
    unsigned bytes[4];
    int n;
};
int launch_copy_out_done(const CopySegs& segs, unsigned* flag, unsigned seq, hipStream_t s);
int launch_copy2_f32(const float* a, float* da, const float* b, float* db, size_t n, hipStream_t s);
int launch_embedding(const float* w, size_t emb_dim, const size_t* idx, size_t n_idx, size_t n_rows,
                     float* out, int* d_oob_flag, hipStream_t s, unsigned* done_flag = nullptr, unsigned done_seq = 0, bool* announced = nullptr);
int launch_split_qkv(const float* in, size_t rows, size_t n_embed, size_t split_idx, float* out, hipStream_t s);
int launch_transpose(const float* in, size_t batch, size_t t, size_t n, size_t h, float* out, hipStream_t s);
int launch_copy_f32(const float* in, float* out, size_t n, hipStream_t s);
int launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
// fp32 [rows][K] -> bf16 [rows][3K] = [hi | mid | lo], hi + mid + lo == x exactly
int launch_split3(const float* in, size_t rows, int K, bf16_t* out, hipStream_t s);
// rows [from_row, ctx) of every (cache, sequence x head) strip set to zero: n_caches caches cache_stride bytes apart, each holding a
// plane of [strips][ctx] rows of row_bytes at plane_off (api_gpt.hip clear_kv: what a new sequence must not inherit)
int launch_kv_clear_tail(void* base, int n_caches, size_t cache_stride, size_t plane_off, int row_bytes, int strips, int ctx, int from_row,
                         hipStream_t s);

// ------------------------------------------------------------------------------------ MFMA GEMM
// C[M,N] = A[M,K] * B[N,K]^T (+ bias) (optional GELU); bf16 operands, fp32 accumulate; C bf16 or fp32.
int launch_gemm_bf16_nt(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                        bool gelu, bool out_bf16, hipStream_t s);
// The same kernel over a list of (A plane, B plane) pairs: A is [M][lda] holding planes of K = 64 kpp columns each
// (plane p at column p K), B is [N][ldb] likewise; C = sum over pairs of A_plane[pa] * B_plane[pb]^T (+ bias).
// fp32 operands split exactly into bf16 hi + mid + lo planes make the bf16 matrix cores deliver fp32-grade
// products (every bf16 x bf16 product is exact in fp32): planes {(2,0),(1,0),(0,0)} for exact-bf16 weights,
// six pairs for arbitrary fp32 weights.  Pair i's planes sit in bits [4i, 4i + 4) of pa_bits / pb_bits.
struct GemmPlanes {
    int lda, ldb;  // row lengths in elements
    int kpp;       // K-steps (of 64) per plane
    int npairs;
    unsigned pa_bits, pb_bits;
    bool b_plane_major;  // B's planes are whole [N][ldb] matrices one behind the other (the model's fp32-weight planes), not column blocks of a row
};
int launch_gemm_planes(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl,
                       int ldc, bool gelu, bool out_bf16, hipStream_t s);
int launch_gemm_s4(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc,
                   bool gelu, bool out_bf16, int bn, hipStream_t s);  // gemm_s4.hip
bool gemm_s4_args_ok(const GemmPlanes& pl, int ldc);            // the bounds of gemm_s4_kernel's packed arguments
// ... and its epilogue kinds for the whole-prompt Linears (gemm_s4.hip; PrefillQkv below)
enum { S4_PLAIN = 0, S4_PARTIAL = 1, S4_QKV = 2, S4_SPLIT3 = 3 };
struct PrefillQkv;
int launch_gemm_s4_prefill(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, int nplanes, int kind, int n_slices,
                           const PrefillQkv* qkv, hipStream_t s);
int gemm_s4_stamps(unsigned long long* out, size_t n_words);  // diagnostic (ZGPT2_GEMM_DBG bit 256)
int gemm_s4_fault(unsigned* out);  // 1: a stream-K consumer of gemm_s4 gave up waiting for its producer since the last call (results wrong)
int gemm_debug_stamps(unsigned long long* out, size_t n_words);  // of the kernel generation launched last
unsigned long long gemm_mfma_launch_count();  // launches of the persistent MFMA GEMMs so far (tests assert the path taken)
void gemm_note_launch();

// ------------------------------------------------------------------------------------ prefill (prefill.hip)
// Whole-prompt forward.  Activations feeding a GEMM are fp32 split exactly into kSplit bf16 terms,
// stored as planes [M][kSplit * K] = [hi | mid | lo].
constexpr int kSplit = 3;
enum PrefillEpilogue { PF_F32 = 0, PF_RESID = 1, PF_GELU_SPLIT = 2, PF_PARTIAL = 3 /* internal: split-K slice */, PF_QKV = 4 };
int launch_embed_prefill(const int* tokens, int token_stride, int B, int P, const void* wte, const void* wpe,
                         int weight_type, int E, float* x, hipStream_t s);
int launch_ln_split(const float* x, int M, int E, const float* g, const float* b, float eps, bf16_t* out, hipStream_t s);
// C = A[M][kSplit K] * W[N][K]^T + bias; PF_F32: fp32 C[M][ldc]; PF_RESID: C += ...; PF_GELU_SPLIT: bf16 C[M][kSplit N] = split(gelu(...))
// ws: fp32 workspace for split-K partial sums (used when the output has too few tiles to fill the chip).
// ln (PF_RESID only, may be null): LayerNorm of the updated rows, written as split planes to ln->out — fused
// into the split-K tail when there is one, a separate launch_ln_split otherwise.
// PF_QKV: fp32 store of the qkv rows (row m = b P + t) AND the cache append of ops.zig:152-157 for the K / V
// columns, into the head-major caches [b][h][ctx][64] (fp32 or fp16).
struct PrefillQkv {
    int P, E, H, ctx, kv_mode;
    void* k_cache;
    void* v_cache;
    size_t kv_lo;  // kv_mode 2: byte offset of the low-mantissa plane
    // stream-K hand-over of the persistent GEMM (gemm_s4.hip, S4_QKV with 1.5 rounds of tiles): workspace for the producers'
    // accumulators (196608 B per shared tile), one flag word per (shared tile, wave), the launch's epoch (flags of earlier
    // launches are smaller); null = whole tiles only
    void* sk_ws = nullptr;
    size_t sk_ws_bytes = 0;
    unsigned* sk_flags = nullptr;
    unsigned sk_flags_words = 0;  // words behind sk_flags (4 per shared tile: G / 2 tiles for G workgroups)
    unsigned sk_epoch = 0;
};
struct PrefillLn {
    const float* g;
    const float* b;
    float eps;
    bf16_t* out;
};
int launch_prefill_gemm(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, int ldc, int epi,
                        float* ws, size_t ws_floats, const PrefillLn* ln, hipStream_t s, const PrefillQkv* qkv = nullptr,
                        int nsplit = kSplit);  // nsplit = 2: multiply the hi + mid planes only (2/3 of the matrix work);
                                               // kWeightPlanes: W is the plane-major three-term split [3][N][K] of an fp32 matrix
constexpr int kWeightPlanes = 33;
void prefill_force_route(int kernel, int slices);  // zg_debug_prefill_route / _linear: pin the GEMM family (1 gemm_s4, 2 the 128-row kernels) and the K slices
// out[M][kSplit E] = split(causal attention of the q / k / v columns of qkv[M][3E]), M = B P rows ordered (b, t)
// (attn_prefill.hip: bf16 matrix cores on exact plane splits; ws = fp32 workspace for the partials of split key ranges)
int launch_attn_prefill(const float* qkv, bf16_t* out, int B, int P, int E, int H, float* ws, size_t ws_floats, const float* k_cache_or_null,
                        const float* v_cache_or_null, int ctx, hipStream_t s, int force_tiles = 0);  // force_tiles: key tiles per workgroup (tests)

// GPT.sample tail: in-place softmax(logits / temp) per sequence + inverse-CDF draw with uniform u[b].
// part_val / n_part / part_stride: the per-workgroup maxima lm_head's argmax epilogue left (the row maximum without a pass over the
// logits); seg_ws: sample_workspace_floats(batch) floats; write_probs: leave softmax(logits / temp) in the logits rows
int launch_sample(float* logits, int batch, int vocab, float temp, const float* u, const float* part_val, int n_part, int part_stride, float* seg_ws,
                  int* token_out, bool write_probs, hipStream_t s);
size_t sample_workspace_floats(int batch);

// Decode-step head kernel: token selection (+ argmax finalisation of the previous step) and
// x = wte[token] + wpe[pos].
struct EmbedArgs {
    // the first 14 dwords are preloaded into SGPRs (zg_common.h ZG_PIN): what the first loads of the kernel depend on
    StepCtrl* ctrl;
    const float* part_val;   // [B][n_partials]
    const int* part_idx;
    int part_stride;
    int n_partials;          // lm_head grid size
    const int* prompt;       // [B][prompt_stride]
    const int* prompt_len;   // [B]
    int batch;
    int prompt_stride;
    // ... the rest is fetched from the kernarg segment under those loads
    const int* forced;       // [B] (mode 1)
    const void* wte;  // [V][E] bf16 or fp32
    const void* wpe;  // [ctx][E]
    int weight_type;
    int n_embed, vocab;
    int* cur_token;          // [B]
    int* out_tokens;         // [B][out_stride]
    int out_stride;
    float* x;                // [B][E]
    bf16_t* pl_out;          // optional planes of pl_g * x for the first Linear of the lock-step batch (GemvArgs.pl_in)
    const float* pl_g;
    unsigned* epoch;         // optional step counter behind the tagged hand-overs (GemvArgs.sk_tag): +1 when a step starts
    float* st_out;           // optional LayerNorm statistics by tile of x (GemvArgs.st_in): [8][n_embed / 16][2]
    int finish_only;         // 1: only record the greedy pick of the last step; 2: argmax -> cur_token
    unsigned* progress;      // set to (T << 8) | 1 (and the XCD of this block beside it) when a step starts; the other decode kernels add 1 each
    const int* sampled;      // [B] (mode 2: generate with the sampler) the token sample_step_kernel drew from the previous step's logits
};
int launch_embed_step(const EmbedArgs& a, hipStream_t s);
// GPT.sample's tail (src/main.zig:200-206) inside the generate loop: temperature and seed live in device memory (one captured
// graph serves every call), the uniform of (sequence b, position T) is the library's counter PRNG of (seed, T, b) — what
// zg_gpt_sample uses when it is given no uniforms
struct SampleParams {
    float inv_temp;
    unsigned pad;
    unsigned long long seed;
};
int launch_sample_step(float* logits, int batch, int vocab, const SampleParams* params, const StepCtrl* ctrl, const float* part_val, int n_part,
                       int part_stride, float* seg_ws, int* token_out, hipStream_t s);

// ------------------------------------------------------------------------------------ multi-GPU (dist.hip)
int dist_broadcast(void* buf, size_t bytes, int root, hipStream_t s);  // in place, over the communicator of zg_dist_init

// ------------------------------------------------------------------------------------ decode prefetcher (prefetch.hip)
enum PfKind { PF_NONE = 0, PF_WEIGHTS = 1, PF_KV = 2 };
constexpr unsigned PF_STOP = 0xffffffffu;  // value of the progress word that ends the prefetcher
// What launch `index` of a decode step reads: weights = n_wg contiguous tiles of wg_bytes (block b reads tile b);
// KV = the rows of earlier positions of one layer's caches, laid out for the (heads, splits, batch) attention grid.
struct PfJob {
    const char* base;
    const char* base2;    // PF_KV: the V cache
    size_t total_bytes;   // PF_WEIGHTS: end of the matrix (the last tile may be short)
    unsigned wg_bytes, n_wg;
    unsigned touch_bytes;  // PF_WEIGHTS: leading part of every tile that is fetched (the whole tile unless the matrix is far larger than the L2s)
    unsigned kind;
    unsigned n_heads, ctx, batch, row_bytes;  // PF_KV
    unsigned cls;          // kernel class of the launch (1 c_attn .. 6 lm_head, as in zg_gpt_profile_step)
    unsigned pad;
    // small per-row / per-column vectors every block reads a few lines of (bias, folded-LayerNorm vectors, ln gain):
    // fetched whole into every XCD — once the weights come from the L2 these are what a block would wait for
    const char* aux[3];
    unsigned aux_bytes[3];
    unsigned pad2;
};
struct alignas(8) PfCtl {
    unsigned progress;    // (T << 8) | launches started in the step at sequence length T;  PF_STOP ends the prefetcher
    unsigned base_xcd;    // 0x100 | XCC_ID of block 0 of the step's launches (block b then runs on XCD (base + b) % 8);
                          // written with `progress` as one 64-bit store by the embed kernel
    unsigned sink_guard, sink;
    unsigned ticket[8];       // prefetcher workgroups that found themselves on XCD x
    unsigned exit_reason[8];  // 1 stop, 2 idle limit
    unsigned jobs_done[8];
    unsigned xcd_log[256];    // -DZG_STAMPS builds: 0x100 | XCC_ID of block 0 of launch i of the last step
};
struct PfArgs {
    PfCtl* ctl;
    const PfJob* jobs;  // [njobs]: index 0 = the embed kernel (nothing to fetch), then the step's launches in order
    int njobs;          // launches of a step with lm_head
    int lead;           // launches ahead of the running one
    int nsub;           // prefetcher workgroups per XCD
    int max_T;          // last sequence length of the generation (nothing is fetched for steps beyond it)
    unsigned idle_limit;
    unsigned sleep;     // s_sleep(8) repetitions between polls
    unsigned cls_mask;  // bit c set: launches of class c are fetched for
    unsigned line_shift;  // log2 of the touch stride in bytes (7 = one load per 128-byte line)
    unsigned cap_bytes;   // most bytes fetched for one launch (the head of every tile when the matrix is larger); 0 = no cap
};
int launch_prefetcher(const PfArgs& a, hipStream_t s);

// Every decode kernel counts itself in at entry (one lane of block 0; fire and forget).
__device__ __forceinline__ void pf_count(unsigned* progress) {
    if (progress != nullptr && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) {
#ifdef ZG_STAMPS
        const unsigned old = __hip_atomic_fetch_add(progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<PfCtl*>(progress)->xcd_log[old & 255u] = 0x100u | (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u);
#else
        __hip_atomic_fetch_add(progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
}

}  // namespace zg
