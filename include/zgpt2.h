/*
 * zgpt2.h — C ABI of libzgpt2_hip.so: the MI355X (gfx950) implementation of zig_gpt2's forward
 * hot path.  This is the drop-in boundary: every entry point below replaces one public decl of
 * the reference's src/ops.zig (op tier) or one method of src/main.zig (model tier); the Zig shim
 * zig_gpt2_amd/zig/ops.zig keeps the reference's decl signatures and forwards each slice as
 * (ptr, len).  Citations are file:line in /root/reference (EugenHotaj/zig_gpt2 @ v1).
 *
 * Conventions
 *   - All functions return 0 on success, a negative zg_status otherwise; zg_last_error() gives
 *     text.  The reference's ops return void and rely on Zig slice bounds checks; the length
 *     checks those imply are done here and reported as ZG_ERR_SHAPE.
 *   - Lengths are ELEMENT counts (Zig slice .len), not bytes.  f32 is IEEE binary32, usize is
 *     size_t (8 bytes).
 *   - Op tier: every pointer may be host memory or device memory (detected with
 *     hipPointerGetAttributes).  Host buffers are staged through arenas that are allocated once
 *     in zg_init — no *_forward call allocates device or host memory (the reference's
 *     "no allocations at runtime" contract, README.md:6, src/main.zig:46-64): a device arena
 *     (ZGPT2_STAGING_MB, 512), a pinned host arena (ZGPT2_PINNED_MB, 32) through which the caller's
 *     pageable buffers travel — small vectors that a kernel touches once are read / written by the
 *     kernel in place across PCIe, the rest moves by one DMA each way — and a device pool for the
 *     mirrors of caller-owned KV caches (ZGPT2_KV_MIRROR_MB, 1024; see zg_attn_forward).  Op-tier
 *     calls are synchronous on return, because the reference's host code reads the buffers next
 *     (src/main.zig:136-145).
 *   - Model tier: weights, KV cache and every scratch buffer live in one device arena created by
 *     zg_gpt_create; zg_gpt_forward/zg_gpt_generate_greedy only launch kernels.
 *   - Single-threaded like the reference: one call at a time per process.
 */
#ifndef ZGPT2_H
#define ZGPT2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZG_OK = 0,
    ZG_ERR_NOT_INITIALIZED = -1,
    ZG_ERR_SHAPE = -2,      /* a slice length does not match the shapes implied by the call */
    ZG_ERR_HIP = -3,        /* a HIP runtime call failed (text in zg_last_error) */
    ZG_ERR_STAGING = -4,    /* host buffers of one call exceed the staging arena */
    ZG_ERR_UNSUPPORTED = -5,
    ZG_ERR_ARG = -6
} zg_status;

/* ------------------------------------------------------------------ runtime ---------------- */

/* Select the device, create the stream and the host-staging arena (default 512 MiB, or
 * ZGPT2_STAGING_MB).  Idempotent. */
int zg_init(int device);
int zg_init_ex(int device, size_t staging_bytes);
int zg_shutdown(void);
const char* zg_last_error(void);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the library's
 * own stream. */
int zg_set_stream(void* hip_stream);
int zg_synchronize(void);
/* Upload a host PARAMETER tensor (Linear / Embedding weight, bias, LayerNorm vectors) once and keep a
 * device mirror keyed by its host address: later op-tier calls that receive the same host pointer WITH THE
 * SAME LENGTH in a parameter position use the mirror instead of re-staging it (weights are borrowed for
 * the life of the arena in the reference, src/main.zig:349-351).  Activation arguments (inputs, idxs,
 * q / k / v) are never looked up, so a freed weight's address may be reused for them.  Re-registering an
 * address replaces its mirror; host weights changed in place after registration must be registered
 * again.  zg_unregister_tensor drops one mirror (the `defer allocator.free(weight)` moment of
 * src/tests.zig) — also the KV-cache mirror zg_attn_forward keeps for that address; zg_unregister_all
 * drops every mirror and empties the KV pool. */
int zg_register_tensor(const float* host_ptr, size_t len);
int zg_unregister_tensor(const float* host_ptr);
int zg_unregister_all(void);

/* ------------------------------------------------------------------ op tier: src/ops.zig --- */

/* Linear.forward — src/ops.zig:21-46.  weight is [out_features, in_features] row-major (the
 * reference's "column major", ops.zig:9); bias may be NULL; batch = inputs_len / in_features
 * (ops.zig:22); outputs [batch, out_features]. */
int zg_linear_forward(size_t in_features, size_t out_features, const float* weight,
                      const float* bias_or_null, const float* inputs, size_t inputs_len,
                      float* outputs, size_t outputs_len);

/* Embedding.forward — src/ops.zig:59-67.  weight [n_rows, emb_dim]; idxs are usize. */
int zg_embedding_forward(size_t emb_dim, const float* weight, size_t weight_len,
                         const size_t* idxs, size_t idxs_len, float* embeddings,
                         size_t embeddings_len);

/* LayerNorm.forward — src/ops.zig:82-104.  In place over inputs_len / n_features rows;
 * std = sqrt(E[x^2] - E[x]^2 + eps) (ops.zig:95). */
int zg_layernorm_forward(size_t n_features, const float* weight, const float* bias, float eps,
                         float* inputs, size_t inputs_len);

/* CausalSelfAttention.forward — src/ops.zig:129-173: one decode step with a caller-owned KV
 * cache, batch 1.  k_cache / v_cache are the caller's [seq_len, n_embed] slices (row t = token
 * t); row seq_len-1 is written by this call (ops.zig:152,157).  _qkv [3E], _q [E], _k/_v
 * [seq_len*E], _attn [seq_len] are the reference's scratch slices; on return _qkv holds the
 * c_attn output and _q the merged heads (ops.zig:171) as in the reference, while _k/_v/_attn
 * (pure scratch whose content no caller reads) are left untouched: the kernel attends over the
 * [T, H, hd] cache in place instead of re-transposing it every step (ops.zig:153,158).
 * HOST caches are mirrored on the device, keyed by the cache's address: a call that continues the
 * sequence the mirror holds (same pointer, same n_embed, seq_len = rows held + 1 — the only pattern
 * src/main.zig:331-338 produces) uploads nothing and returns only row seq_len-1 to the caller's
 * buffer; any other call (first sight, position 1 again, a jump, a repeated position) uploads rows
 * 0..seq_len-2 from the caller's buffer first.  A caller that EDITS rows it already handed over and
 * then continues with the next position must drop the mirror first (zg_unregister_tensor(cache)).
 * The pool is a bump allocator: when the caches of a call do not fit what is left of it, every mirror is dropped and the pool
 * starts over (live caches upload once more at their next call); a cache larger than the whole pool is staged whole on every
 * call, as before. */
int zg_attn_forward(size_t n_heads, size_t n_embed, const float* c_attn_weight,
                    const float* c_attn_bias, const float* c_proj_weight, const float* c_proj_bias,
                    size_t seq_len, const float* inputs, size_t inputs_len, float* k_cache,
                    size_t k_cache_len, float* v_cache, size_t v_cache_len, float* outputs,
                    size_t outputs_len, float* _qkv, size_t _qkv_len, float* _q, size_t _q_len,
                    float* _k, size_t _k_len, float* _v, size_t _v_len, float* _attn,
                    size_t _attn_len);

/* CausalSelfAttention.split_qkv — src/ops.zig:177-196: [B, T, 3E] -> [B, T, E]. */
int zg_split_qkv(size_t n_embed, size_t seq_len, const float* inputs, size_t inputs_len,
                 size_t split_idx, float* outputs, size_t outputs_len);

/* CausalSelfAttention.transpose — src/ops.zig:199-216: (b, t, n, h) -> (b, n, t, h). */
int zg_transpose(size_t seq_len, size_t n_heads, size_t head_dim, const float* inputs,
                 size_t inputs_len, float* outputs, size_t outputs_len);

/* scaled_dot_product_attention — src/ops.zig:249-307.  q [B,H,1,hd], k/v [B,H,T,hd],
 * outputs [B,H,1,hd]; B = k_len / (H*T*hd) (ops.zig:259); _attn_len must be >= seq_len.  Any head_dim up to 2048: 64 (every
 * GPT-2 configuration) runs the split-KV kernels, anything else a general fp32 kernel (one workgroup per sequence and head). */
int zg_scaled_dot_product_attention(const float* q, size_t q_len, const float* k, size_t k_len,
                                    const float* v, size_t v_len, size_t n_heads, size_t seq_len,
                                    size_t head_dim, float* outputs, size_t outputs_len,
                                    float* _attn, size_t _attn_len);

/* gelu — src/ops.zig:221-228, in place. */
int zg_gelu(float* inputs, size_t inputs_len);

/* softmax — src/ops.zig:231-241, in place; the whole slice is one vector. */
int zg_softmax(float* inputs, size_t inputs_len);

/* Batched-regime Linear on the matrix cores (MFMA): C[M,N] = A[M,K] * B[N,K]^T (+ bias) with an
 * optional fused GELU — the same contraction as Linear.forward's cblas_sgemm(NoTrans, Trans)
 * (src/ops.zig:30-45) for large batch (prefill), with both operands in ops.Linear's K-contiguous
 * layouts.  A, B are bf16 bit patterns and C is bf16 (out_bf16 != 0) or fp32, all DEVICE pointers;
 * any M; N a multiple of 8 for bf16 C, any N for fp32 C (an N that is not a multiple of 4 is stored by the four-wave kernel
 * only, whose packed arguments end at K = 16320: beyond that ZG_ERR_UNSUPPORTED — zg_linear_forward, which has no such limit,
 * takes such a Linear through its GEMV kernels); K a multiple of 64 and at least 128.  Asynchronous on
 * the library stream.  zg_linear_forward itself takes this path for batch >= 16 (fp32 operands split
 * exactly into bf16 planes, so the result stays fp32-sgemm grade).
 * zg_f32_to_bf16 converts a device or host fp32 array into a device bf16 array (round to nearest even). */
int zg_gemm_bf16_nt(const uint16_t* A, const uint16_t* B, const float* bias_or_null, void* C, size_t M,
                    size_t N, size_t K, int gelu, int out_bf16);
/* Diagnostic: number of matrix-core GEMM launches so far in this process (tests assert which path a
 * Linear took; no reference counterpart). */
unsigned long long zg_debug_gemm_launches(void);
/* Diagnostic: shader-clock stamps {count, start, end} of workgroup 0 / wave 0 of the last GEMM launched with
 * ZGPT2_GEMM_DBG bit 256 (tools/microbench/gemm_bench.cpp: cycles vs wall time = the clock the chip ran at). */
int zg_debug_gemm_stamps(unsigned long long* out, size_t n_words);
/* Diagnostic / test entry: ONE whole-prompt Linear exactly as zg_gpt_prefill launches it (src/ops.zig:21-46 for M = batch x
 * prompt rows).  A = the activation planes [M][3 K] bf16 (hi | mid | lo: the exact split of the fp32 rows), W = bf16 weights
 * [N][K], all device pointers.  epilogue: 0 fp32 C[M][N] = A W^T + bias; 1 C[M][N] += ... (residual add; ws required);
 * 2 bf16 planes C[M][3 N] of gelu(...).  force_kernel: 0 = the library's choice, 1 = the persistent four-wave GEMM
 * (gemm_s4.hip; slices = its K slices for epilogue 1, 0 = chosen), 2 = the 128-row prompt GEMM (prefill.hip).  ws: fp32
 * workspace of ws_floats elements for partial slabs.  The cache-append epilogue is exercised through zg_gpt_prefill. */
int zg_debug_prefill_linear(const uint16_t* A_planes, const uint16_t* W, const float* bias_or_null, void* C, size_t M, size_t N,
                            size_t K, int epilogue, int force_kernel, int slices, float* ws, size_t ws_floats);
/* Diagnostic / test entry: the causal prompt attention of zg_gpt_prefill alone (scaled_dot_product_attention of src/ops.zig:249-307
 * for all n_tokens positions of `batch` sequences at once).  qkv: fp32 [batch n_tokens][3 n_embed] rows (q | k | v columns);
 * out: bf16 planes [batch n_tokens][3 n_embed] = hi | mid | lo of the attention output; k_cache / v_cache: NULL (K and V are the
 * qkv columns) or head-major fp32 caches [batch][heads][ctx][64] holding the same rows; ws: fp32 workspace for the partials of
 * split key ranges (may be NULL: whole rows per workgroup); key_tiles: key tiles of 32 per workgroup, 0 = the library's choice.
 * Device pointers. */
int zg_debug_attn_prefill(const float* qkv, uint16_t* out, size_t batch, size_t n_tokens, size_t n_embed, size_t n_heads,
                          const float* k_cache, const float* v_cache, size_t ctx, float* ws, size_t ws_floats, int key_tiles);
/* Test hook: pin the route of every whole-prompt Linear of this process until called again with (0, 0) — force_kernel / slices
 * as in zg_debug_prefill_linear; force_kernel >= 16: the library's rule with that many 256 x 192 tiles (x K slices) as the
 * threshold from which the persistent GEMM takes a Linear (default 192; tools/experiments/pf_route_ab.py).
 * tests/test_prefill_gpu.py runs small models through the persistent GEMM's epilogues with it. */
int zg_debug_prefill_route(int force_kernel, int slices);
/* Diagnostic: name of the kernel instantiation the last decode-kernel launcher of this thread picked (launches recorded
 * into a graph count; bench.py reports it as the symbol of the roofline kernel). */
int zg_debug_last_kernel(char* out, size_t n);
int zg_f32_to_bf16(const float* src, uint16_t* dst_device, size_t len);

/* ------------------------------------------------------------------ model tier: src/main.zig */

/* GPTConfig — src/main.zig:5-23, field for field. */
typedef struct {
    size_t vocab_size;
    size_t context_size;
    size_t n_layer;
    size_t n_heads;
    size_t n_embed;
} zg_gpt_config;

typedef struct zg_gpt zg_gpt;

enum {
    ZG_GPT_WEIGHTS_BF16 = 0,     /* Linear/embedding matrices stored bf16 (RNE) — default */
    ZG_GPT_WEIGHTS_F32 = 1 << 0, /* keep matrices in fp32 (exact with arbitrary checkpoints) */
    ZG_GPT_NO_GRAPH = 1 << 1,    /* launch kernels eagerly instead of replaying a hipGraph */
    ZG_GPT_KV_F16 = 1 << 2,      /* store the KV cache as fp16 instead of fp32 */
    ZG_GPT_NO_PREFILL = 1 << 3,  /* generate: feed prompts one position at a time, as main.zig:331-334 does */
    ZG_GPT_PREFILL_2PLANE = 1 << 4, /* whole-prompt GEMMs multiply two bf16 planes of the fp32 activations instead of the exact
                                      three: 2/3 of the matrix work, ~2e-5 of the logit scale (inside the 1e-3 parity bound,
                                      outside the tests' near-zero floor); bf16-weight handles only */
    ZG_GPT_NO_PREFETCH = 1 << 5, /* zg_gpt_generate_*: no side-stream L2 prefetcher beside the decode chain (results are
                                    identical either way; a measurement switch) */
    ZG_GPT_SAMPLED_GENERATE = 1 << 7, /* capture the decode graphs of zg_gpt_generate_sample_* at create as well (otherwise they are
                                      captured by the first sampled generation: the one place a generate call may allocate) */
    ZG_GPT_KV_B24 = 1 << 6       /* store the KV cache as 24-bit floats (the fp32 value rounded to 16 mantissa bits, kept as a
                                    bf16 plane + a plane of 8 more mantissa bits): 3/4 of the fp32 cache's traffic, 2^-17 per
                                    cached element — inside the 1e-3 parity bound at full context, unlike ZG_GPT_KV_F16 (which
                                    it excludes) */
};

/* Per-block tensor slots (load_block, src/main.zig:271-302) and top-level slots (load_gpt,
 * src/main.zig:304-314).  Linear weights are [out, in] like ops.Linear.weight. */
enum {
    ZG_LN_1_G = 0, ZG_LN_1_B, ZG_C_ATTN_W, ZG_C_ATTN_B, ZG_C_PROJ_W, ZG_C_PROJ_B,
    ZG_LN_2_G, ZG_LN_2_B, ZG_C_FC_W, ZG_C_FC_B, ZG_MLP_PROJ_W, ZG_MLP_PROJ_B, ZG_N_BLOCK_SLOTS
};
enum { ZG_WTE = 0, ZG_WPE, ZG_LN_F_G, ZG_LN_F_B, ZG_N_TOP_SLOTS };

/* State.init + load_gpt's allocations (src/main.zig:46-64, :298-299): one device arena holding
 * weights, `batch` private KV caches and all scratch.  batch = number of independent prompts
 * decoded in lock step (the reference supports 1, src/ops.zig:126-128). */
int zg_gpt_create(zg_gpt** out, const zg_gpt_config* config, size_t batch, unsigned flags);
int zg_gpt_destroy(zg_gpt* g);
/* Independent prompt GROUPS on one GPU (no reference counterpart: the reference decodes one sequence, src/ops.zig:126-128).
 * A decode step is a chain of dependent launches that leaves the chip idle across every launch boundary, and prompts are
 * independent units — so instead of one handle of 8 sequences in lock step, G handles of 8/G sequences each run their chains
 * side by side: own_stream gives a handle a private stream (stream_priority: 0 normal, > 0 high, < 0 low — streams of
 * different priorities never share a hardware queue), share_weights_with lets it read the weight region of another handle of
 * the same config and weight flags instead of holding a copy (that handle loads / broadcasts the weights and must be destroyed
 * last).  A handle with a private stream runs without the side-stream L2 prefetcher unless ZGPT2_PREFETCH=1 forces it (the
 * prefetcher's placement assumes one chain on the chip).  zg_gpt_create(...) == zg_gpt_create_ex(..., NULL). */
typedef struct {
    zg_gpt* share_weights_with; /* NULL: own weight region */
    int own_stream;             /* 0: the library stream (zg_set_stream applies); 1: a private stream made here */
    int stream_priority;
} zg_gpt_options;
int zg_gpt_create_ex(zg_gpt** out, const zg_gpt_config* config, size_t batch, unsigned flags, const zg_gpt_options* options_or_null);
/* The hipStream_t a handle launches on (its private stream, or the library stream of the moment). */
int zg_gpt_stream(zg_gpt* g, void** hip_stream_out);

/* Weight upload (replaces load_linear/load_layer_norm/load_embedding, src/main.zig:210-269).
 * src is fp32, host or device; matrices are converted to the arena's storage type on device. */
int zg_gpt_load_block_tensor(zg_gpt* g, size_t layer, int slot, const float* src, size_t len);
int zg_gpt_load_tensor(zg_gpt* g, int slot, const float* src, size_t len);

/* The weight region of the arena (for an RCCL broadcast to the other GPUs of a node). */
int zg_gpt_weight_arena(zg_gpt* g, void** device_ptr, size_t* bytes);
/* Multi-GPU (no reference counterpart — the reference is single-device): prompts are independent units, sharded
 * over ONE PROCESS PER GPU; the weights are replicated by one RCCL broadcast over xGMI, nothing is exchanged while tokens are
 * generated.  The host program (src/main.zig's role) starts one process per GPU; rank 0 makes the id, hands its 128 bytes to
 * the other ranks by any means (file, pipe, environment), every rank calls zg_init(its device) and zg_dist_init; rank 0 loads
 * the weights (zg_gpt_load_*), every rank calls zg_gpt_broadcast_weights on its handle of the same config and flags, then
 * generates its own prompts.  zg_dist_allgather collects equal-sized device buffers (the ranks' token matrices) in rank
 * order.  RCCL is bound at run time: on a box without librccl.so these return ZG_ERR_UNSUPPORTED, everything else works. */
#define ZG_DIST_ID_BYTES 128
/* Not a collective: ZG_OK when RCCL can be bound in this process.  zg_dist_init is a rendezvous — every rank must enter it or
 * none — so a launcher lets its ranks agree on this answer first (bench.py does, over torch.distributed). */
int zg_dist_available(void);
int zg_dist_unique_id(void* id_out, size_t id_bytes);
int zg_dist_init(const void* id, size_t id_bytes, int rank, int world_size);
int zg_dist_world(int* rank, int* world_size); /* world_size 0: no communicator */
int zg_gpt_broadcast_weights(zg_gpt* g, int root, float* ms_out_or_null);
int zg_dist_allgather(const void* send_device, void* recv_device, size_t bytes_per_rank);
int zg_dist_finalize(void);

/* Bytes one decode step must read at sequence length T: weights + KV. */
int zg_gpt_step_bytes(zg_gpt* g, size_t seq_len, size_t* weight_bytes, size_t* kv_bytes);

/* GPT.forward — src/main.zig:178-195, for all `batch` sequences at once: tokens[b] is fed at
 * position seq_len-1.  If logits_out != NULL (host or device, [batch, vocab]) it receives
 * state.logits and the call is synchronous; compute_logits == 0 skips lm_head (main.zig:192).
 * seq_len == 1 starts a new sequence: the KV caches are cleared first (so are they by zg_gpt_prefill and zg_gpt_generate_*) —
 * the reference's State is zero-initialised once and never reads a row it has not written; here the decode attention reads
 * the rows of its whole 64-position bucket with weight 0, and a NaN left behind by an earlier sequence must not survive. */
int zg_gpt_forward(zg_gpt* g, size_t seq_len, const size_t* tokens, size_t n_tokens,
                   int compute_logits, float* logits_out, size_t logits_len);
/* The prompt loop of generate (src/main.zig:331-334: gpt.forward(i + 1, prompt[i], ...) for every prompt
 * position) as ONE pass: positions 0..n_tokens-1 of all `batch` sequences go through each Block together
 * (Linears as matrix-core GEMMs, causal attention), filling the KV caches exactly as n_tokens calls of
 * zg_gpt_forward would.  tokens is [batch][token_stride].  With compute_logits != 0 the logits of position
 * n_tokens-1 are produced as zg_gpt_forward(n_tokens, ...) would (zg_gpt_argmax / logits_out as there).
 * Afterwards decoding continues with zg_gpt_forward(n_tokens + 1, ...).  Both weight modes: bf16-weight handles multiply the
 * exact three-plane split of the activations with the weights, ZG_GPT_WEIGHTS_F32 handles split both operands (six plane
 * products: fp32-sgemm grade).  Not available on handles created with ZG_GPT_NO_PREFILL. */
int zg_gpt_prefill(zg_gpt* g, const size_t* tokens, size_t token_stride, size_t n_tokens,
                   int compute_logits, float* logits_out, size_t logits_len);
/* argmax of the logits of the last zg_gpt_forward(compute_logits=1) per sequence (lowest index
 * wins ties; logits that are all NaN give index 0) — the greedy replacement for GPT.sample (src/main.zig:198-207). */
int zg_gpt_argmax(zg_gpt* g, size_t* tokens_out, size_t n_tokens);
/* GPT.sample — src/main.zig:198-207 for all sequences: zg_gpt_forward(seq_len, tokens, logits), then
 * logits /= temp, softmax, and an index drawn with probability proportional to the result
 * (std.rand weightedIndex: first index whose running sum exceeds u * total).  The reference re-seeds
 * its PRNG from the wall clock on every call; here the uniforms u[b] in [0,1) are supplied by the
 * caller (uniforms == NULL: derived from `seed`, seq_len and b with the library's counter PRNG), so a
 * run is reproducible.  probs_out (host or device, [batch, vocab]) optionally receives the softmax. */
int zg_gpt_sample(zg_gpt* g, size_t seq_len, const size_t* tokens, size_t n_tokens, float temp,
                  const float* uniforms, uint64_t seed, size_t* tokens_out, float* probs_out, size_t probs_len);

/* ln_f output of the last forward, [batch, n_embed] (state.x, main.zig:189). */
int zg_gpt_hidden(zg_gpt* g, float* x_out, size_t len);

/* generate — src/main.zig:322-342 with greedy argmax instead of the sampler, for `batch`
 * prompts in lock step.  prompts is [batch, prompt_stride] (usize), prompt_lens[b] >= 1 tokens
 * are used from row b.  Runs n_steps (<= context_size; the reference always runs context_size,
 * main.zig:330) decode steps without any host round trip per token; out_tokens [batch, n_steps]
 * receives the token after every step (prompt tokens included, main.zig:339-340).
 * As in the reference the last prompt token is fed twice (main.zig:334,337). */
int zg_gpt_generate_greedy(zg_gpt* g, const size_t* prompts, size_t prompt_stride,
                           const size_t* prompt_lens, size_t n_steps, size_t* out_tokens,
                           size_t out_len);
/* Asynchronous form used by the benchmark: enqueue the same n_steps on the stream and return;
 * results stay on the device until zg_gpt_generate_fetch. */
int zg_gpt_generate_enqueue(zg_gpt* g, const size_t* prompts, size_t prompt_stride,
                            const size_t* prompt_lens, size_t n_steps);
int zg_gpt_generate_fetch(zg_gpt* g, size_t n_steps, size_t* out_tokens, size_t out_len);
/* generate AS THE REFERENCE RUNS IT (src/main.zig:322-342 with GPT.sample, :198-207, temp 0.8 in main): every token behind the prompt is
 * drawn — softmax(logits / temp), first index whose running sum exceeds u x total — with the whole loop on the device (the sampler
 * is a node of the captured decode step).  The reference re-seeds from the wall clock per token; here the uniform of (sequence b,
 * position T) is the library's counter PRNG of (seed, T, b), the one zg_gpt_sample uses for uniforms == NULL: the call returns
 * exactly the tokens of the host loop `tok = zg_gpt_sample(g, T, &tok, 1, temp, NULL, seed, ...)`, without a host round trip per
 * token (4.6 k tokens/s at 124M; the per-token call: 4.0 k).  Results through zg_gpt_generate_fetch. */
int zg_gpt_generate_sample_enqueue(zg_gpt* g, const size_t* prompts, size_t prompt_stride, const size_t* prompt_lens, size_t n_steps,
                                   float temp, uint64_t seed);
int zg_gpt_generate_sample(zg_gpt* g, const size_t* prompts, size_t prompt_stride, const size_t* prompt_lens, size_t n_steps, float temp,
                           uint64_t seed, size_t* out_tokens, size_t out_len);
/* The same for the prompts of several handles at once (handles on distinct streams: zg_gpt_create_ex): prompts is
 * [sum of the handles' batches, prompt_stride], rows in handle order, prompt_lens alike; every handle generates its own rows.
 * A graph launch returns only when its hardware queue has room, so the handles are fed by one short-lived feeder thread each
 * (inside this call; ZGPT2_MANY_THREADS=0: one thread, one graph launch per handle in turn) — no queue starves while another
 * one is being filled.
 * zg_gpt_generate_fetch_many drains every handle and returns out_tokens [sum of batches, n_steps] in the same row order.
 * Token for token the result equals one handle per prompt (sequences never interact). */
int zg_gpt_generate_enqueue_many(zg_gpt* const* handles, size_t n_handles, const size_t* prompts, size_t prompt_stride,
                                 const size_t* prompt_lens, size_t n_steps);
int zg_gpt_generate_fetch_many(zg_gpt* const* handles, size_t n_handles, size_t n_steps, size_t* out_tokens, size_t out_len);

/* ------------------------------------------------------------------ measurement helpers ---- */

/* Replay one kernel class of the decode step (layer 0; classes numbered as in
 * zg_gpt_profile_step: 0 embed ... 6 lm_head) `iters` times back to back from a hipGraph at
 * seq_len = context/2 and return the average device time per launch in microseconds (launch
 * boundary included) and the kernel's algorithmic weight bytes.  A warm-cache microbenchmark: the
 * in-situ numbers are zg_gpt_profile_step's.  Options ride in the upper bits of the class argument: ZG_TIME_WALK_LAYERS walks
 * the layers (launch i takes layer i mod n_layer, so that no launch finds its weights in the L2s: the memory-side cost of the
 * real step), ZG_TIME_AT(t) runs the chain at sequence length t (t < 32768: the option field is 15 bits wide). */
#define ZG_TIME_WALK_LAYERS 0x100
#define ZG_TIME_AT(t) ((int)((unsigned)(t) << 16))
int zg_gpt_time_kernel(zg_gpt* g, int which_and_options, int iters, float* avg_us, size_t* algorithmic_bytes);

/* Run `iters` consecutive decode steps starting at sequence length seq_len as EAGER launches with a
 * HIP event between every two kernels, and return the average device time in microseconds that
 * one step spends per kernel class (classes 1-5 summed over the layers):
 *   [0] token select + embedding   [1] ln_1 + c_attn + KV append   [2] decode attention
 *   [3] head merge + attn c_proj + residual   [4] ln_2 + c_fc + gelu   [5] mlp c_proj + residual
 *   [6] ln_f + lm_head + argmax     [7] sum of all intervals
 *   [8] (when n_out >= 9) the interval of a one-element copy kernel recorded the same way: the launch
 *       boundary + event overhead contained in every interval above.
 * Intervals are event-to-event, so each includes the boundary to the next kernel. */
int zg_gpt_profile_step(zg_gpt* g, size_t seq_len, int iters, float* us_out, size_t n_out);

/* Diagnostic: how the side-stream prefetcher of the last zg_gpt_generate_* call ended (waits for both streams).
 * out[0] = 1 when the handle has a prefetcher (2: it found no concurrency with the decode stream once and is no longer
 * launched); then per XCD x = 0..7: out[1 + x] prefetcher workgroups that ran
 * there, out[9 + x] why they left (1 = stop behind the last step, 2 = progress stalled), out[17 + x] launches they
 * fetched for.  n_out >= 25.  No reference counterpart. */
int zg_debug_prefetch_stats(zg_gpt* g, unsigned* out, size_t n_out);

/* ------------------------------------------------------------------------------------------------
 * Tokenizer — src/bpe.zig (host side).  Encoder.init takes the two JSON objects of
 * models/<size>/encoder.json (token string -> id) and byte_encoder.json (unicode char -> byte,
 * download_weights.py:69-90) as parallel arrays of NUL-terminated UTF-8 keys and their values.
 * encode / decode keep bpe.zig:60-118's behaviour, including its deviations from GPT-2 BPE (greedy longest
 * prefix instead of merges, POSIX character classes, runs of spaces become their own word).  A word longer
 * than the reference's 20-byte buffer (bpe.zig:73) returns ZG_ERR_SHAPE; n_out receives the count. */
typedef struct zg_bpe zg_bpe;
int zg_bpe_create(zg_bpe** out, const char* const* tokens, const size_t* token_ids, size_t n_tokens,
                  const char* const* unicode_chars, const unsigned char* bytes, size_t n_bytes);
int zg_bpe_destroy(zg_bpe* e);
int zg_bpe_encode(zg_bpe* e, const char* text, size_t text_len, size_t* out, size_t out_cap, size_t* n_out);
int zg_bpe_decode(zg_bpe* e, const size_t* ids, size_t n_ids, char* out, size_t out_cap, size_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* ZGPT2_H */
