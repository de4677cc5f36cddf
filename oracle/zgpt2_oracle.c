/*
 * zgpt2_oracle.c — CPU restatement of zig_gpt2's forward hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle (and the timed "CPU path" of bench.py's cpu_baseline leg).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * path (zig_gpt2_amd/, libzgpt2_hip.so) never links, imports or calls anything in oracle/.
 *
 * It restates, in plain C and IEEE fp32 exactly like the reference, the algorithm of
 *   src/ops.zig   (Linear, Embedding, LayerNorm, CausalSelfAttention{forward,split_qkv,transpose},
 *                  gelu, softmax, scaled_dot_product_attention)
 *   src/main.zig  (State buffers, MLP.forward, Block.forward, GPT.forward, generate loop)
 * of /root/reference (EugenHotaj/zig_gpt2 @ v1).  Every function cites the reference lines it
 * follows.  The Zig reference cannot be compiled in this environment (no zig toolchain), so
 * PARITY IS PINNED against the reference's own PyTorch oracles instead: tests/golden/ holds
 * outputs of /root/reference/generate_test_data.py (op level, the 8 tests of src/tests.zig) and of
 * the GPT definition in /root/reference/generate_nano_gpt.py:24-152 (model level), produced by
 * tests/golden/make_golden.py; tests/test_oracle_golden.py checks this file against them at the
 * reference tolerance (src/tests.zig:4-20).
 *
 * Third-party arithmetic: the reference calls cblas_sgemm (Apple Accelerate, or an un-vendored,
 * unpinned OpenBLAS: build.zig:26-32) at src/ops.zig:30,268,289.  BLAS accumulation order is
 * unpinned, so orc_sgemm below is a straightforward fp32 restatement of the BLAS contract
 * C = alpha*op(A)*op(B) + beta*C; optionally a real CBLAS can be plugged in with orc_use_cblas()
 * (used only for timing the CPU baseline).
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off).
 */
#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* Portable synthetic-data PRNG (not in the reference: there are no weights offline).           */
/* Bit-exact twin of zig_gpt2_amd/synth.py.  Integer-only up to one fp32 multiply/add.          */
/* ------------------------------------------------------------------------------------------ */

static inline uint64_t orc_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint64_t orc_rand_u64(uint64_t key, uint64_t i) {
    return orc_mix64(key + (i + 1) * 0x9E3779B97F4A7C15ULL);
}

static inline uint64_t orc_key(uint64_t seed) { return orc_mix64(seed + 0x9E3779B97F4A7C15ULL); }

static inline float orc_round_bf16(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    b += 0x7FFFu + ((b >> 16) & 1u);
    b &= 0xFFFF0000u;
    memcpy(&x, &b, 4);
    return x;
}

/* out[i] = mean + std * z_i, z_i ~ Irwin-Hall(4) of 16-bit uniforms, centred and scaled to unit
 * variance (|z| <= 3.46).  round_bf16 != 0 rounds each value to the nearest bf16 (RNE). */
void orc_fill_normal(uint64_t seed, size_t n, float mean, float std, int round_bf16, float* out) {
    const uint64_t key = orc_key(seed);
    const float scale = (float)((double)std / 37837.22753904532);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint64_t r = orc_rand_u64(key, i);
        int32_t s = (int32_t)(r & 0xFFFF) + (int32_t)((r >> 16) & 0xFFFF) +
                    (int32_t)((r >> 32) & 0xFFFF) + (int32_t)((r >> 48) & 0xFFFF) - 131070;
        float v = (float)s * scale;
        v = v + mean;
        out[i] = round_bf16 ? orc_round_bf16(v) : v;
    }
}

/* out[i] = lo + u_i * (hi - lo), u_i = 24 random bits * 2^-24. */
void orc_fill_uniform(uint64_t seed, size_t n, float lo, float hi, int round_bf16, float* out) {
    const uint64_t key = orc_key(seed);
    const float width = hi - lo;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint64_t r = orc_rand_u64(key, i);
        float u = (float)(uint32_t)(r >> 40) * 5.9604644775390625e-08f;
        float v = u * width;
        v = v + lo;
        out[i] = round_bf16 ? orc_round_bf16(v) : v;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SGEMM — the BLAS contract behind src/ops.zig:30,268,289 (row-major only).                     */
/* ------------------------------------------------------------------------------------------ */

typedef void (*cblas_sgemm_fn)(int order, int transa, int transb, int m, int n, int k, float alpha,
                               const float* a, int lda, const float* b, int ldb, float beta,
                               float* c, int ldc);
static cblas_sgemm_fn g_cblas = NULL;
static void* g_cblas_handle = NULL;
static int g_accum_double = 0;

int orc_num_threads(void);

/* Plug a real CBLAS in (timing only).  Returns 0 on success. */
int orc_use_cblas(const char* lib_path, const char* symbol) {
    if (!lib_path) {
        g_cblas = NULL;
        return 0;
    }
    void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    void* f = dlsym(h, symbol ? symbol : "cblas_sgemm");
    if (!f) {
        dlclose(h);
        return -2;
    }
    g_cblas_handle = h;
    g_cblas = (cblas_sgemm_fn)f;
    /* OpenBLAS sizes its thread pool from the CPUs it SEES (256 on a GPU box whose cgroup grants 16): left alone it
     * oversubscribes the quota and runs an order of magnitude slow.  Hold it to the threads the port itself uses
     * (OPENBLAS_NUM_THREADS = all usable cores). */
    const char* setters[] = {"scipy_openblas_set_num_threads", "openblas_set_num_threads", "goto_set_num_threads"};
    for (unsigned i = 0; i < sizeof(setters) / sizeof(setters[0]); ++i) {
        void (*set_threads)(int) = (void (*)(int))dlsym(h, setters[i]);
        if (set_threads) {
            set_threads(orc_num_threads());
            break;
        }
    }
    return 0;
}

/* Diagnostic: accumulate dot products in double (tells accumulation-order noise from real bugs). */
void orc_set_accum_double(int on) { g_accum_double = on; }

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* C[M,N] = alpha * A[M,K] * op(B) + beta * C ; transb: B is [N,K] (ldb>=K), else B is [K,N]. */
static void orc_sgemm(int transb, size_t M, size_t N, size_t K, float alpha, const float* A,
                      size_t lda, const float* B, size_t ldb, float beta, float* C, size_t ldc) {
    if (g_cblas) {
        /* CblasRowMajor=101, CblasNoTrans=111, CblasTrans=112 */
        g_cblas(101, 111, transb ? 112 : 111, (int)M, (int)N, (int)K, alpha, A, (int)lda, B,
                (int)ldb, beta, C, (int)ldc);
        return;
    }
    if (transb) {
        const int par = (M * N * K) > (1u << 16);
#pragma omp parallel for schedule(static) if (par)
        for (size_t n = 0; n < N; ++n) {
            const float* b = B + n * ldb;
            for (size_t m = 0; m < M; ++m) {
                const float* a = A + m * lda;
                float r;
                if (g_accum_double) {
                    double acc = 0.0;
                    for (size_t k = 0; k < K; ++k) acc += (double)a[k] * (double)b[k];
                    r = (float)acc;
                } else {
                    float acc = 0.0f;
#pragma omp simd reduction(+ : acc)
                    for (size_t k = 0; k < K; ++k) acc += a[k] * b[k];
                    r = acc;
                }
                float* c = C + m * ldc + n;
                *c = (beta == 0.0f) ? alpha * r : alpha * r + beta * (*c);
            }
        }
    } else {
        for (size_t m = 0; m < M; ++m) {
            float* c = C + m * ldc;
            const float* a = A + m * lda;
            if (g_accum_double) {
                for (size_t n = 0; n < N; ++n) {
                    double acc = 0.0;
                    for (size_t k = 0; k < K; ++k) acc += (double)a[k] * (double)B[k * ldb + n];
                    c[n] = (beta == 0.0f) ? alpha * (float)acc : alpha * (float)acc + beta * c[n];
                }
                continue;
            }
            if (beta == 0.0f) {
                for (size_t n = 0; n < N; ++n) c[n] = 0.0f;
            } else if (beta != 1.0f) {
                for (size_t n = 0; n < N; ++n) c[n] *= beta;
            }
            for (size_t k = 0; k < K; ++k) {
                const float av = alpha * a[k];
                const float* b = B + k * ldb;
#pragma omp simd
                for (size_t n = 0; n < N; ++n) c[n] += av * b[n];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/ops.zig restated                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* Linear.forward — src/ops.zig:21-46.  weight is [out,in] row-major (ops.zig:9 "column major"),
 * bias (may be NULL) is memcpy'd into every output row and the GEMM then runs with beta=1
 * (ops.zig:23-29, :42). */
void orc_linear_forward(size_t in_features, size_t out_features, const float* weight,
                        const float* bias, const float* inputs, size_t inputs_len,
                        float* outputs) {
    const size_t batch = inputs_len / in_features; /* ops.zig:22 */
    float beta = 0.0f;
    if (bias) {
        for (size_t b = 0; b < batch; ++b)
            memcpy(outputs + b * out_features, bias, out_features * sizeof(float));
        beta = 1.0f;
    }
    orc_sgemm(1, batch, out_features, in_features, 1.0f, inputs, in_features, weight, in_features,
              beta, outputs, out_features);
}

/* Embedding.forward — src/ops.zig:59-67 (row gather by memcpy; idxs are usize). */
void orc_embedding_forward(size_t emb_dim, const float* weight, const size_t* idxs, size_t n_idxs,
                           float* embeddings) {
    for (size_t i = 0; i < n_idxs; ++i)
        memcpy(embeddings + i * emb_dim, weight + emb_dim * idxs[i], emb_dim * sizeof(float));
}

/* LayerNorm.forward — src/ops.zig:82-104.  In place; single pass sum / sum of squares (:88-92);
 * std = sqrt(E[x^2] - mean^2 + eps) (:95); (x - mean) / std * w + b (:101). */
void orc_layernorm_forward(size_t n_features, const float* weight, const float* bias, float eps,
                           float* inputs, size_t inputs_len) {
    const size_t batch = inputs_len / n_features;
    for (size_t b = 0; b < batch; ++b) {
        float mean = 0.0f, std_ = 0.0f;
        float* row = inputs + b * n_features;
        for (size_t i = 0; i < n_features; ++i) {
            const float x = row[i];
            mean += x;
            std_ += x * x;
        }
        const float n = (float)n_features;
        mean /= n;
        std_ = sqrtf((std_ / n) - (mean * mean) + eps);
        for (size_t i = 0; i < n_features; ++i) {
            const float x = row[i];
            row[i] = (x - mean) / std_ * weight[i] + bias[i];
        }
    }
}

/* gelu — src/ops.zig:221-228 (tanh form, constants 0.7978845608 and 0.044715). */
void orc_gelu(float* inputs, size_t len) {
    for (size_t i = 0; i < len; ++i) {
        const float x = inputs[i];
        inputs[i] = 0.5f * x * (1.0f + tanhf(x * 0.7978845608f * (1.0f + 0.044715f * x * x)));
    }
}

/* softmax — src/ops.zig:231-241 (whole slice is one vector; max, exp, sum, divide). */
void orc_softmax(float* inputs, size_t len) {
    float max = inputs[0];
    for (size_t i = 1; i < len; ++i)
        if (inputs[i] > max) max = inputs[i];
    float sum = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        inputs[i] = expf(inputs[i] - max);
        sum += inputs[i];
    }
    for (size_t i = 0; i < len; ++i) inputs[i] /= sum;
}

/* CausalSelfAttention.split_qkv — src/ops.zig:177-196: [B,T,3E] -> [B,T,E], slice split_idx. */
void orc_split_qkv(size_t n_embed, size_t seq_len, const float* inputs, size_t inputs_len,
                   size_t split_idx, float* outputs) {
    const size_t n_embed_ = 3 * n_embed;
    const size_t batch = inputs_len / (seq_len * n_embed_);
    for (size_t b = 0; b < batch; ++b)
        for (size_t r = 0; r < seq_len; ++r) {
            const size_t out_off = b * seq_len * n_embed + r * n_embed;
            const size_t in_off = b * seq_len * n_embed_ + r * n_embed_ + split_idx * n_embed;
            memcpy(outputs + out_off, inputs + in_off, n_embed * sizeof(float));
        }
}

/* CausalSelfAttention.transpose — src/ops.zig:199-216: (b,t,n,h) -> (b,n,t,h). */
void orc_transpose(size_t seq_len, size_t n_heads, size_t head_dim, const float* inputs,
                   size_t inputs_len, float* outputs) {
    const size_t batch = inputs_len / (seq_len * n_heads * head_dim);
    for (size_t b = 0; b < batch; ++b)
        for (size_t h = 0; h < n_heads; ++h)
            for (size_t s = 0; s < seq_len; ++s) {
                const size_t in_off =
                    b * seq_len * n_heads * head_dim + s * n_heads * head_dim + h * head_dim;
                const size_t out_off =
                    b * seq_len * n_heads * head_dim + h * seq_len * head_dim + s * head_dim;
                memcpy(outputs + out_off, inputs + in_off, head_dim * sizeof(float));
            }
}

/* scaled_dot_product_attention — src/ops.zig:249-307.  q [B,H,1,hd], k/v [B,H,T,hd];
 * per (b,h): _attn[T] = (1/sqrt(hd)) q K^T (sgemm alpha, :275); softmax(_attn) (:284);
 * out[hd] = _attn V (:289).  _attn must be exactly T long. */
void orc_sdpa(const float* q, const float* k, size_t k_len, const float* v, size_t n_heads,
              size_t seq_len, size_t head_dim, float* outputs, float* _attn) {
    const size_t batch = k_len / (n_heads * seq_len * head_dim); /* ops.zig:259 */
    const float alpha = 1.0f / sqrtf((float)head_dim);
    for (size_t b = 0; b < batch; ++b)
        for (size_t h = 0; h < n_heads; ++h) {
            const size_t qo = b * n_heads * head_dim + h * head_dim;
            const size_t kv = b * n_heads * seq_len * head_dim + h * seq_len * head_dim;
            orc_sgemm(1, 1, seq_len, head_dim, alpha, q + qo, head_dim, k + kv, head_dim, 0.0f,
                      _attn, seq_len);
            orc_softmax(_attn, seq_len);
            orc_sgemm(0, 1, head_dim, seq_len, 1.0f, _attn, seq_len, v + kv, head_dim, 0.0f,
                      outputs + qo, head_dim);
        }
}

/* CausalSelfAttention.forward — src/ops.zig:129-173 (batch == 1, decode step with KV cache).
 * k_cache / v_cache are the caller's [seq_len, E] slices (row t = token t); `outputs` doubles as
 * scratch (:146,151,156); the whole cache is re-transposed into _k/_v every call (:153,:158). */
void orc_attn_forward(size_t n_heads, size_t n_embed, const float* c_attn_w, const float* c_attn_b,
                      const float* c_proj_w, const float* c_proj_b, size_t seq_len,
                      const float* inputs, float* k_cache, float* v_cache, float* outputs,
                      float* _qkv, float* _q, float* _k, float* _v, float* _attn) {
    const size_t head_dim = n_embed / n_heads;
    orc_linear_forward(n_embed, 3 * n_embed, c_attn_w, c_attn_b, inputs, n_embed, _qkv); /* :143 */

    orc_split_qkv(n_embed, 1, _qkv, 3 * n_embed, 0, outputs);              /* :146 */
    orc_transpose(1, n_heads, head_dim, outputs, n_embed, _q);             /* :147 */

    orc_split_qkv(n_embed, 1, _qkv, 3 * n_embed, 1, outputs);              /* :151 */
    memcpy(k_cache + (seq_len - 1) * n_embed, outputs, n_embed * sizeof(float)); /* :152 */
    orc_transpose(seq_len, n_heads, head_dim, k_cache, seq_len * n_embed, _k);   /* :153 */

    orc_split_qkv(n_embed, 1, _qkv, 3 * n_embed, 2, outputs);              /* :156 */
    memcpy(v_cache + (seq_len - 1) * n_embed, outputs, n_embed * sizeof(float)); /* :157 */
    orc_transpose(seq_len, n_heads, head_dim, v_cache, seq_len * n_embed, _v);   /* :158 */

    orc_sdpa(_q, _k, seq_len * n_embed, _v, n_heads, seq_len, head_dim, outputs, _attn); /* :160 */
    orc_transpose(n_heads, 1, head_dim, outputs, n_embed, _q);             /* :171 */
    orc_linear_forward(n_embed, n_embed, c_proj_w, c_proj_b, _q, n_embed, outputs); /* :172 */
}

/* ------------------------------------------------------------------------------------------ */
/* src/main.zig restated: GPTConfig (:5-23), State (:26-65), MLP (:67-83), Block (:85-147),     */
/* GPT (:149-208), generate (:322-342) with greedy argmax in place of the sampler.              */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    size_t vocab_size, context_size, n_layer, n_heads, n_embed;
} orc_gpt_config;

typedef struct {
    const float *ln_1_g, *ln_1_b;
    const float *c_attn_w, *c_attn_b;
    const float *c_proj_w, *c_proj_b;
    const float *ln_2_g, *ln_2_b;
    const float *c_fc_w, *c_fc_b;
    const float *mlp_proj_w, *mlp_proj_b;
    float *k_cache, *v_cache; /* owned: [context_size * n_embed] each (main.zig:298-299) */
} orc_block;

typedef struct {
    orc_gpt_config config;
    const float* wte; /* [vocab, E]; lm_head shares it (main.zig:312) */
    const float* wpe; /* [ctx, E] */
    const float *ln_f_g, *ln_f_b;
    orc_block* h;
    /* State — main.zig:26-65 */
    float *pos_emb, *x, *o, *logits, *_h, *_4xh, *_qkv, *_q, *_k, *_v, *_attn;
} orc_gpt;

/* Block-weight slots for orc_gpt_set_block_tensor */
enum {
    ORC_LN_1_G = 0,
    ORC_LN_1_B,
    ORC_C_ATTN_W,
    ORC_C_ATTN_B,
    ORC_C_PROJ_W,
    ORC_C_PROJ_B,
    ORC_LN_2_G,
    ORC_LN_2_B,
    ORC_C_FC_W,
    ORC_C_FC_B,
    ORC_MLP_PROJ_W,
    ORC_MLP_PROJ_B
};
enum { ORC_WTE = 0, ORC_WPE, ORC_LN_F_G, ORC_LN_F_B };

static float* orc_alloc(size_t n) {
    float* p = (float*)calloc(n ? n : 1, sizeof(float));
    if (!p) {
        fprintf(stderr, "zgpt2_oracle: out of memory (%zu floats)\n", n);
        abort();
    }
    return p;
}

/* State.init + per-block cache allocation (main.zig:46-64, :298-299).  Weights are borrowed. */
orc_gpt* orc_gpt_create(size_t vocab_size, size_t context_size, size_t n_layer, size_t n_heads,
                        size_t n_embed) {
    orc_gpt* g = (orc_gpt*)calloc(1, sizeof(orc_gpt));
    g->config = (orc_gpt_config){vocab_size, context_size, n_layer, n_heads, n_embed};
    g->h = (orc_block*)calloc(n_layer, sizeof(orc_block));
    for (size_t i = 0; i < n_layer; ++i) {
        g->h[i].k_cache = orc_alloc(context_size * n_embed);
        g->h[i].v_cache = orc_alloc(context_size * n_embed);
    }
    g->pos_emb = orc_alloc(n_embed);
    g->x = orc_alloc(n_embed);
    g->o = orc_alloc(n_embed);
    g->logits = orc_alloc(vocab_size);
    g->_h = orc_alloc(n_embed);
    g->_4xh = orc_alloc(4 * n_embed);
    g->_qkv = orc_alloc(3 * n_embed);
    g->_q = orc_alloc(n_embed);
    g->_k = orc_alloc(context_size * n_embed);
    g->_v = orc_alloc(context_size * n_embed);
    g->_attn = orc_alloc(context_size);
    return g;
}

void orc_gpt_destroy(orc_gpt* g) {
    if (!g) return;
    for (size_t i = 0; i < g->config.n_layer; ++i) {
        free(g->h[i].k_cache);
        free(g->h[i].v_cache);
    }
    free(g->h);
    free(g->pos_emb);
    free(g->x);
    free(g->o);
    free(g->logits);
    free(g->_h);
    free(g->_4xh);
    free(g->_qkv);
    free(g->_q);
    free(g->_k);
    free(g->_v);
    free(g->_attn);
    free(g);
}

int orc_gpt_set_block_tensor(orc_gpt* g, size_t layer, int slot, const float* p) {
    if (layer >= g->config.n_layer) return -1;
    orc_block* b = &g->h[layer];
    switch (slot) {
        case ORC_LN_1_G: b->ln_1_g = p; break;
        case ORC_LN_1_B: b->ln_1_b = p; break;
        case ORC_C_ATTN_W: b->c_attn_w = p; break;
        case ORC_C_ATTN_B: b->c_attn_b = p; break;
        case ORC_C_PROJ_W: b->c_proj_w = p; break;
        case ORC_C_PROJ_B: b->c_proj_b = p; break;
        case ORC_LN_2_G: b->ln_2_g = p; break;
        case ORC_LN_2_B: b->ln_2_b = p; break;
        case ORC_C_FC_W: b->c_fc_w = p; break;
        case ORC_C_FC_B: b->c_fc_b = p; break;
        case ORC_MLP_PROJ_W: b->mlp_proj_w = p; break;
        case ORC_MLP_PROJ_B: b->mlp_proj_b = p; break;
        default: return -2;
    }
    return 0;
}

int orc_gpt_set_tensor(orc_gpt* g, int slot, const float* p) {
    switch (slot) {
        case ORC_WTE: g->wte = p; break;
        case ORC_WPE: g->wpe = p; break;
        case ORC_LN_F_G: g->ln_f_g = p; break;
        case ORC_LN_F_B: g->ln_f_b = p; break;
        default: return -2;
    }
    return 0;
}

float* orc_gpt_logits(orc_gpt* g) { return g->logits; }
float* orc_gpt_x(orc_gpt* g) { return g->x; }

/* MLP.forward — main.zig:78-82: _4xh = c_fc(x); gelu; o = c_proj(_4xh). */
static void orc_mlp_forward(orc_gpt* g, const orc_block* b, const float* inputs) {
    const size_t E = g->config.n_embed;
    orc_linear_forward(E, 4 * E, b->c_fc_w, b->c_fc_b, inputs, E, g->_4xh);
    orc_gelu(g->_4xh, 4 * E);
    orc_linear_forward(4 * E, E, b->mlp_proj_w, b->mlp_proj_b, g->_4xh, 4 * E, g->o);
}

/* Block.forward — main.zig:119-146 (the loop bounds `0..state.o` at :136,:142 are read as
 * `0..state.o.len`): x1 = x + attn(ln_1(x)); out = x1 + mlp(ln_2(x1)); result in x and o. */
static void orc_block_forward(orc_gpt* g, const orc_block* b, size_t seq_len, const float* inputs) {
    const size_t E = g->config.n_embed;
    memcpy(g->_h, inputs, E * sizeof(float));                                         /* :121 */
    orc_layernorm_forward(E, b->ln_1_g, b->ln_1_b, 1e-5f, g->_h, E);                  /* :123 */
    orc_attn_forward(g->config.n_heads, E, b->c_attn_w, b->c_attn_b, b->c_proj_w, b->c_proj_b,
                     seq_len, g->_h, b->k_cache, b->v_cache, g->o, g->_qkv, g->_q, g->_k, g->_v,
                     g->_attn);                                                       /* :124-135 */
    for (size_t i = 0; i < E; ++i) {                                                  /* :136-139 */
        g->_h[i] = g->o[i] + inputs[i];
        g->x[i] = g->_h[i];
    }
    orc_layernorm_forward(E, b->ln_2_g, b->ln_2_b, 1e-5f, g->_h, E);                  /* :140 */
    orc_mlp_forward(g, b, g->_h);                                                     /* :141 */
    for (size_t i = 0; i < E; ++i) {                                                  /* :142-145 */
        g->o[i] += g->x[i];
        g->x[i] = g->o[i];
    }
}

/* GPT.forward — main.zig:178-195. */
void orc_gpt_forward(orc_gpt* g, size_t seq_len, size_t token, int compute_logits) {
    const size_t E = g->config.n_embed;
    const size_t pos = seq_len - 1;
    orc_embedding_forward(E, g->wpe, &pos, 1, g->pos_emb);                            /* :179 */
    orc_embedding_forward(E, g->wte, &token, 1, g->x);                                /* :180 */
    for (size_t i = 0; i < E; ++i) g->x[i] += g->pos_emb[i];                          /* :181-183 */
    for (size_t l = 0; l < g->config.n_layer; ++l)                                    /* :186-188 */
        orc_block_forward(g, &g->h[l], seq_len, g->x);
    orc_layernorm_forward(E, g->ln_f_g, g->ln_f_b, 1e-5f, g->x, E);                   /* :189 */
    if (compute_logits)                                                               /* :192-194 */
        orc_linear_forward(E, g->config.vocab_size, g->wte, NULL, g->x, E, g->logits);
}

/* Greedy replacement for GPT.sample (main.zig:198-207): forward with logits, then argmax
 * (lowest index wins ties) instead of temperature softmax + time-seeded multinomial. */
size_t orc_gpt_sample_greedy(orc_gpt* g, size_t seq_len, size_t token) {
    orc_gpt_forward(g, seq_len, token, 1);
    size_t best = 0;
    for (size_t i = 1; i < g->config.vocab_size; ++i)
        if (g->logits[i] > g->logits[best]) best = i;
    return best;
}

/* GPT.sample — main.zig:198-207: forward with logits; logits /= temp (:200-202); softmax (:203);
 * random.weightedIndex(f32, logits) (:204-206).  The reference seeds a fresh PRNG from the wall clock on
 * every call, so its draw is not reproducible; here the uniform u in [0,1) is an argument.  weightedIndex
 * (Zig std.rand) picks `point = u * sum(weights)` and returns the first index whose running sum exceeds
 * it (the last index if rounding leaves none). */
size_t orc_gpt_sample(orc_gpt* g, size_t seq_len, size_t token, float temp, float u) {
    orc_gpt_forward(g, seq_len, token, 1);
    const size_t V = g->config.vocab_size;
    for (size_t i = 0; i < V; ++i) g->logits[i] /= temp;
    orc_softmax(g->logits, V);
    float sum = 0.0f;
    for (size_t i = 0; i < V; ++i) sum += g->logits[i];
    const float point = u * sum;
    float acc = 0.0f;
    for (size_t i = 0; i < V; ++i) {
        acc += g->logits[i];
        if (point < acc) return i;
    }
    return V - 1;
}

/* generate — main.zig:322-342, greedy.  Runs n_steps (<= context_size; the reference always runs
 * context_size, :330) iterations; prompt tokens are fed one at a time without logits (:331-334),
 * then sample(s+1, token) re-feeds the previous token at the next position (:337) — so the last
 * prompt token is fed twice, exactly as the reference does.  out_tokens[s] = token after step s.
 * If logits_out != NULL it receives the logits of every generation step, [n_steps - n_prompt, V]. */
void orc_gpt_generate_greedy(orc_gpt* g, const size_t* prompt, size_t n_prompt, size_t n_steps,
                             size_t* out_tokens, float* logits_out) {
    size_t token = 0;
    size_t gen = 0;
    for (size_t s = 0; s < n_steps; ++s) {
        if (s < n_prompt) {
            token = prompt[s];
            orc_gpt_forward(g, s + 1, token, 0);
        } else {
            token = orc_gpt_sample_greedy(g, s + 1, token);
            if (logits_out)
                memcpy(logits_out + gen * g->config.vocab_size, g->logits,
                       g->config.vocab_size * sizeof(float));
            ++gen;
        }
        out_tokens[s] = token;
    }
}

/* Teacher-forced variant used by parity tests: feeds forced[s] at every step (so a near-tie in
 * argmax cannot fork the two implementations) and records logits for every step s >= n_prompt. */
void orc_gpt_forced_logits(orc_gpt* g, const size_t* forced, size_t n_steps, size_t first_logit_step,
                           float* logits_out) {
    for (size_t s = 0; s < n_steps; ++s) {
        const int want = s >= first_logit_step;
        orc_gpt_forward(g, s + 1, forced[s], want);
        if (want)
            memcpy(logits_out + (s - first_logit_step) * g->config.vocab_size, g->logits,
                   g->config.vocab_size * sizeof(float));
    }
}
