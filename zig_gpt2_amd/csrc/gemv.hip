// gemv.hip — the decode-regime Linear: y[M,N] = x[M,K] * W[N,K]^T (+ bias) for M <= 8.
//
// Replaces the cblas_sgemm call of Linear.forward (reference src/ops.zig:21-46) when the batch
// is tiny (decode: M = number of lock-step sequences).  HBM-bandwidth bound: every weight byte is
// read exactly once, 16 B per lane, K-contiguous rows ([out,in] layout of ops.Linear.weight), so
// one row is read by a group of LPR lanes with fully coalesced 16-B loads and reduced with DPP
// row rotations (M == 1: no barrier, except where a workgroup shares one copy of a wide input).
//
// Fused around the dot products (the reference does these as separate host loops / ops):
//   prologue  PRO_LAYERNORM   LayerNorm.forward of the input row   (src/ops.zig:82-104)
//             PRO_ATTN_MERGE  combine split-KV attention partials  (src/ops.zig:284-305 tail)
//   epilogue  EPI_RESIDUAL    state.o + state.x residual adds      (src/main.zig:136-145)
//             EPI_GELU        ops.gelu                             (src/ops.zig:221-228)
//             EPI_QKV         split_qkv + KV-cache append          (src/ops.zig:146-157)
//             EPI_ARGMAX      greedy sampler partial argmax        (replaces src/main.zig:198-207)
#include <stdlib.h>

#include "zg_kernels.h"

// Diagnostic build (-DZG_STAMPS): wave 0 of the first and of the last workgroup record s_memtime at
// fixed points of the kernel and append them to GemvArgs::dbg.  Compiled out of the product build.
#ifdef ZG_STAMPS
#define ZG_STAMP_DECL() unsigned long long zg_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define ZG_STAMP(i) zg_ts[i] = __builtin_amdgcn_s_memtime()
#define ZG_STAMP_FLUSH()                                                                              \
    if (a.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {             \
        const unsigned long long slot = atomicAdd(a.dbg, 1ull);                                       \
        unsigned long long* d = a.dbg + 16 + slot * 10;                                               \
        for (int i = 0; i < 8; ++i) d[i] = zg_ts[i];                                                  \
        d[8] = blockIdx.x;                                                                            \
        d[9] = __builtin_amdgcn_s_memtime();                                                          \
    }
#else
#define ZG_STAMP_DECL()
#define ZG_STAMP(i)
#define ZG_STAMP_FLUSH()
#endif

namespace zg {

namespace {

struct W8 {
    float v[8];
};

// Raw (still packed) 8-element weight chunk: kept packed in registers until the FMAs so that two
// passes of loads in flight cost 4 VGPRs per bf16 chunk, not 8.
template <typename WT>
struct Raw;
template <>
struct Raw<bf16_t> {
    u32x4 p;
};
template <>
struct Raw<float> {
    f32x4 a, b;
};

__device__ __forceinline__ Raw<bf16_t> load_raw(const bf16_t* row, int c) {
    Raw<bf16_t> r;
    r.p = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row) + c);
    return r;
}
__device__ __forceinline__ Raw<float> load_raw(const float* row, int c) {
    Raw<float> r;
    r.a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row) + 2 * c);
    r.b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row) + 2 * c + 1);
    return r;
}
__device__ __forceinline__ void zero_raw(Raw<bf16_t>& r) { r.p = u32x4{0u, 0u, 0u, 0u}; }
__device__ __forceinline__ void zero_raw(Raw<float>& r) {
    r.a = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    r.b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
__device__ __forceinline__ W8 unpack(const Raw<bf16_t>& r) {
    W8 w;
    w.v[0] = bf16_lo(r.p.x); w.v[1] = bf16_hi(r.p.x);
    w.v[2] = bf16_lo(r.p.y); w.v[3] = bf16_hi(r.p.y);
    w.v[4] = bf16_lo(r.p.z); w.v[5] = bf16_hi(r.p.z);
    w.v[6] = bf16_lo(r.p.w); w.v[7] = bf16_hi(r.p.w);
    return w;
}
__device__ __forceinline__ W8 unpack(const Raw<float>& r) {
    W8 w;
    w.v[0] = r.a.x; w.v[1] = r.a.y; w.v[2] = r.a.z; w.v[3] = r.a.w;
    w.v[4] = r.b.x; w.v[5] = r.b.y; w.v[6] = r.b.z; w.v[7] = r.b.w;
    return w;
}

__device__ __forceinline__ W8 zero_w8() {
    W8 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) w.v[j] = 0.0f;
    return w;
}

__device__ __forceinline__ W8 load_x8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    W8 w;
    w.v[0] = a.x; w.v[1] = a.y; w.v[2] = a.z; w.v[3] = a.w;
    w.v[4] = b.x; w.v[5] = b.y; w.v[6] = b.z; w.v[7] = b.w;
    return w;
}

__device__ __forceinline__ float dot8(const W8& w, const W8& x, float acc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(w.v[j], x.v[j], acc);
    return acc;
}

struct Best {
    float val;
    int idx;
};
__device__ __forceinline__ Best better(Best a, Best b) {
    return (b.val > a.val || (b.val == a.val && b.idx < a.idx)) ? b : a;
}
__device__ __forceinline__ Best wave_best(Best b) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Best o;
        o.val = __shfl_xor(b.val, off, 64);
        o.idx = __shfl_xor(b.idx, off, 64);
        b = better(b, o);
    }
    return b;
}

template <typename KV>
__device__ __forceinline__ void kv_store(void* cache, size_t off, float v) {
    // fp16 cache: saturate instead of overflowing to inf (a masked position holding inf would turn p = 0 into NaN)
    if (sizeof(KV) == 2) v = fminf(fmaxf(v, -65504.0f), 65504.0f);
    reinterpret_cast<KV*>(cache)[off] = (KV)v;
}

// bias_n / resid_mn were fetched together with the row's weights (no dependent round trip here).
__device__ __forceinline__ float epilogue_row(const GemvArgs& a, int m, int n, float acc, float bias_n,
                                              float resid_mn, int pos, Best& best) {
    float v = acc + bias_n;
    switch (a.epilogue) {
        case EPI_STORE:
            a.y[(size_t)m * a.y_stride + n] = v;
            break;
        case EPI_RESIDUAL:
            v += resid_mn;
            a.y[(size_t)m * a.y_stride + n] = v;
            break;
        case EPI_GELU:
            v = gelu_ref(v);
            if (a.yg)  // two-stream decode: (value, tag) for the K-split kernel resident on another stream
                __hip_atomic_store(a.yg + n, ((unsigned long long)((*a.epoch2 << 8) | a.yout_id) << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            else if (a.y) a.y[(size_t)m * a.y_stride + n] = v;  // null: the output leaves as planes only (GemvArgs.pl_out)
            break;
        case EPI_QKV: {
            const int E = a.N / 3;
            if (n < E) {
                a.q[(size_t)m * E + n] = v;
            } else {
                const int which = n >= 2 * E;
                const int e = n - (which ? 2 * E : E);
                int h, d;
                if (a.head_dim == 64) {  // the GPT-2 family: no integer division in the epilogue
                    h = e >> 6;
                    d = e & 63;
                } else {
                    h = e / a.head_dim;
                    d = e % a.head_dim;
                }
                const size_t off = (((size_t)m * a.n_heads + h) * a.ctx + pos) * a.head_dim + d;
                void* cache = which ? a.v_cache : a.k_cache;
                if (a.kv_f16) kv_store<_Float16>(cache, off, v);
                else kv_store<float>(cache, off, v);
            }
            break;
        }
        case EPI_ARGMAX: {
            if (a.logits) a.logits[(size_t)m * a.logits_stride + n] = v;
            Best c;
            c.val = v;
            c.idx = n;
            best = better(best, c);
            break;
        }
    }
    return v;
}

// Merged attention output for elements [e0, e0+4) of sequence m: all loads issued before any math.
__device__ __forceinline__ f32x4 merge_attn4(const GemvArgs& a, int m, int e0, int nsplit) {
    const int h = e0 / a.head_dim, d0 = e0 % a.head_dim;
    const float* p = a.part + ((size_t)(m * a.n_heads + h) * a.max_splits) * kPartStride;
    constexpr int MAXS = 4;  // ctx 1024 / 256; more splits fall back to the loop below
    if (nsplit <= MAXS) {
        float ms[MAXS], ls[MAXS];
        float o[MAXS][4];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {  // branch-free: surplus splits re-read the last valid one ...
            const float* ps = p + min(s, nsplit - 1) * kPartStride;
            ms[s] = ps[64];
            ls[s] = ps[65];
            const float2 lo = *reinterpret_cast<const float2*>(ps + d0);      // 8-B aligned: kPartStride
            const float2 hi = *reinterpret_cast<const float2*>(ps + d0 + 2);  // and d0 are even
            o[s][0] = lo.x; o[s][1] = lo.y; o[s][2] = hi.x; o[s][3] = hi.y;
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s >= nsplit) ms[s] = -1e30f;  // ... and get weight exp(-1e30 - max) == 0
        const float mx = fmaxf(fmaxf(ms[0], ms[1]), fmaxf(ms[2], ms[3]));
        float l = 0.0f;
        float r[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const float w = __expf(ms[s] - mx);
            l = fmaf(w, ls[s], l);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = fmaf(w, o[s][j], r[j]);
        }
        const float inv = 1.0f / l;
        return f32x4{r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv};
    }
    float mx = -1e30f;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, p[s * kPartStride + 64]);
    float l = 0.0f;
    float r[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < nsplit; ++s) {
        const float w = __expf(p[s * kPartStride + 64] - mx);
        l = fmaf(w, p[s * kPartStride + 65], l);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = fmaf(w, p[s * kPartStride + d0 + j], r[j]);
    }
    const float inv = 1.0f / l;
    return f32x4{r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv};
}

// Branch-free: out-of-range rows / chunks are clamped to a valid address instead of predicated, so the
// loads stay in straight-line code and the compiler can wait for them with counted vmcnt (predicated
// loads sit in exec-masked branches, after which it falls back to vmcnt(0) and the pass pipeline
// collapses).  A clamped chunk multiplies an input that is zero; a clamped row's result is discarded.
template <typename WT, int LPR, int CPL>
__device__ __forceinline__ void load_pass(Raw<WT> (&w)[CPL], const WT* W, int K, int nch, int row, int n_rows,
                                          int lr) {
    const WT* wp = W + (size_t)max(min(row, n_rows - 1), 0) * K;  // n_rows = this wave's row_end: surplus slots re-read its own last row
#pragma unroll
    for (int i = 0; i < CPL; ++i) w[i] = load_raw(wp, min(lr + LPR * i, nch - 1));
}

// Per-row epilogue operands, requested together with the row's weights.
template <int MT>
struct RowExtra {
    float bias;
    float resid[MT];
};
template <int MT>
__device__ __forceinline__ RowExtra<MT> load_extra(const GemvArgs& a, int epilogue, int M, int N, int r) {
    RowExtra<MT> e;
    const int rr = min(r, N - 1);
    e.bias = *(a.bias ? a.bias + rr : a.zero);
#pragma unroll
    for (int m = 0; m < MT; ++m)
        e.resid[m] = *((epilogue == EPI_RESIDUAL && m < M) ? a.resid + (size_t)m * a.resid_stride + rr : a.zero);
    return e;
}

// LayerNorm of one input row by ONE wave into its private LDS strip (NJ float4 per lane cover the row).
// Single pass sum / sum of squares; std = sqrt(E[x^2] - mean^2 + eps): reference src/ops.zig:88-101.
// Loads are branch-free (index clamped, surplus zeroed afterwards) so they all fly together.
template <int NJ>
__device__ __forceinline__ void ln_strip(const float* __restrict__ xin, const float* __restrict__ ln_g,
                                         const float* __restrict__ ln_b, f32x4* xw4, int nq, int K, float eps,
                                         int lane) {
    f32x4 v[NJ], g4[NJ], b4[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ic = min(lane + 64 * j, nq - 1);
        v[j] = reinterpret_cast<const f32x4*>(xin)[ic];
        g4[j] = reinterpret_cast<const f32x4*>(ln_g)[ic];
        b4[j] = reinterpret_cast<const f32x4*>(ln_b)[ic];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (lane + 64 * j >= nq) v[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        t1 += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        t2 = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, t2))));
    }
    t1 = wave_allsum(t1);
    t2 = wave_allsum(t2);
    const float inv_k = 1.0f / (float)K;
    const float mean = t1 * inv_k;
    const float rstd = __builtin_amdgcn_rsqf(t2 * inv_k - mean * mean + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = lane + 64 * j;
        if (i < nq) {
            f32x4 o;
            o.x = fmaf((v[j].x - mean) * rstd, g4[j].x, b4[j].x);
            o.y = fmaf((v[j].y - mean) * rstd, g4[j].y, b4[j].y);
            o.z = fmaf((v[j].z - mean) * rstd, g4[j].z, b4[j].z);
            o.w = fmaf((v[j].w - mean) * rstd, g4[j].w, b4[j].w);
            xw4[i] = o;
        }
    }
}

// One workgroup = 1..4 waves (M == 1: gemv_plan picks one wave for narrow matrices so that the dispatcher spreads
// them over all CUs, two / four where the waves share one input strip; M > 1: four); each wave owns rows
// [gw * rows_per_wave, +rows_per_wave).
// LPR lanes share one row (RPP = 64 / LPR rows per pass); CPL 16-B chunks per lane per row.
//
// Every kernel of a decode step except lm_head is bound by its chain of dependent memory round
// trips, not by bandwidth, so the structure minimises that chain:
//   * the hot scalars (W, x, N, K, ...) are separate leading kernel arguments so that they are
//     pre-loaded into SGPRs with the wave (kernarg preload) instead of fetched by s_load;
//   * the first pass of weights is requested before anything else;
//   * M == 1: each WAVE builds the transformed input row (LayerNorm / head merge) for itself in a
//     private LDS strip with wave-level reductions only — no workgroup barrier, and the input is
//     fetched 4x per workgroup instead of once per 16-lane group (which made hundreds of waves
//     hammer the same few cache lines).  M > 1: one cooperative build per workgroup;
//   * bias / residual operands of a row travel with the row's weights;
//   * passes are software-pipelined one deep (next pass in flight while this one is reduced).
template <typename WT, int MT, int LPR, int CPL, bool ARGMAX>
__global__ __launch_bounds__(256) void gemv_kernel(const void* __restrict__ Wv, const float* __restrict__ xin,
                                                   int N, int K, unsigned mpew, int rows_per_wave,
                                                   const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                   const int* __restrict__ cw, const GemvArgs a) {
    // mpew = M | prologue << 4 | epilogue << 8 | waves per workgroup << 12 (blockDim is a scalar load from the kernarg
    // segment: zg_common.h ZG_PIN); cw = the step control block, always a readable address
    const int M = (int)(mpew & 15u), prologue = (int)((mpew >> 4) & 15u), epilogue = (int)((mpew >> 8) & 15u);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RPP = 64 / LPR;
    constexpr bool XREG = (MT == 1) && (CPL <= 8);  // input row cached in registers
    constexpr bool PERWAVE = (MT == 1);             // wave-private prologue, no barrier
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane % LPR, rsub = lane / LPR;
    const int nch = K >> 3, nq = K >> 2;
    const WT* W = reinterpret_cast<const WT*>(Wv);
    // M == 1: waves are independent (wave-private prologue), so the workgroup may be 1..4 waves: matrices with
    // few rows are launched as one-wave workgroups that the dispatcher spreads over all CUs
    const int wpw = PERWAVE ? (int)(mpew >> 12) : 4;
    // A wide un-normalised input (mlp c_proj: K = 4 E) is as many bytes per wave as the wave's weight rows, so the
    // waves of a workgroup share ONE copy of it (a quarter of the fetch each, one barrier); everything else
    // keeps wave-private strips and no barrier.
    const bool shared_x = PERWAVE && wpw > 1 && (prologue == PRO_NONE || prologue == PRO_ATTN_MERGE);
    float* xs = (PERWAVE && !shared_x) ? smem + (size_t)wave * K : smem;  // [MT][K] (per wave when M == 1)
    float* red = smem + (size_t)(PERWAVE ? wpw : MT) * K;           // cross-wave scratch

    const int gw = blockIdx.x * wpw + wave;
    const int row_begin = gw * rows_per_wave;
    const int row_end = min(row_begin + rows_per_wave, N);

    // ---- 0. first pass of weights (+ its epilogue operands): independent of every other input
    Raw<WT> wa[CPL], wb[CPL];
    RowExtra<MT> ea, eb;
    ZG_STAMP_DECL();
    ZG_STAMP(0);
    load_pass<WT, LPR, CPL>(wa, W, K, nch, row_begin + rsub, row_end, lr);
    ea = load_extra<MT>(a, epilogue, M, N, row_begin + rsub);

    // position-dependent scalar (consumed late: merge split count when t_hi == 0, KV scatter position)
    const int T = max(cw[1], 1);
    {   // the argument-block fields of the tail, fetched under the first weight loads (zg_common.h ZG_PIN)
        ZG_PIN(a.progress); ZG_PIN(a.epilogue); ZG_PIN(a.y); ZG_PIN(a.y_stride); ZG_PIN(__float_as_uint(a.eps));
        if (ARGMAX) {
            ZG_PIN(a.logits); ZG_PIN(a.logits_stride); ZG_PIN(a.part_val); ZG_PIN(a.part_idx); ZG_PIN(gridDim.x);
        }
    }
    pf_count(a.progress);
    ZG_STAMP(1);

    // ---- 1. prologue: build the (transformed) input rows in LDS
    if constexpr (PERWAVE) {
        f32x4* xw4 = reinterpret_cast<f32x4*>(xs);
        if (prologue == PRO_LAYERNORM && nq <= 512) {
            if (nq <= 192) ln_strip<3>(xin, ln_g, ln_b, xw4, nq, K, a.eps, lane);
            else if (nq <= 256) ln_strip<4>(xin, ln_g, ln_b, xw4, nq, K, a.eps, lane);
            else ln_strip<8>(xin, ln_g, ln_b, xw4, nq, K, a.eps, lane);
            ZG_STAMP(2);
        } else if (prologue == PRO_LAYERNORM) {
            float t1 = 0.0f, t2 = 0.0f;
            for (int k = lane; k < K; k += 64) {
                const float val = xin[k];
                xs[k] = val;
                t1 += val;
                t2 = fmaf(val, val, t2);
            }
            t1 = wave_allsum(t1);
            t2 = wave_allsum(t2);
            const float mean = t1 / (float)K;
            const float rstd = 1.0f / sqrtf(t2 / (float)K - mean * mean + a.eps);
            for (int k = lane; k < K; k += 64) xs[k] = fmaf((xs[k] - mean) * rstd, ln_g[k], ln_b[k]);
        } else if (prologue == PRO_ATTN_MERGE && shared_x) {
            // the head merge is spread over the whole workgroup: one float4 of the merged vector per thread
            const int t_hi = a.t_hi > 0 ? a.t_hi : T;
            const int nsplit = (t_hi + kAttnChunk - 1) / kAttnChunk;
            for (int i = tid; i < nq; i += 64 * wpw) xw4[i] = merge_attn4(a, 0, i * 4, nsplit);
            __syncthreads();
        } else if (prologue == PRO_ATTN_MERGE) {
            const int t_hi = a.t_hi > 0 ? a.t_hi : T;
            const int nsplit = (t_hi + kAttnChunk - 1) / kAttnChunk;
            for (int i = lane; i < nq; i += 64) xw4[i] = merge_attn4(a, 0, i * 4, nsplit);
        } else if (shared_x) {
            const int nthr = 64 * wpw;
            for (int base = 0; base < nq; base += 8 * nthr) {  // 8 loads per thread in flight at once
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const f32x4*>(xin)[min(base + tid + nthr * j, nq - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (base + tid + nthr * j < nq) xw4[base + tid + nthr * j] = v[j];
            }
            __syncthreads();
        } else {
            for (int i = lane; i < nq; i += 64) xw4[i] = reinterpret_cast<const f32x4*>(xin)[i];
        }
        // same-wave LDS traffic is ordered: no barrier between the strip's writes and reads below
    } else {
        if (prologue == PRO_LAYERNORM && nq <= 512) {
            f32x4 v[MT][2], g4[2], b4[2];
            // branch-free (clamped) loads: predicated ones serialise into one round trip each
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 256 * j, ic = min(i, nq - 1);
                g4[j] = reinterpret_cast<const f32x4*>(ln_g)[ic];
                b4[j] = reinterpret_cast<const f32x4*>(ln_b)[ic];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    v[m][j] = reinterpret_cast<const f32x4*>(xin + (size_t)min(m, M - 1) * a.x_stride)[ic];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (tid + 256 * j >= nq || m >= M) v[m][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    t1 += v[m][j].x + v[m][j].y + v[m][j].z + v[m][j].w;
                    t2 = fmaf(v[m][j].x, v[m][j].x, fmaf(v[m][j].y, v[m][j].y, fmaf(v[m][j].z, v[m][j].z, fmaf(v[m][j].w, v[m][j].w, t2))));
                }
                t1 = wave_allsum(t1);
                t2 = wave_allsum(t2);
                if (lane == 0) {
                    red[(wave * MT + m) * 2] = t1;
                    red[(wave * MT + m) * 2 + 1] = t2;
                }
            }
            __syncthreads();
            ZG_STAMP(2);
            const float inv_k = 1.0f / (float)K;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float s1 = red[m * 2] + red[(MT + m) * 2] + red[(2 * MT + m) * 2] + red[(3 * MT + m) * 2];
                const float s2 = red[m * 2 + 1] + red[(MT + m) * 2 + 1] + red[(2 * MT + m) * 2 + 1] + red[(3 * MT + m) * 2 + 1];
                const float mean = s1 * inv_k;
                const float rstd = __builtin_amdgcn_rsqf(s2 * inv_k - mean * mean + a.eps);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = tid + 256 * j;
                    if (i < nq) {
                        f32x4 o;
                        o.x = fmaf((v[m][j].x - mean) * rstd, g4[j].x, b4[j].x);
                        o.y = fmaf((v[m][j].y - mean) * rstd, g4[j].y, b4[j].y);
                        o.z = fmaf((v[m][j].z - mean) * rstd, g4[j].z, b4[j].z);
                        o.w = fmaf((v[m][j].w - mean) * rstd, g4[j].w, b4[j].w);
                        reinterpret_cast<f32x4*>(xs + (size_t)m * K)[i] = o;
                    }
                }
            }
        } else if (prologue == PRO_LAYERNORM) {
            for (int m = wave; m < MT; m += 4) {  // wide rows: one wave per row, two sweeps
                float* xm = xs + (size_t)m * K;
                const float* x = xin + (size_t)m * a.x_stride;
                float t1 = 0.0f, t2 = 0.0f;
                for (int k = lane; k < K; k += 64) {
                    const float val = (m < M) ? x[k] : 0.0f;
                    xm[k] = val;
                    t1 += val;
                    t2 = fmaf(val, val, t2);
                }
                t1 = wave_allsum(t1);
                t2 = wave_allsum(t2);
                const float mean = t1 / (float)K;
                const float rstd = 1.0f / sqrtf(t2 / (float)K - mean * mean + a.eps);
                for (int k = lane; k < K; k += 64) xm[k] = fmaf((xm[k] - mean) * rstd, ln_g[k], ln_b[k]);
            }
        } else if (prologue == PRO_ATTN_MERGE) {
            const int t_hi = a.t_hi > 0 ? a.t_hi : T;
            const int nsplit = (t_hi + kAttnChunk - 1) / kAttnChunk;
            for (int i = tid; i < nq; i += 256) {
                f32x4 o[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) o[m] = merge_attn4(a, min(m, M - 1), i * 4, nsplit);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    reinterpret_cast<f32x4*>(xs + (size_t)m * K)[i] = (m < M) ? o[m] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        } else {
            for (int i = tid; i < nq; i += 256) {
                f32x4 o[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) o[m] = reinterpret_cast<const f32x4*>(xin + (size_t)min(m, M - 1) * a.x_stride)[i];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    reinterpret_cast<f32x4*>(xs + (size_t)m * K)[i] = (m < M) ? o[m] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
        __syncthreads();
    }
    ZG_STAMP(3);

    W8 xr[XREG ? CPL : 1];
    if constexpr (XREG) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lr + LPR * i;
            xr[i] = (c < nch) ? load_x8(xs + c * 8) : zero_w8();
        }
    }
    ZG_STAMP(4);

    // ---- 2. rows, software-pipelined one pass (RPP rows) deep
    Best best[ARGMAX ? MT : 1];
#pragma unroll
    for (int m = 0; m < (ARGMAX ? MT : 1); ++m) {
        best[m].val = -3.0e38f;
        best[m].idx = 0x7fffffff;
    }
    const int pos = T - 1;

    auto do_pass = [&](const Raw<WT>(&w)[CPL], const RowExtra<MT>& ex, int r, bool valid) {
        float acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = 0.0f;
        if constexpr (XREG) {
            // four independent partial sums: a single accumulator is one 8*CPL-long dependent FMA chain,
            // and with one wave per SIMD (small grids) nothing else hides the VALU latency
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const W8 u = unpack(w[i]);
                p0 = fmaf(u.v[0], xr[i].v[0], p0); p1 = fmaf(u.v[1], xr[i].v[1], p1);
                p2 = fmaf(u.v[2], xr[i].v[2], p2); p3 = fmaf(u.v[3], xr[i].v[3], p3);
                p0 = fmaf(u.v[4], xr[i].v[4], p0); p1 = fmaf(u.v[5], xr[i].v[5], p1);
                p2 = fmaf(u.v[6], xr[i].v[6], p2); p3 = fmaf(u.v[7], xr[i].v[7], p3);
            }
            acc[0] = (p0 + p1) + (p2 + p3);
        } else {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = lr + LPR * i;
                if (c < nch) {
                    const W8 u = unpack(w[i]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = dot8(u, load_x8(xs + (size_t)m * K + c * 8), acc[m]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = group_allsum<LPR>(acc[m]);
        if (lr == 0 && valid) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < M) epilogue_row(a, m, r, acc[m], ex.bias, ex.resid[m], pos, best[ARGMAX ? m : 0]);
        }
    };

    if constexpr (XREG) {
        for (int rb = row_begin; rb < row_end; rb += 2 * RPP) {
            const int r0 = rb + rsub, r1 = rb + RPP + rsub, r2 = rb + 2 * RPP + rsub;
            load_pass<WT, LPR, CPL>(wb, W, K, nch, r1, row_end, lr);
            eb = load_extra<MT>(a, epilogue, M, N, r1);
            do_pass(wa, ea, r0, r0 < row_end);
            ZG_STAMP(5);
            load_pass<WT, LPR, CPL>(wa, W, K, nch, r2, row_end, lr);
            ea = load_extra<MT>(a, epilogue, M, N, r2);
            do_pass(wb, eb, r1, r1 < row_end);
            ZG_STAMP(6);
        }
    } else {
        // Batched / wide-K path: the input rows live in LDS, and LDS read bandwidth is what bounds it, so
        // two weight rows share every input chunk that is read (halves the ds_read traffic per weight).
        load_pass<WT, LPR, CPL>(wb, W, K, nch, row_begin + RPP + rsub, row_end, lr);
        eb = load_extra<MT>(a, epilogue, M, N, row_begin + RPP + rsub);
        for (int rb = row_begin; rb < row_end; rb += 2 * RPP) {
            const int r0 = rb + rsub, r1 = rb + RPP + rsub;
            float acc0[MT], acc1[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc0[m] = acc1[m] = 0.0f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = lr + LPR * i;
                if (c < nch) {
                    const W8 u0 = unpack(wa[i]), u1 = unpack(wb[i]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const W8 x = load_x8(xs + m * K + c * 8);
                        acc0[m] = dot8(u0, x, acc0[m]);
                        acc1[m] = dot8(u1, x, acc1[m]);
                    }
                }
            }
            const RowExtra<MT> e0 = ea, e1 = eb;
            // next two rows' weights: requested before the reductions / epilogue of this pair
            load_pass<WT, LPR, CPL>(wa, W, K, nch, r0 + 2 * RPP, row_end, lr);
            ea = load_extra<MT>(a, epilogue, M, N, r0 + 2 * RPP);
            load_pass<WT, LPR, CPL>(wb, W, K, nch, r1 + 2 * RPP, row_end, lr);
            eb = load_extra<MT>(a, epilogue, M, N, r1 + 2 * RPP);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc0[m] = group_allsum<LPR>(acc0[m]);
                acc1[m] = group_allsum<LPR>(acc1[m]);
            }
            if (lr == 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (m < M) {
                        if (r0 < row_end) epilogue_row(a, m, r0, acc0[m], e0.bias, e0.resid[m], pos, best[ARGMAX ? m : 0]);
                        if (r1 < row_end) epilogue_row(a, m, r1, acc1[m], e1.bias, e1.resid[m], pos, best[ARGMAX ? m : 0]);
                    }
            }
        }
    }

    ZG_STAMP(7);
    ZG_STAMP_FLUSH();
    // ---- 3. argmax partials
    if constexpr (ARGMAX) {
        __syncthreads();  // the per-wave strips may still be read by slower waves
        float* s_val = red;
        int* s_idx = reinterpret_cast<int*>(red + 4 * 8);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const Best b = wave_best(best[m]);
            if (lane == 0) {
                s_val[wave * 8 + m] = b.val;
                s_idx[wave * 8 + m] = b.idx;
            }
        }
        __syncthreads();
        if (tid < MT && tid < M) {
            const int m = tid;
            Best b;
            b.val = s_val[m];
            b.idx = s_idx[m];
            for (int w = 1; w < 4; ++w) {
                Best o;
                o.val = s_val[w * 8 + m];
                o.idx = s_idx[w * 8 + m];
                b = better(b, o);
            }
            a.part_val[(size_t)m * gridDim.x + blockIdx.x] = b.val;
            a.part_idx[(size_t)m * gridDim.x + blockIdx.x] = b.idx;
        }
    }
}

// ================================================================================================
// M == 1, wide un-normalised input (mlp c_proj: K = 4 E): the four waves of a workgroup SPLIT K.
//
// In the kernel above a wide input row is as many bytes per wave as the wave's weight rows, so it went through a
// shared LDS strip behind a barrier — a memory round trip, an LDS round trip and a barrier in front of the first
// FMA.  Here wave w owns columns [w K/4, (w+1) K/4) of every row of the workgroup: its quarter of the input goes
// straight from global memory into registers (fetched next to the weights, no LDS, no barrier), every wave streams
// the same 2 * RPP rows (quarter-row segments of >= 1.5 KB, fully coalesced), and the four partial sums per row
// meet in LDS after the arithmetic, where one thread per row runs the epilogue.
template <typename WT, int LPR, int CPL, int NP = 2>  // NP passes of 64 / LPR rows per workgroup
__global__ __launch_bounds__(256) void gemv_ksplit_kernel(const void* __restrict__ Wv, const float* __restrict__ xin, int N,
                                                          int K, unsigned em, const float* __restrict__ part_in, int max_splits,
                                                          const float* __restrict__ bias, const float* __restrict__ resid,
                                                          const GemvArgs a) {
    // 14 preloaded dwords: Wv, xin, N, K, em = epilogue | merge_splits << 8 | has_bias << 16 | has_resid << 17 | xg_resid << 18
    // | xg_out << 19, the attention partials and their split stride, bias and residual (a few zero floats when absent:
    // read at index 0)
    const int epilogue = (int)(em & 0xffu), merge_splits = (int)((em >> 8) & 0xffu);
    const int has_bias = (int)((em >> 16) & 1u), has_resid = (int)((em >> 17) & 1u);
    const int xg_res = (int)((em >> 18) & 1u), xg_out = (int)((em >> 19) & 1u);  // two-stream decode: GemvArgs.xg
    const int in_gran = (int)((em >> 20) & 1u);                                  // ... the input arrives as granules (xin = in_g)
    ZG_STAMP_DECL();
    ZG_STAMP(0);
    // merge_splits > 0 (attn c_proj): the input is the head merge of the attention partials — every lane combines
    // the <= 4 split partials of ITS OWN 8-element chunks (a chunk lies inside one head), all loads issued with the
    // weights; no shared strip, no barrier in front of the FMAs (the merge through an LDS strip cost 3.9 us per
    // launch against 2.55 us for the plain K-split kernel).
    constexpr int RPP = 64 / LPR, ROWS = NP * RPP;
    __shared__ float part[4][ROWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane % LPR, rsub = lane / LPR;
    const int Kq = K >> 2, nchq = Kq >> 3;
    const WT* W = reinterpret_cast<const WT*>(Wv) + (size_t)wave * Kq;
    const int row0 = blockIdx.x * ROWS;
    // all loads of the kernel up front: NP passes of weights, the input quarter, the epilogue operands
    Raw<WT> wq[NP][CPL];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const WT* pr = W + (size_t)min(row0 + p * RPP + rsub, N - 1) * K;
#pragma unroll
        for (int i = 0; i < CPL; ++i) wq[p][i] = load_raw(pr, min(lr + LPR * i, nchq - 1));
    }
    W8 xr[CPL];
    if (merge_splits > 0) {
        constexpr int MAXS = 4;
        const int nsplit = merge_splits;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int e0 = wave * Kq + min(lr + LPR * i, nchq - 1) * 8;
            const int h = e0 >> 6, d0 = e0 & 63;  // head_dim 64
            const float* p = part_in + ((size_t)h * max_splits) * kPartStride;
            float ms[MAXS], ls[MAXS];
            W8 o[MAXS];
#pragma unroll
            for (int sp = 0; sp < MAXS; ++sp) {  // branch-free: surplus splits re-read the last valid one, weight 0 below
                const float* ps = p + min(sp, nsplit - 1) * kPartStride;
                ms[sp] = ps[64];
                ls[sp] = ps[65];
                const float2 a0 = *reinterpret_cast<const float2*>(ps + d0), a1 = *reinterpret_cast<const float2*>(ps + d0 + 2);
                const float2 a2 = *reinterpret_cast<const float2*>(ps + d0 + 4), a3 = *reinterpret_cast<const float2*>(ps + d0 + 6);
                o[sp].v[0] = a0.x; o[sp].v[1] = a0.y; o[sp].v[2] = a1.x; o[sp].v[3] = a1.y;
                o[sp].v[4] = a2.x; o[sp].v[5] = a2.y; o[sp].v[6] = a3.x; o[sp].v[7] = a3.y;
            }
#pragma unroll
            for (int sp = 0; sp < MAXS; ++sp)
                if (sp >= nsplit) ms[sp] = -1e30f;
            const float mx = fmaxf(fmaxf(ms[0], ms[1]), fmaxf(ms[2], ms[3]));
            float l = 0.0f;
            W8 r = zero_w8();
#pragma unroll
            for (int sp = 0; sp < MAXS; ++sp) {
                const float w = __expf(ms[sp] - mx);
                l = fmaf(w, ls[sp], l);
#pragma unroll
                for (int j = 0; j < 8; ++j) r.v[j] = fmaf(w, o[sp].v[j], r.v[j]);
            }
            const float inv = 1.0f / l;
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[i].v[j] = r.v[j] * inv;
        }
    } else if (in_gran) {
        // the input as (value, tag) granules from the LayerNorm-fed kernel on another stream (see gemv_lnk_kernel): the wave
        // watches one granule of its K quarter, then every lane checks the chunks it multiplies
        const unsigned want = (*a.epoch2 << 8) | a.xin_id;
        const unsigned long long* gin = reinterpret_cast<const unsigned long long*>(xin);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (unsigned)K * 8u, 0x00020000);
        unsigned spins = 0;
        while ((unsigned)(__hip_atomic_load(gin + wave * Kq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != want && spins < a.spin_limit) {
            __builtin_amdgcn_s_sleep(4);
            ++spins;
        }
        for (;; ++spins) {
            asm volatile("" ::: "memory");  // (plain intrinsics below: keep them inside the loop)
            bool ok = true;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const unsigned off = ((unsigned)(wave * Kq) + (unsigned)min(lr + LPR * i, nchq - 1) * 8u) * 8u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 gq = __builtin_amdgcn_raw_buffer_load_b128(rx, off + 16u * j, 0, 16);  // sc1
                    ok = ok && gq.y == want && gq.w == want;
                    xr[i].v[2 * j] = __uint_as_float(gq.x);
                    xr[i].v[2 * j + 1] = __uint_as_float(gq.z);
                }
            }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
            if (spins >= a.spin_limit) {
                if (lane == 0 && a.fault) __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPL; ++i) xr[i] = load_x8(xin + (size_t)wave * Kq + (size_t)min(lr + LPR * i, nchq - 1) * 8);
    }
    float bias_n = 0.0f, resid_n = 0.0f;
    unsigned long long resid_g = 0;
    // (the granule buffer, or the zero words as its stand-in: both loads are unconditional — a load inside a uniform branch
    // whose result is merged with a constant makes the compiler wait at the join for every load issued before it)
    const unsigned long long* xgp = xg_res ? a.xg : reinterpret_cast<const unsigned long long*>(a.zero);
    if (tid < ROWS) {
        const int n = min(row0 + tid, N - 1);
        bias_n = bias[n * has_bias];
        resid_n = resid[n * (has_resid & (xg_res ^ 1))];
        resid_g = __hip_atomic_load(xgp + n * xg_res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ZG_STAMP(1);
    ZG_PIN(a.y);  // the tail's argument-block fields, fetched under the vector loads (zg_common.h ZG_PIN)
    ZG_PIN(a.xg); ZG_PIN(a.epoch2); ZG_PIN(a.xout_id);
    ZG_PIN(a.progress);
    pf_count(a.progress);
    ZG_STAMP(2);
#pragma unroll
    for (int i = 0; i < CPL; ++i)
        if (lr + LPR * i >= nchq) xr[i] = zero_w8();  // clamped surplus chunks multiply zeros
    auto dot = [&](const Raw<WT>(&w)[CPL]) {
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const W8 u = unpack(w[i]);
            p0 = fmaf(u.v[0], xr[i].v[0], p0); p1 = fmaf(u.v[1], xr[i].v[1], p1);
            p2 = fmaf(u.v[2], xr[i].v[2], p2); p3 = fmaf(u.v[3], xr[i].v[3], p3);
            p0 = fmaf(u.v[4], xr[i].v[4], p0); p1 = fmaf(u.v[5], xr[i].v[5], p1);
            p2 = fmaf(u.v[6], xr[i].v[6], p2); p3 = fmaf(u.v[7], xr[i].v[7], p3);
        }
        return group_allsum<LPR>((p0 + p1) + (p2 + p3));
    };
    float sp[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) sp[p] = dot(wq[p]);
    ZG_STAMP(3);
    if (lr == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) part[wave][p * RPP + rsub] = sp[p];
    }
    __syncthreads();
    ZG_STAMP(4);
    if (tid < ROWS && row0 + tid < N) {
        const int n = row0 + tid;
        const float v = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + bias_n;
        if (xg_res) resid_n = __uint_as_float((unsigned)resid_g);
        const float out = epilogue == EPI_RESIDUAL ? v + resid_n : (epilogue == EPI_GELU ? gelu_ref(v) : v);
        if (xg_out) {  // (value, tag) in one 8-byte agent-scope store: the consumer on the other stream polls the tag
            const unsigned long long tg = (unsigned long long)((*a.epoch2 << 8) | a.xout_id) << 32;
            __hip_atomic_store(a.xg + n, tg | __float_as_uint(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
            a.y[n] = out;
    }
    ZG_STAMP(5);
    ZG_STAMP(6);
    ZG_STAMP(7);
    ZG_STAMP_FLUSH();
}

template <typename WT>
int launch_ksplit(const GemvArgs& a, hipStream_t s) {
    const int nchq = a.K / 32;  // 16-B chunks per quarter row
    const int merge_splits = a.prologue == PRO_ATTN_MERGE ? (a.t_hi + kAttnChunk - 1) / kAttnChunk : 0;
    const unsigned has_resid = a.epilogue == EPI_RESIDUAL ? 1u : 0u;
    const unsigned xg_res = (a.xg && a.xg_resid && has_resid) ? 1u : 0u, xg_out = (a.xg && a.xout_id) ? 1u : 0u;
    const unsigned in_gran = (a.in_g && merge_splits == 0) ? 1u : 0u;
    const unsigned em = (unsigned)a.epilogue | ((unsigned)merge_splits << 8) | ((a.bias ? 1u : 0u) << 16) | (has_resid << 17) | (xg_res << 18) |
                        (xg_out << 19) | (in_gran << 20);
    // two passes of 64 / LPR rows per workgroup (four measured slower: 2.65 -> 3.3 us for mlp c_proj)
#define ZG_KS(LPR_, CPL_)                                                                                                \
    {                                                                                                                    \
        constexpr int rows = 2 * (64 / LPR_);                                                                            \
        note_kernel("gemv_ksplit_kernel<%s, %d, %d, 2>", sizeof(WT) == 2 ? "unsigned short" : "float", LPR_, CPL_);         \
        hipLaunchKernelGGL((gemv_ksplit_kernel<WT, LPR_, CPL_, 2>), dim3((a.N + rows - 1) / rows), dim3(256), 0, s, a.W,     \
                           in_gran ? reinterpret_cast<const float*>(a.in_g) : a.x,                                      \
                           a.N, a.K, em, a.part ? a.part : a.zero, a.max_splits, a.bias ? a.bias : a.zero,              \
                           (has_resid && !xg_res) ? a.resid : a.zero, a);                                                \
        ZG_HIP(hipGetLastError());                                                                                       \
        return ZG_OK;                                                                                                    \
    }
    if (nchq <= 16 * 2) ZG_KS(16, 2)
    if (nchq <= 32 * 3) ZG_KS(32, 3)
    if (nchq <= 32 * 5) ZG_KS(32, 5)
    if (nchq <= 32 * 7) ZG_KS(32, 7)
    if (nchq <= 64 * 4) ZG_KS(64, 4)
#undef ZG_KS
    zg::set_error("gemv (K split): K=%d too large", a.K);
    return ZG_ERR_UNSUPPORTED;
}

// M == 1 plain Linear over a wide input: the K-split kernel (measured against the shared-strip form in situ)
bool gemv_use_ksplit(const GemvArgs& a) {
    static const int off = getenv("ZGPT2_NO_KSPLIT") ? atoi(getenv("ZGPT2_NO_KSPLIT")) : 0;
    static const int min_k = getenv("ZGPT2_KSPLIT_MIN_K") ? atoi(getenv("ZGPT2_KSPLIT_MIN_K")) : 2048;
    if (off || a.M != 1) return false;
    if (a.epilogue != EPI_STORE && a.epilogue != EPI_RESIDUAL && a.epilogue != EPI_GELU) return false;
    if (a.prologue == PRO_ATTN_MERGE)  // head merge folded into the lanes' own chunks: model tier, <= 4 splits known at launch
        return a.head_dim == 64 && a.t_hi > 0 && (a.t_hi + kAttnChunk - 1) / kAttnChunk <= 4 && a.K % 32 == 0 && a.K <= 1024;  // wider rows (XL, K = 1600: three chunks per lane) measured slower than the shared strip
    if (a.prologue != PRO_NONE) return false;
    return a.K >= min_k && a.K % 32 == 0 && a.K / 32 <= 256;
}

// ================================================================================================
// M == 1, LayerNorm in front (ln_1 + c_attn, ln_2 + c_fc): the LayerNorm is LINEARISED out of the dot product.
//
//   y_n = sum_k W_nk ((x_k - mu) r g_k + b_k) + bias_n  =  r (S1_n - mu c2_n) + c3_n
//   S1_n = sum_k W_nk (g_k x_k),   c2_n = sum_k W_nk g_k,   c3_n = sum_k W_nk b_k + bias_n
//
// c2 / c3 depend on the weights only (launch_ln_fold, once after loading); S1 needs no statistics, so the kernel
// has the shape of the K-split kernel above — every load issued at entry, wave w owns K quarter w, FMAs straight
// from registers — and mu, r (single pass sum / sum of squares, std = sqrt(E[x^2] - mean^2 + eps): ops.zig:88-101)
// are needed only by the one thread per row that combines the four partial sums.  The kernel it replaces spent a
// third of its time in the dependent chain load x -> two wave reductions -> normalise -> LDS -> registers before
// its first FMA.  (Same real-number result; in floating point r (S1 - mu c2) cancels when |mu| >> sigma, which costs
// log2(|mu| / sigma) bits of the fp32 product sums — far inside the 1e-3 bound for any LayerNorm input.)
template <typename WT, int LPR, int CPL, int NP = 2, bool XG = false>  // NP passes of 64 / LPR rows per workgroup; XG: x as granules
__global__ __launch_bounds__(256) void gemv_lnk_kernel(const void* __restrict__ Wv, const float* __restrict__ xin, unsigned ne, int K,
                                                       const float* __restrict__ ln_g, const float* __restrict__ c2,
                                                       const float* __restrict__ c3, const int* __restrict__ cw,
                                                       const GemvArgs a) {
    // 14 preloaded dwords: Wv, xin, ne = N | epilogue << 24, K, ln_g, c2, c3, cw = the step control block (always a
    // readable address: its second word is the sequence length of the KV append)
    const int N = (int)(ne & 0xffffffu), epilogue = (int)(ne >> 24);
    ZG_STAMP_DECL();
    ZG_STAMP(0);
    constexpr int RPP = 64 / LPR, ROWS = NP * RPP;
    __shared__ float part[4][ROWS];
    __shared__ float stat[4][2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane % LPR, rsub = lane / LPR;
    // The wave's share of K.  bf16 rows of whole 128-byte lines (K % 64 == 0): whole lines per wave — 7, 6, 6, 6 of
    // the 25 at K = 1600 instead of four times 6.25, whose quarters begin mid-line and make every wave touch the
    // boundary lines of its neighbours as well (31 line touches per row instead of 25).
    int kbeg, nchq;
    if (sizeof(WT) == 2 && (K & 63) == 0 && ((K >> 6) & 3) != 0 && (((K >> 6) >> 2) + 1) * 8 <= LPR * CPL) {
        const int lines = K >> 6, base = lines >> 2, rem = lines & 3;
        kbeg = (wave * base + min(wave, rem)) * 64;
        nchq = (base + (wave < rem ? 1 : 0)) * 8;
    } else {
        kbeg = wave * (K >> 2);
        nchq = K >> 5;
    }
    const WT* W = reinterpret_cast<const WT*>(Wv) + kbeg;
    const int row0 = blockIdx.x * ROWS;
    Raw<WT> wq[NP][CPL];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const WT* pr = W + (size_t)min(row0 + p * RPP + rsub, N - 1) * K;
#pragma unroll
        for (int i = 0; i < CPL; ++i) wq[p][i] = load_raw(pr, min(lr + LPR * i, nchq - 1));
    }
    W8 xr[CPL], gr[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const size_t off = (size_t)kbeg + (size_t)min(lr + LPR * i, nchq - 1) * 8;
        if constexpr (!XG) xr[i] = load_x8(xin + off);
        gr[i] = load_x8(ln_g + off);
    }
    float c2n = 0.0f, c3n = 0.0f;
    if (tid < ROWS) {
        const int n = min(row0 + tid, N - 1);
        c2n = c2[n];
        c3n = c3[n];
    }
    const int T = max(cw[1], 1);  // KV append position (EPI_QKV)
    if constexpr (XG) {
        // Two-stream decode: x arrives as (value, tag) granules from a kernel of the OTHER stream, possibly still running — this
        // kernel was launched beside it and has its weights and LayerNorm vectors in flight.  xin = the granules; every lane
        // polls the chunks it multiplies (16-byte agent-scope loads of two granules, each granule one 8-byte store of its
        // writer) until all carry (epoch2 << 8 | xin_id); xin_id == 0: the input is known to be complete, one pass.
        const unsigned want = (*a.epoch2 << 8) | a.xin_id;
        const bool poll = a.xin_id != 0;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (unsigned)K * 8u, 0x00020000);
        // Cheap wait first: the whole wave watches ONE granule of its K quarter (a single request per poll) until it carries the
        // tag — with every lane of 1024 waves re-reading its 16 granules the pollers alone moved ~8 MB per round through the
        // L2s and slowed the producers they were waiting for (269 against 212 us per token).  The writers of x finish within
        // a fraction of a microsecond of each other, so the full check below then passes on its first or second round.
        unsigned spins = 0;
        if (poll) {
            const unsigned long long* g0 = reinterpret_cast<const unsigned long long*>(xin) + kbeg;
            while ((unsigned)(__hip_atomic_load(g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != want && spins < a.spin_limit) {
                if (a.xout_id >= 4) __builtin_amdgcn_s_sleep(8);       // (xout_id is unused by this kernel: carries the A/B knob
                else if (a.xout_id >= 2) __builtin_amdgcn_s_sleep(4);  //  ZGPT2_DUAL_SLEEP of the poll pause)
                else if (a.xout_id >= 1) __builtin_amdgcn_s_sleep(2);
                else __builtin_amdgcn_s_sleep(1);
                ++spins;
            }
        }
        for (;; ++spins) {
            asm volatile("" ::: "memory");  // (the loads below are plain intrinsics: without this the compiler hoists them out of the loop)
            u32x4 gq[CPL][4];
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const unsigned off = ((unsigned)kbeg + (unsigned)min(lr + LPR * i, nchq - 1) * 8u) * 8u;
#pragma unroll
                for (int j = 0; j < 4; ++j) gq[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + 16u * j, 0, 16);  // sc1
            }
            bool ok = true;
#pragma unroll
            for (int i = 0; i < CPL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ok = ok && gq[i][j].y == want && gq[i][j].w == want;
                    xr[i].v[2 * j] = __uint_as_float(gq[i][j].x);
                    xr[i].v[2 * j + 1] = __uint_as_float(gq[i][j].z);
                }
            if (!poll || __builtin_amdgcn_ballot_w64(!ok) == 0) break;
            if (spins >= a.spin_limit) {  // bounded: never hang the queue — and never pass silently
                if (lane == 0 && a.fault) __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    ZG_STAMP(1);
    {   // the argument-block fields of the tail, fetched under the vector loads (zg_common.h ZG_PIN)
        ZG_PIN(a.progress); ZG_PIN(a.y); ZG_PIN(a.y_stride); ZG_PIN(a.epilogue); ZG_PIN(__float_as_uint(a.eps));
        if constexpr (XG) { ZG_PIN(a.fault); ZG_PIN(a.spin_limit); }
        if (epilogue == EPI_QKV) {
            ZG_PIN(a.q); ZG_PIN(a.k_cache); ZG_PIN(a.v_cache); ZG_PIN(a.N); ZG_PIN(a.head_dim); ZG_PIN(a.n_heads); ZG_PIN(a.ctx); ZG_PIN(a.kv_f16);
        }
    }
    pf_count(a.progress);
    ZG_STAMP(2);
    // statistics of this wave's quarter (every LPR-lane group holds the whole quarter) and z = g x
    float sx = 0.0f, sxx = 0.0f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        if (lr + LPR * i >= nchq) xr[i] = zero_w8();  // clamped surplus chunks contribute nothing
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sx += xr[i].v[j];
            sxx = fmaf(xr[i].v[j], xr[i].v[j], sxx);
            xr[i].v[j] *= gr[i].v[j];
        }
    }
    sx = group_allsum<LPR>(sx);
    sxx = group_allsum<LPR>(sxx);
    auto dot = [&](const Raw<WT>(&w)[CPL]) {
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const W8 u = unpack(w[i]);
            p0 = fmaf(u.v[0], xr[i].v[0], p0); p1 = fmaf(u.v[1], xr[i].v[1], p1);
            p2 = fmaf(u.v[2], xr[i].v[2], p2); p3 = fmaf(u.v[3], xr[i].v[3], p3);
            p0 = fmaf(u.v[4], xr[i].v[4], p0); p1 = fmaf(u.v[5], xr[i].v[5], p1);
            p2 = fmaf(u.v[6], xr[i].v[6], p2); p3 = fmaf(u.v[7], xr[i].v[7], p3);
        }
        return group_allsum<LPR>((p0 + p1) + (p2 + p3));
    };
    float sp[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) sp[p] = dot(wq[p]);
    ZG_STAMP(3);
    if (lr == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) part[wave][p * RPP + rsub] = sp[p];
    }
    if (lane == 0) {
        stat[wave][0] = sx;
        stat[wave][1] = sxx;
    }
    __syncthreads();
    ZG_STAMP(4);
    if (tid < ROWS && row0 + tid < N) {
        const int n = row0 + tid;
        const float inv_k = 1.0f / (float)K;
        const float mean = ((stat[0][0] + stat[1][0]) + (stat[2][0] + stat[3][0])) * inv_k;
        const float ex2 = ((stat[0][1] + stat[1][1]) + (stat[2][1] + stat[3][1])) * inv_k;
        const float rstd = __builtin_amdgcn_rsqf(ex2 - mean * mean + a.eps);
        const float S1 = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        const float y = fmaf(rstd, fmaf(-mean, c2n, S1), c3n);
        Best nobest;
        epilogue_row(a, 0, n, y, 0.0f, 0.0f, T - 1, nobest);
    }
    ZG_STAMP(5);
    ZG_STAMP(6);
    ZG_STAMP(7);
    ZG_STAMP_FLUSH();
}

// c2[n] = sum_k W[n][k] g[k], c3[n] = sum_k W[n][k] b[k] + bias[n]: one wave per row, fp32 accumulation.
template <typename WT>
__global__ __launch_bounds__(256) void ln_fold_kernel(const void* __restrict__ Wv, const float* __restrict__ g,
                                                      const float* __restrict__ b, const float* __restrict__ bias, int N, int K,
                                                      float* __restrict__ c2, float* __restrict__ c3) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const WT* w = reinterpret_cast<const WT*>(Wv) + (size_t)row * K;
    float s2 = 0.0f, s3 = 0.0f;
    for (int k = lane; k < K; k += 64) {
        float wv;
        if constexpr (sizeof(WT) == 2) wv = __uint_as_float((uint32_t)w[k] << 16);
        else wv = w[k];
        s2 = fmaf(wv, g[k], s2);
        s3 = fmaf(wv, b[k], s3);
    }
    s2 = wave_allsum(s2);
    s3 = wave_allsum(s3);
    if (lane == 0) {
        c2[row] = s2;
        c3[row] = s3 + (bias ? bias[row] : 0.0f);
    }
}

template <typename WT>
int launch_lnk(const GemvArgs& a, hipStream_t s) {
    const int nchq = a.K / 32;
    // four passes of 64 / LPR rows per workgroup (2.93 against 3.2 us per launch with two; +1 % tokens/s in situ)
#define ZG_LK(LPR_, CPL_)                                                                                              \
    {                                                                                                                  \
        constexpr int rows = 4 * (64 / LPR_);                                                                          \
        note_kernel("gemv_lnk_kernel<%s, %d, %d, 4%s>", sizeof(WT) == 2 ? "unsigned short" : "float", LPR_, CPL_, a.xg ? ", granules" : ""); \
        if (a.xg)                                                                                                      \
            hipLaunchKernelGGL((gemv_lnk_kernel<WT, LPR_, CPL_, 4, true>), dim3((a.N + rows - 1) / rows), dim3(256), 0, s, a.W, \
                               reinterpret_cast<const float*>(a.xg), (unsigned)a.N | ((unsigned)a.epilogue << 24), a.K, a.ln_g,  \
                               a.ln_c2, a.ln_c3, a.ctrl ? reinterpret_cast<const int*>(a.ctrl) : reinterpret_cast<const int*>(a.zero), a); \
        else                                                                                                           \
            hipLaunchKernelGGL((gemv_lnk_kernel<WT, LPR_, CPL_, 4>), dim3((a.N + rows - 1) / rows), dim3(256), 0, s, a.W, a.x, \
                               (unsigned)a.N | ((unsigned)a.epilogue << 24), a.K, a.ln_g, a.ln_c2, a.ln_c3,             \
                               a.ctrl ? reinterpret_cast<const int*>(a.ctrl) : reinterpret_cast<const int*>(a.zero), a); \
        ZG_HIP(hipGetLastError());                                                                                     \
        return ZG_OK;                                                                                                  \
    }
    if (nchq <= 16 * 2) ZG_LK(16, 2)
    if (nchq <= 32 * 2) ZG_LK(32, 2)
    if (nchq <= 32 * 3) ZG_LK(32, 3)
    if (nchq <= 64 * 2) ZG_LK(64, 2)
#undef ZG_LK
    zg::set_error("gemv (LayerNorm, K split): K=%d too large", a.K);
    return ZG_ERR_UNSUPPORTED;
}

bool gemv_use_lnk(const GemvArgs& a) {
    static const int off = getenv("ZGPT2_NO_LNK") ? atoi(getenv("ZGPT2_NO_LNK")) : 0;
    if (off || a.M != 1 || a.prologue != PRO_LAYERNORM || a.ln_c2 == nullptr || a.ln_c3 == nullptr) return false;
    if (a.epilogue != EPI_STORE && a.epilogue != EPI_GELU && a.epilogue != EPI_QKV) return false;
    return a.K % 32 == 0 && a.K / 32 <= 128 && a.N <= 16384;
}

// Slow generic fallback for K % 8 != 0 (op tier only): one wave per row, scalar loads.
template <typename WT>
__global__ __launch_bounds__(256) void gemv_generic_kernel(const GemvArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.N) return;
    const WT* w = reinterpret_cast<const WT*>(a.W) + (size_t)row * a.K;
    for (int m = 0; m < a.M; ++m) {
        const float* x = a.x + (size_t)m * a.x_stride;
        float acc = 0.0f;
        for (int k = lane; k < a.K; k += 64) {
            float wv;
            if constexpr (sizeof(WT) == 2) wv = __uint_as_float((uint32_t)w[k] << 16);
            else wv = w[k];
            acc = fmaf(wv, x[k], acc);
        }
        acc = wave_allsum(acc);
        if (lane == 0) a.y[(size_t)m * a.y_stride + row] = acc + (a.bias ? a.bias[row] : 0.0f);
    }
}


// ================================================================================================
// Batched decode (2 <= M <= 8 sequences in lock step) on the matrix cores.
//
// The VALU kernel above re-reads the M input rows from LDS for every weight chunk and ends up bound
// by LDS traffic and VGPRs (M = 8: 10-17 us per layer GEMV, 38 us for lm_head).  Here one
// v_mfma_f32_16x16x32_bf16 multiplies 16 weight rows by the (padded) batch for 32 k at once:
//   B operand = 8 consecutive k of weight row n0 + (lane & 15)  -> one 16-B global load per lane,
//               straight from the bf16 [N, K] matrix (ops.Linear.weight layout), no staging;
//   A operand = the input rows, kept in LDS as THREE bf16 planes hi + mid + lo with
//               hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 mantissa bits carry
//               the full fp32 value, bf16 x bf16 products are exact in fp32 and the MFMA accumulates
//               in fp32, so the result has fp32-FMA quality (the north_star 1e-3 bound would not
//               survive a plain bf16 rounding of the activations: 2^-9 per element);
//   D          = 16 (batch, 8 used) x 16 (weight rows); lane l holds n = l & 15, m = 4 (l >> 4) + r.
// A workgroup owns a range of 16-row tiles; its waves split K (wave w takes the 32-k steps w, w + NW, ...)
// and combine their partial tiles through LDS; wave 0 runs the fused epilogue.
// ================================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 mf_bf16x8;
typedef __attribute__((ext_vector_type(4))) float mf_f32x4;

constexpr int kMfmaRows = 8;  // batch rows held in LDS (rows 8..15 of the MFMA tile alias rows 0..7)

// planes: [3][kMfmaRows][S] bytes, S = 2 K + 16 (the 16-B pad spreads the rows over the LDS banks)
__device__ __forceinline__ void store_split4(char* planes, int S, int m, int k, f32x4 v) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split3_pk(v.x, v.y, h0, m0, l0);
    split3_pk(v.z, v.w, h1, m1, l1);
    const u32x2 h = {h0, h1}, md = {m0, m1}, l = {l0, l1};
    const size_t off = (size_t)m * S + (size_t)k * 2;
    const size_t plane = (size_t)kMfmaRows * S;
    *reinterpret_cast<u32x2*>(planes + off) = h;
    *reinterpret_cast<u32x2*>(planes + plane + off) = md;
    *reinterpret_cast<u32x2*>(planes + 2 * plane + off) = l;
}

// KS = 32-k steps per wave (K / 32 split over the NW waves of the workgroup).  NW = 16 (1024 threads, one
// workgroup per CU) for the per-layer Linears, NW = 4 for the vocabulary-wide lm_head (many tiles per
// workgroup, three workgroups per CU).
//
// Prologue layout: the 8 input rows are dealt to the waves — NW = 16: wave w owns half (w >> 3) of row
// w & 7; NW = 4: wave w owns rows w and w + 4 — so a lane touches at most JT float4 per row, LayerNorm
// needs two wave reductions per row and one partial-sum exchange through LDS, and the three-plane split
// is 8..24 elements per lane.  (The first version gave every thread a column slice of ALL rows: 16 wave
// reductions and 64 elements of split per thread made the prologue 9k of the kernel's 15k cycles.)
// KSL > 1: the workgroup handles one of KSL equal K slices (blockIdx.y; the K argument is the slice width, the
// weight row stride is KSL * K) — for wide, thin matrices (mlp c_proj at 8 sequences: 48 tiles of K = 3072, where a
// single workgroup per tile spent half of its time staging 8 x 3072 activations).  The slices' partial tiles meet in a
// workspace and are combined in FIXED slice order by the last workgroup of the tile to arrive.  A template parameter,
// so that the KSL == 1 kernels are untouched (two more leading scalar arguments cost them 6 % in situ).
// LINE: the weights are fetched as full 128-byte lines — lane = (row >> 3, 16-B piece & 7), two instructions cover 16
// rows x 64 k — and turned into B fragments through a wave-private 2-KiB LDS slot (see lm_head_wpt_kernel): the B
// fragment layout itself puts 16 different rows into the 16 lanes of a group, i.e. half a line per row per
// instruction, which costs 0.5..0.8 us per launch at 124M and 1.2..2.3 us at GPT-2 XL (8 sequences).  A wave then owns
// PAIRS of 32-k steps (wave + NW i); needs K % 64 == 0 and room for the slots.
// GPL: the input rows arrive as planes in global memory, written by the previous kernel's epilogue (GemvArgs.pl_in,
// layout zg_common.h plane_elem): every lane loads the A fragments of its own 32-k steps straight into registers next
// to the weights — no LDS planes, no split and no barrier in front of the MFMAs (that prologue, repeated by every
// workgroup for all 8 rows, was 55-60 % of these kernels).  With the folded LayerNorm only the row statistics are
// summed from x, beside the loads, and reach wave 0 through the exchange barrier of the first tile.
template <int KS, int NW, bool ARGMAX, int KSL = 1, bool LINE = false, bool GPL = false>
__global__ __launch_bounds__(NW * 64) void gemv_mfma_kernel(const bf16_t* __restrict__ W, const float* __restrict__ xin,
                                                            int N, int K, int M, int tiles_per_wg, int prologue,
                                                            int epilogue, const float* __restrict__ ln_g,
                                                            const float* __restrict__ ln_b, const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_mf[];
    const int ldw = K * KSL;
    if constexpr (KSL > 1) {
        W += (size_t)blockIdx.y * K;
        xin += (size_t)blockIdx.y * K;
    }
    ZG_STAMP_DECL();
    ZG_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = K >> 2, nsteps = K >> 5;
    const int S = 2 * K + 16;
    static_assert(!GPL || (!ARGMAX && NW == 16), "global planes: the per-layer Linears only");
    char* planes = smem_mf;                                                       // [3][8][S] (not with GPL)
    float* red = reinterpret_cast<float*>(smem_mf + (GPL ? (size_t)0 : (size_t)3 * kMfmaRows * S));  // LN partial sums, then partial tiles
    const int ntiles = (N + 15) >> 4;
    const int tile_begin = blockIdx.x * tiles_per_wg;
    const int tile_end = min(tile_begin + tiles_per_wg, ntiles);
    const int brow = lane & 15, bq = lane >> 4;  // B fragment: weight row within the tile, k quarter

    // ---- 0. first tile's weight fragments: independent of everything else
    constexpr int KP = (KS + 1) / 2;  // LINE: pairs of steps per wave
    const int npairs = nsteps >> 1;
    const int lrow = lane >> 3, lpc = lane & 7;  // LINE load shape: row within the half tile, 16-B piece of the line
    u32x4 wq[LINE ? 2 * KP : KS];
    auto load_tile = [&](int tile) {
        if constexpr (LINE) {  // wq[2 i] = rows 0..7, wq[2 i + 1] = rows 8..15 of the k range of pair wave + NW i
            const int r0 = min(tile, ntiles - 1) * 16 + lrow;
            const bf16_t* p0 = W + (size_t)min(r0, N - 1) * ldw + lpc * 8;
            const bf16_t* p1 = W + (size_t)min(r0 + 8, N - 1) * ldw + lpc * 8;
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int q = min(wave + NW * i, npairs - 1);  // surplus pairs re-read the last one (weight 0 below)
                wq[2 * i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p0 + q * 64));
                wq[2 * i + 1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p1 + q * 64));
            }
        } else {
            const bf16_t* wp = W + (size_t)min(min(tile, ntiles - 1) * 16 + brow, N - 1) * ldw + bq * 8;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int st = min(wave + NW * i, nsteps - 1);  // surplus steps re-read the last one (weight 0 below)
                wq[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + st * 32));
            }
        }
    };
    load_tile(tile_begin);
    const int T = a.ctrl ? a.ctrl->seq_len : 1;
    pf_count(a.progress);
    ZG_STAMP(1);
    // bias / residual of the FIRST tile are fetched here, next to the weights, instead of one more
    // dependent L2 round trip inside the epilogue
    // GPL: A fragments of this wave's steps, all three planes; tile rows 8..15 have no batch row behind them: zeros
    constexpr int NAF = GPL ? (LINE ? 2 * KP : KS) * 3 : 1;
    u32x4 af[NAF];
    if constexpr (GPL) {
        const bf16_t* pin = a.pl_in + (KSL > 1 ? (size_t)blockIdx.y * nsteps * kPlaneStep : (size_t)0) + (lane & 7) * 32 + bq * 8;
#pragma unroll
        for (int i = 0; i < NAF / 3; ++i) {
            const int st = LINE ? 2 * min(wave + NW * (i >> 1), npairs - 1) + (i & 1) : min(wave + NW * i, nsteps - 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[i * 3 + p] = u32x4{0u, 0u, 0u, 0u};
                if ((lane & 8) == 0) af[i * 3 + p] = *reinterpret_cast<const u32x4*>(pin + (size_t)(st * 3 + p) * 256);
            }
        }
    }
    float pre_bias = 0.0f, pre_res[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pre_g = 1.0f, pre_c2 = 0.0f, pre_c3 = 0.0f;
    if (wave == 0) {
        const int n = min(min(tile_begin, ntiles - 1) * 16 + brow, N - 1);
        if (a.bias) pre_bias = a.bias[n];
        if (a.pl_out && a.pl_g) pre_g = a.pl_g[n];
        if (GPL && prologue == PRO_LAYERNORM) {
            pre_c2 = a.ln_c2[n];
            pre_c3 = a.ln_c3[n];
        }
        if (epilogue == EPI_RESIDUAL)
#pragma unroll
            for (int r = 0; r < 4; ++r) pre_res[r] = a.resid[(size_t)min(bq * 4 + r, M - 1) * a.resid_stride + n];
    }

    // ---- 1. prologue: transformed input rows -> three bf16 planes in LDS (once per workgroup)
    constexpr int RPW = NW >= 8 ? 1 : kMfmaRows / NW;    // rows per wave
    constexpr int PARTS = NW >= 8 ? NW / kMfmaRows : 1;  // waves sharing a row
    constexpr int JT = NW >= 8 ? KS : (KS + 1) / 2;      // float4 per lane per row (covers K <= 32 KS NW)
    const int part = NW >= 8 ? wave >> 3 : 0;
    const int cols = (nq + PARTS - 1) / PARTS;           // float4 columns per wave
    const int c0 = part * cols, c1 = min(nq, c0 + cols);
    int cidx[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t) cidx[t] = c0 + lane + 64 * t;
    auto row_of = [&](int j) { return NW >= 8 ? (wave & 7) : wave + NW * j; };

    constexpr bool kHasLn = NW == 4 || NW * KS * 32 <= 2048;  // fused LayerNorm is dispatched only for K <= 2048
    // LayerNorm folded out of the product (see gemv_lnk_kernel): the planes hold split(g x) — no statistics in front
    // of the MFMAs, no barrier in the prologue — and the epilogue applies r_m (S1 - mu_m c2_n) + c3_n with the row
    // statistics that were summed alongside.
    // (not for the vocabulary-wide form: its per-tile c2 / c3 fetches cost more than the one prologue barrier saves)
    const bool lin_ln = !ARGMAX && kHasLn && prologue == PRO_LAYERNORM && a.ln_c2 != nullptr;
    if constexpr (GPL) {
        if (lin_ln) {  // row statistics only
            f32x4 v[RPW][JT];
#pragma unroll
            for (int t = 0; t < JT; ++t)
#pragma unroll
                for (int j = 0; j < RPW; ++j)
                    v[j][t] = reinterpret_cast<const f32x4*>(xin + (size_t)min(row_of(j), M - 1) * a.x_stride)[min(cidx[t], nq - 1)];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int m = row_of(j);
                float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int t = 0; t < JT; ++t) {
                    if (cidx[t] >= c1 || m >= M) v[j][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    t1 += (v[j][t].x + v[j][t].y) + (v[j][t].z + v[j][t].w);
                    t2 = fmaf(v[j][t].x, v[j][t].x, fmaf(v[j][t].y, v[j][t].y, fmaf(v[j][t].z, v[j][t].z, fmaf(v[j][t].w, v[j][t].w, t2))));
                }
                t1 = wave_allsum(t1);
                t2 = wave_allsum(t2);
                if (lane == 0) {
                    red[(m * PARTS + part) * 2] = t1;
                    red[(m * PARTS + part) * 2 + 1] = t2;
                }
            }
        }
    } else if (lin_ln) {
        f32x4 v[RPW][JT], g4[JT];
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            const int ic = min(cidx[t], nq - 1);
            g4[t] = reinterpret_cast<const f32x4*>(ln_g)[ic];
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                v[j][t] = reinterpret_cast<const f32x4*>(xin + (size_t)min(row_of(j), M - 1) * a.x_stride)[ic];
        }
        float* stat = red;  // [8 rows][PARTS][2]
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int m = row_of(j);
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int t = 0; t < JT; ++t) {
                if (cidx[t] >= c1 || m >= M) v[j][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                t1 += (v[j][t].x + v[j][t].y) + (v[j][t].z + v[j][t].w);
                t2 = fmaf(v[j][t].x, v[j][t].x, fmaf(v[j][t].y, v[j][t].y, fmaf(v[j][t].z, v[j][t].z, fmaf(v[j][t].w, v[j][t].w, t2))));
                if (cidx[t] < c1) store_split4(planes, S, m, cidx[t] * 4, v[j][t] * g4[t]);
            }
            t1 = wave_allsum(t1);
            t2 = wave_allsum(t2);
            if (lane == 0) {
                stat[(m * PARTS + part) * 2] = t1;
                stat[(m * PARTS + part) * 2 + 1] = t2;
            }
        }
    } else if (kHasLn && prologue == PRO_LAYERNORM) {
        f32x4 v[RPW][JT], g4[JT], b4[JT];
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            const int ic = min(cidx[t], nq - 1);
            g4[t] = reinterpret_cast<const f32x4*>(ln_g)[ic];
            b4[t] = reinterpret_cast<const f32x4*>(ln_b)[ic];
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                v[j][t] = reinterpret_cast<const f32x4*>(xin + (size_t)min(row_of(j), M - 1) * a.x_stride)[ic];
        }
        float* stat = red;  // [8 rows][PARTS][2]
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int t = 0; t < JT; ++t) {
                if (cidx[t] >= c1) v[j][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                t1 += (v[j][t].x + v[j][t].y) + (v[j][t].z + v[j][t].w);
                t2 = fmaf(v[j][t].x, v[j][t].x, fmaf(v[j][t].y, v[j][t].y, fmaf(v[j][t].z, v[j][t].z, fmaf(v[j][t].w, v[j][t].w, t2))));
            }
            t1 = wave_allsum(t1);
            t2 = wave_allsum(t2);
            if (lane == 0) {
                stat[(row_of(j) * PARTS + part) * 2] = t1;
                stat[(row_of(j) * PARTS + part) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        ZG_STAMP(2);
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int m = row_of(j);
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int p = 0; p < PARTS; ++p) {
                s1 += stat[(m * PARTS + p) * 2];
                s2 += stat[(m * PARTS + p) * 2 + 1];
            }
            const float mean = s1 * inv_k;
            const float rstd = __builtin_amdgcn_rsqf(s2 * inv_k - mean * mean + a.eps);
#pragma unroll
            for (int t = 0; t < JT; ++t) {
                if (cidx[t] < c1) {
                    f32x4 o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (m < M) {
                        o.x = fmaf((v[j][t].x - mean) * rstd, g4[t].x, b4[t].x);
                        o.y = fmaf((v[j][t].y - mean) * rstd, g4[t].y, b4[t].y);
                        o.z = fmaf((v[j][t].z - mean) * rstd, g4[t].z, b4[t].z);
                        o.w = fmaf((v[j][t].w - mean) * rstd, g4[t].w, b4[t].w);
                    }
                    store_split4(planes, S, m, cidx[t] * 4, o);
                }
            }
        }
    } else if (prologue == PRO_ATTN_MERGE) {
        const int t_hi = a.t_hi > 0 ? a.t_hi : T;
        const int nsplit = (t_hi + kAttnChunk - 1) / kAttnChunk;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int m = row_of(j);
#pragma unroll
            for (int t = 0; t < JT; ++t) {  // one merge at a time: each already has 4 x 6 loads in flight
                if (cidx[t] >= c1) continue;
                const f32x4 o = merge_attn4(a, min(m, M - 1), cidx[t] * 4, nsplit);
                store_split4(planes, S, m, cidx[t] * 4, (m < M) ? o : f32x4{0.0f, 0.0f, 0.0f, 0.0f});
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int m = row_of(j);
            f32x4 o[JT];
#pragma unroll
            for (int t = 0; t < JT; ++t)
                o[t] = reinterpret_cast<const f32x4*>(xin + (size_t)min(m, M - 1) * a.x_stride)[min(cidx[t], nq - 1)];
#pragma unroll
            for (int t = 0; t < JT; ++t)
                if (cidx[t] < c1) store_split4(planes, S, m, cidx[t] * 4, (m < M) ? o[t] : f32x4{0.0f, 0.0f, 0.0f, 0.0f});
        }
    }
    if constexpr (!GPL) __syncthreads();
    ZG_STAMP(3);

    // ---- 2. tiles: the NW waves split K (wave w takes the 32-k steps w, w + NW, ...) and combine their
    // partial tiles through LDS; wave 0 runs the fused epilogue while the others start the next tile
    Best best[ARGMAX ? 4 : 1];
#pragma unroll
    for (int r = 0; r < (ARGMAX ? 4 : 1); ++r) {
        best[r].val = -3.0e38f;
        best[r].idx = 0x7fffffff;
    }
    const int pos = T - 1;
    float ln_mu[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ln_rs[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    auto ln_stats = [&]() {  // rows m = 4 bq + r of this lane
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (bq & 1) * 4 + r;  // lanes 32..63 duplicate rows 0..7
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int p = 0; p < PARTS; ++p) {
                s1 += red[(m * PARTS + p) * 2];
                s2 += red[(m * PARTS + p) * 2 + 1];
            }
            ln_mu[r] = s1 * inv_k;
            ln_rs[r] = __builtin_amdgcn_rsqf(s2 * inv_k - ln_mu[r] * ln_mu[r] + a.eps);
        }
    };
    if (!GPL && lin_ln && wave == 0) ln_stats();
    const size_t plane = (size_t)kMfmaRows * S;
    const char* arow = planes + (size_t)(lane & 7) * S + bq * 16;  // A fragment: batch row (lane & 15) & 7
    // [2 buffers][NW waves][64 lanes][4].  When the planes alone nearly fill the LDS (K = 3072: 148 KiB) and the
    // workgroup owns a single tile, the partial tiles reuse the plane area once every wave has read its fragments.
    const bool alias_partial = a.waves_per_wg < 0;
    float* partial = alias_partial ? reinterpret_cast<float*>(planes) : red + 64;
    // LINE: transposing slot of this wave behind the partial tiles — row rho (0..15) x 8 pieces of 16 B, piece p at
    // p ^ ((rho >> 1) & 7): the line-shaped writes and the fragment-shaped reads are both conflict free
    char* lslot = reinterpret_cast<char*>(red + 64 + 2 * NW * 64 * 4) + wave * 2048;
    const int wr0 = lrow * 128 + ((lpc ^ ((lrow >> 1) & 7)) << 4);
    const int wr1 = (lrow + 8) * 128 + ((lpc ^ (((lrow + 8) >> 1) & 7)) << 4);
    const int rd0 = brow * 128 + ((bq ^ ((brow >> 1) & 7)) << 4);        // step 2 q:     k = 64 q + 8 bq
    const int rd1 = brow * 128 + (((4 + bq) ^ ((brow >> 1) & 7)) << 4);  // step 2 q + 1: k = 64 q + 32 + 8 bq
    int buf = 0;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        mf_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (LINE) {
            mf_f32x4 acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int q = wave + NW * i;
                const int qc = min(q, npairs - 1);
                *reinterpret_cast<u32x4*>(lslot + wr0) = wq[2 * i];
                *reinterpret_cast<u32x4*>(lslot + wr1) = wq[2 * i + 1];
                __builtin_amdgcn_wave_barrier();  // LDS operations of one wave complete in order: no s_barrier needed
                u32x4 v0 = *reinterpret_cast<const u32x4*>(lslot + rd0);
                u32x4 v1 = *reinterpret_cast<const u32x4*>(lslot + rd1);
                __builtin_amdgcn_wave_barrier();
                if (q >= npairs) v0 = v1 = u32x4{0u, 0u, 0u, 0u};
                const mf_bf16x8 b0 = __builtin_bit_cast(mf_bf16x8, v0), b1 = __builtin_bit_cast(mf_bf16x8, v1);
#pragma unroll
                for (int p = 2; p >= 0; --p) {  // smallest plane first
                    mf_bf16x8 a0, a1;
                    if constexpr (GPL) {
                        a0 = __builtin_bit_cast(mf_bf16x8, af[(2 * i) * 3 + p]);
                        a1 = __builtin_bit_cast(mf_bf16x8, af[(2 * i + 1) * 3 + p]);
                    } else {
                        a0 = *reinterpret_cast<const mf_bf16x8*>(arow + p * plane + (2 * qc) * 64);
                        a1 = *reinterpret_cast<const mf_bf16x8*>(arow + p * plane + (2 * qc + 1) * 64);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc1, 0, 0, 0);
                }
            }
            acc += acc1;
        } else {
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int st = wave + NW * i;
                const int stc = min(st, nsteps - 1);
                u32x4 wv = wq[i];
                if (st >= nsteps) wv = u32x4{0u, 0u, 0u, 0u};
                const mf_bf16x8 b = __builtin_bit_cast(mf_bf16x8, wv);
                mf_bf16x8 a_lo, a_mid, a_hi;
                if constexpr (GPL) {
                    a_lo = __builtin_bit_cast(mf_bf16x8, af[i * 3 + 2]);
                    a_mid = __builtin_bit_cast(mf_bf16x8, af[i * 3 + 1]);
                    a_hi = __builtin_bit_cast(mf_bf16x8, af[i * 3]);
                } else {
                    a_lo = *reinterpret_cast<const mf_bf16x8*>(arow + 2 * plane + stc * 64);
                    a_mid = *reinterpret_cast<const mf_bf16x8*>(arow + plane + stc * 64);
                    a_hi = *reinterpret_cast<const mf_bf16x8*>(arow + stc * 64);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mid, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b, acc, 0, 0, 0);
            }
        }
        if (tile == tile_begin) ZG_STAMP(4);
        if (tile + 1 < tile_end) load_tile(tile + 1);  // next tile's weights fly under the epilogue
        if (alias_partial) __syncthreads();  // all A fragments consumed: the plane area becomes the exchange buffer
        if constexpr (GPL) {  // lanes 32..63 (tile rows 8..15) hold zeros: half-size partial tiles
            if (lane < 32) *reinterpret_cast<mf_f32x4*>(partial + ((buf * NW + wave) * 32 + lane) * 4) = acc;
        } else {
            *reinterpret_cast<mf_f32x4*>(partial + ((buf * NW + wave) * 64 + lane) * 4) = acc;
        }
        __syncthreads();
        if (tile == tile_begin) ZG_STAMP(5);
        if (wave == 0) {
            if (GPL && lin_ln && tile == tile_begin) ln_stats();
            // lanes 32..63 hold duplicates of lanes 0..31 (tile rows 8..15 alias the batch rows 0..7): with 16
            // waves each half of the wave sums 8 of the partial tiles, one cross-half exchange adds the two
            constexpr int NSUM = NW == 16 ? 8 : NW;
            const int w0 = NW == 16 ? (lane >> 5) * 8 : 0;
            mf_f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int w = 0; w < NSUM; ++w) {
                if constexpr (GPL) sum += *reinterpret_cast<const mf_f32x4*>(partial + ((buf * NW + w0 + w) * 32 + (lane & 31)) * 4);
                else sum += *reinterpret_cast<const mf_f32x4*>(partial + ((buf * NW + w0 + w) * 64 + lane) * 4);
            }
            if (NW == 16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], 32, 64);
            }
            const int n = tile * 16 + brow;
            bool run_epilogue = true;
            if constexpr (KSL > 1) {
                // Publish this slice's tile with write-through (agent-scope relaxed atomic = sc1) stores, drain them, take
                // a ticket; the last arriver reads all slices back with agent-scope loads and adds them in slice order.
                // No release / acquire fences: a fence pair (buffer_wbl2 + buffer_inv) cost 1.8 us of a 5 us kernel.
                typedef __attribute__((address_space(1))) unsigned gu32;
                gu32* slot = (gu32*)(a.sk_ws + ((size_t)tile * KSL + blockIdx.y) * 128);
                if (lane < 32) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __hip_atomic_store(slot + lane * 4 + r, __float_as_uint(sum[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int ticket = 0;
                if (lane == 0) ticket = __hip_atomic_fetch_add(a.sk_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ticket = __builtin_amdgcn_readfirstlane(ticket);
                run_epilogue = ticket == KSL - 1;
                if (run_epilogue) {
                    const gu32* base = (const gu32*)(a.sk_ws + (size_t)tile * KSL * 128) + (lane & 31) * 4;
                    unsigned bits[KSL][4];
#pragma unroll
                    for (int ks = 0; ks < KSL; ++ks)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            bits[ks][r] = __hip_atomic_load(base + ks * 128 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sum = mf_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int ks = 0; ks < KSL; ++ks)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum[r] += __uint_as_float(bits[ks][r]);
                    if (lane == 0) __hip_atomic_store(a.sk_cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
                }
            }
            if (run_epilogue && lane < 32 && n < N) {  // lanes 0..31 hold batch rows 0..7
                const bool first = KSL == 1 && tile == tile_begin;
                float bias_n = first ? pre_bias : (a.bias ? a.bias[n] : 0.0f);
                if (lin_ln) {  // y = r_m (S1 - mu_m c2_n) + c3_n; c3 already holds the bias
                    const float c2n = (GPL && first) ? pre_c2 : a.ln_c2[n], c3n = (GPL && first) ? pre_c3 : a.ln_c3[n];
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[r] = fmaf(ln_rs[r], fmaf(-ln_mu[r], c2n, sum[r]), c3n);
                    bias_n = 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = bq * 4 + r;
                    if (m < M) {
                        const float res = first ? pre_res[r]
                                                : ((epilogue == EPI_RESIDUAL) ? a.resid[(size_t)m * a.resid_stride + n] : 0.0f);
                        sum[r] = epilogue_row(a, m, n, sum[r], bias_n, res, pos, best[ARGMAX ? r : 0]);
                    }
                }
                if (!ARGMAX && a.pl_out) {  // the next Linear reads these rows as planes (of g * y when a LayerNorm follows)
                    const float gn = a.pl_g ? (tile == tile_begin ? pre_g : a.pl_g[n]) : 1.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = bq * 4 + r;
                        if (m < M) {
                            uint32_t hi, mid, lo;
                            split3_pk(a.pl_g ? sum[r] * gn : sum[r], 0.0f, hi, mid, lo);
                            a.pl_out[plane_elem(0, m, n)] = (bf16_t)hi;
                            a.pl_out[plane_elem(1, m, n)] = (bf16_t)mid;
                            a.pl_out[plane_elem(2, m, n)] = (bf16_t)lo;
                        }
                    }
                }
            }
        }
        if (tile == tile_begin) ZG_STAMP(6);
        buf ^= 1;
    }
    ZG_STAMP(7);

    // ---- 3. argmax partials: rows m = 4 bq + r live in the 16 lanes of DPP row bq of wave 0
    if constexpr (ARGMAX) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Best b = best[r];
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                Best o;
                o.val = __shfl_xor(b.val, off, 64);
                o.idx = __shfl_xor(b.idx, off, 64);
                b = better(b, o);
            }
            const int m = bq * 4 + r;
            if (wave == 0 && brow == 0 && lane < 32 && m < M) {
                a.part_val[(size_t)m * gridDim.x + blockIdx.x] = b.val;
                a.part_idx[(size_t)m * gridDim.x + blockIdx.x] = b.idx;
            }
        }
    }
    ZG_STAMP_FLUSH();
}

// ================================================================================================
// Plane-fed Linear of the lock-step batch, FOUR-WAVE workgroups (the per-layer Linears whose input rows arrive as
// planes, GemvArgs.pl_in): one 16-row tile per workgroup, the four waves split the 64-k pairs of the (slice of) K.
//
// In-kernel timeline of the 16-wave kernel above at 124M x 8 (tools/kernel_stamps.py, ticks of ~0.45 ns): of a
// 8.6-9.9 k tick body, 1.3-1.6 k are spent by wave 0 at the exchange barrier waiting for the last-launched of the 16
// waves, and 2.0-3.3 k in the epilogue that wave 0 runs alone for all 128 outputs of the tile (some 450 VALU
// instructions at 4 cycles each) — with the planes coming from global memory nothing is left that 16 waves would
// share.  Here every lane issues all its loads at entry (weights as full 128-byte lines through a wave-private
// transposing LDS slot, A fragments of its own pairs, its epilogue operands, the LayerNorm statistics — tile sums the
// producer of x wrote, or its share of x itself), the waves meet once, and each wave finishes ONE accumulator register
// of the tile: lane (n = lane & 15, half = lane >> 4 < 2) of wave w owns output (m = 4 half + w, n).  K slices over
// blockIdx.y (KSL > 1) combine per wave, without a workgroup barrier: by tagged data (slices 1.. store (value, tag)
// words, the last slice polls them and adds in slice order) or, without an epoch word, by a ticket on counter [tile][w] whose
// last arriver adds the slices in fixed order.
// KP = 64-k pairs per wave (K <= 256 KP per slice).
// ================================================================================================
// Arguments: everything an address of the up-front loads depends on sits in the first 16 dwords, which the hardware
// preloads into SGPRs at wave launch (-amdgpu-kernarg-preload-count=16); fields of the GemvArgs block behind them cost a
// scalar load from the kernarg segment first, ~2 k ticks on a cold launch (the first version took W alone as a leading
// argument and spent 3.4 k of its 7.4 k ticks before the last load was issued).  14 dwords are preloaded (16 user SGPRs less
// the kernarg pointer): W, pl_in, xg, nk, flags, e0, e1, cp = 14.  nk = N | K << 16 (K = slice width), flags = M |
// prologue << 4 | epilogue << 8 | operand bits; behind a folded LayerNorm xg = the tile statistics [8][K / 16][2] (flags bit
// 15) or x (rows K apart), e0 = c2, e1 = c3;
// otherwise xg = gain of the planes written, e0 = bias, e1 = residual (rows N apart) — a few zero floats stand in for an
// absent one (flags bits 12..14 say which are real; the loads are unconditional and read index 0 then: a load inside a
// branch makes the compiler wait for it, and with it for every load issued before, at the join); cp = the step control
// block (sequence length of the KV append) or, for K slices, the epoch word of the tags — always a readable address.
// Fields of the argument block that the tail of the kernel needs are touched right behind the vector loads (ZG_PIN):
// the compiler issues a scalar load where a field is first used and waits for it on the spot, which put two to three
// scalar round trips into the epilogue and one in front of the barrier.
template <int KP, int KSL>
__global__ __launch_bounds__(256) void gemv_pl4_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ pl_in,
                                                       const float* __restrict__ xg, unsigned nk, unsigned flags,
                                                       const float* __restrict__ e0, const float* __restrict__ e1,
                                                       const void* __restrict__ cp, const GemvArgs a) {
    __shared__ __attribute__((aligned(16))) float s_stat[16];           // [8 rows][sum, sum of squares]
    __shared__ __attribute__((aligned(16))) float s_part[4 * 32 * 4];   // [wave][lane < 32][4]
    __shared__ __attribute__((aligned(16))) char s_slot[4 * KP * 2048];  // transposing slots: one per wave and pair
    const int N = (int)(nk & 0xffffu), K = (int)(nk >> 16);
    const int M = (int)(flags & 15u), prologue = (int)((flags >> 4) & 15u), epilogue = (int)((flags >> 8) & 15u);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int ldw = K * KSL, npairs = K >> 6, nq = K >> 2;
    ZG_STAMP_DECL();
    ZG_STAMP(0);
    if constexpr (KSL > 1) W += (size_t)blockIdx.y * K;
    const int brow = lane & 15, bq = lane >> 4, lrow = lane >> 3, lpc = lane & 7;

    // ---- every load of the kernel, issued up front
    u32x4 wq[2 * KP];
    {
        const int r0 = tile * 16 + lrow;
        const bf16_t* p0 = W + (size_t)min(r0, N - 1) * ldw + lpc * 8;
        const bf16_t* p1 = W + (size_t)min(r0 + 8, N - 1) * ldw + lpc * 8;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const int qc = min(wave + 4 * i, npairs - 1);  // surplus pairs re-read the last one (weight 0 below)
            wq[2 * i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p0 + qc * 64));
            wq[2 * i + 1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p1 + qc * 64));
        }
    }
    // A fragments: tile rows 8..15 have no batch row behind them (their outputs, lanes 32..63 of the accumulators, are
    // never read), so lanes 8..15 of every 16-lane row fetch the SECOND step of the pair while lanes 0..7 fetch the first:
    // one full-wave load per (pair, plane) instead of two half-empty ones — the vector-memory issue slots, shared by the
    // four waves, were what the entry phase of this kernel waited for.  The second step's operand is the same register
    // rotated by 8 lanes inside the rows (DPP row_ror:8).
    u32x4 af[3 * KP];  // [pair][plane]
    {
        const bf16_t* pin = pl_in + (KSL > 1 ? (size_t)blockIdx.y * (K >> 5) * kPlaneStep : (size_t)0) + (lane & 7) * 32 + bq * 8;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const int st = 2 * min(wave + 4 * i, npairs - 1) + ((lane >> 3) & 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i * 3 + p] = *reinterpret_cast<const u32x4*>(pin + (size_t)(st * 3 + p) * 256);
        }
    }
    const unsigned cpw = static_cast<const unsigned*>(cp)[KSL == 1 ? 1 : 0];  // StepCtrl.seq_len, or the epoch
    const bool ln = prologue == PRO_LAYERNORM;
    const int n = tile * 16 + brow, nc = min(n, N - 1);
    const int m_out = (bq & 1) * 4 + wave, mc = min(m_out, M - 1);
    const int has_e0 = (int)((flags >> 12) & 1u), has_e1 = (int)((flags >> 13) & 1u), has_g = (int)((flags >> 14) & 1u);
    // ln: (c2, c3 — which already holds the bias); otherwise (bias, residual); index 0 of the stand-in when absent
    const float e0v = e0[nc * has_e0];
    const float e1v = e1[ln ? nc : (mc * N + nc) * has_e1];
    const float e_g = xg[nc * has_g];  // (behind a LayerNorm xg is x: any in-range element, unused)
    const float e_c2 = e0v, e_res = ln ? 0.0f : e1v;
    float e_bias = ln ? e1v : e0v;
    // LayerNorm statistics of rows wave and wave + 4 (the rows of this wave's outputs): from the producer's tile sums
    // (flags bit 15: xg = st_in [8][K / 16][2]; lanes = tiles) or from x itself
    const bool st_tiles = (flags >> 15) & 1u;
    float2 sv[2][2];  // (no initialiser: a value that is either loaded or a constant makes the compiler wait for the load,
                      // and every load before it, where the two paths join)
    if (ln && st_tiles) {
        const int ntl = K >> 4;  // <= 128
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                sv[j][t] = reinterpret_cast<const float2*>(xg)[(size_t)(wave + 4 * j) * ntl + min(lane + 64 * t, ntl - 1)];
    }
    f32x4 xv[2][KP];
    if (ln && !st_tiles) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < KP; ++t)
                xv[j][t] = reinterpret_cast<const f32x4*>(xg + (size_t)min(wave + 4 * j, M - 1) * K)[min(lane + 64 * t, nq - 1)];
    }
    {   // the argument-block fields of the tail, fetched under the vector loads
        ZG_PIN(a.progress); ZG_PIN(a.pl_out); ZG_PIN(a.y); ZG_PIN(a.y_stride); ZG_PIN(a.epilogue); ZG_PIN(__float_as_uint(a.eps));
        ZG_PIN(a.st_out);
        if (epilogue == EPI_QKV) {
            ZG_PIN(a.q); ZG_PIN(a.k_cache); ZG_PIN(a.v_cache); ZG_PIN(a.N); ZG_PIN(a.head_dim); ZG_PIN(a.n_heads); ZG_PIN(a.ctx); ZG_PIN(a.kv_f16);
        }
        if constexpr (KSL > 1) {
            ZG_PIN(a.sk_tag); ZG_PIN(a.launch_id); ZG_PIN(a.sk_ws); ZG_PIN(a.sk_cnt); ZG_PIN(a.fault); ZG_PIN(a.spin_limit);
        }
    }
    const int T = (int)cpw;
    const unsigned tag = (cpw << 8) | a.launch_id;

    ZG_STAMP(1);
    // ---- MFMAs: weights -> B fragments through the slot (row rho x 8 pieces of 16 B, piece p at p ^ ((rho >> 1) & 7))
    // (one slot per pair: all writes, then all reads, then the MFMAs — the LDS round trips of the pairs overlap)
    char* lslot = s_slot + wave * (KP * 2048);
    const int wr0 = lrow * 128 + ((lpc ^ ((lrow >> 1) & 7)) << 4);
    const int wr1 = (lrow + 8) * 128 + ((lpc ^ (((lrow + 8) >> 1) & 7)) << 4);
    const int rd0 = brow * 128 + ((bq ^ ((brow >> 1) & 7)) << 4);        // step 2 q:     k = 64 q + 8 bq
    const int rd1 = brow * 128 + (((4 + bq) ^ ((brow >> 1) & 7)) << 4);  // step 2 q + 1: k = 64 q + 32 + 8 bq
    mf_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        *reinterpret_cast<u32x4*>(lslot + i * 2048 + wr0) = wq[2 * i];
        *reinterpret_cast<u32x4*>(lslot + i * 2048 + wr1) = wq[2 * i + 1];
    }
    __builtin_amdgcn_wave_barrier();  // LDS operations of one wave complete in order: no s_barrier needed
    u32x4 bv[2 * KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        bv[2 * i] = *reinterpret_cast<const u32x4*>(lslot + i * 2048 + rd0);
        bv[2 * i + 1] = *reinterpret_cast<const u32x4*>(lslot + i * 2048 + rd1);
        if (wave + 4 * i >= npairs) bv[2 * i] = bv[2 * i + 1] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const mf_bf16x8 b0 = __builtin_bit_cast(mf_bf16x8, bv[2 * i]), b1 = __builtin_bit_cast(mf_bf16x8, bv[2 * i + 1]);
#pragma unroll
        for (int p = 2; p >= 0; --p) {  // smallest plane first
            const u32x4 r = af[i * 3 + p];
            u32x4 r1;
#pragma unroll
            for (int j = 0; j < 4; ++j) r1[j] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r[j], 0x128, 0xF, 0xF, true);  // row_ror:8
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, r), b0, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, r1), b1, acc1, 0, 0, 0);
        }
    }
    ZG_STAMP(2);
    float st_mu = 0.0f, st_rs = 1.0f;  // st_tiles: statistics of row m_out, in registers
    if (ln && st_tiles) {
        const int ntl = K >> 4;
        float t1[2], t2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            t1[j] = (lane < ntl ? sv[j][0].x : 0.0f) + (lane + 64 < ntl ? sv[j][1].x : 0.0f);
            t2[j] = (lane < ntl ? sv[j][0].y : 0.0f) + (lane + 64 < ntl ? sv[j][1].y : 0.0f);
            t1[j] = wave_allsum(t1[j]);
            t2[j] = wave_allsum(t2[j]);
        }
        const float inv_k = 1.0f / (float)K;
        const float s1 = (bq & 1) ? t1[1] : t1[0], s2 = (bq & 1) ? t2[1] : t2[0];
        st_mu = s1 * inv_k;
        st_rs = __builtin_amdgcn_rsqf(s2 * inv_k - st_mu * st_mu + a.eps);
    } else if (ln) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = wave + 4 * j;
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int t = 0; t < KP; ++t) {
                f32x4 v = xv[j][t];
                if (lane + 64 * t >= nq || m >= M) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                t1 += (v.x + v.y) + (v.z + v.w);
                t2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, t2))));
            }
            t1 = wave_allsum(t1);
            t2 = wave_allsum(t2);
            if (lane == 0) {
                s_stat[m * 2] = t1;
                s_stat[m * 2 + 1] = t2;
            }
        }
    }
    acc += acc1;
    ZG_STAMP(3);
    if (lane < 32) *reinterpret_cast<mf_f32x4*>(s_part + (wave * 32 + lane) * 4) = acc;
    pf_count(a.progress);
    __syncthreads();
    ZG_STAMP(4);

    // The epilogue operands were loaded at entry; claim them here, while no store is in flight: vmcnt counts loads and
    // stores in order, so a first use behind a store waits for that store's round trip as well.
    asm volatile("" ::"v"(e0v), "v"(e1v), "v"(e_g));
    // ---- this wave's register of the tile: output (m_out, n) in lanes 0..31
    float y = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) y += s_part[(w * 32 + (lane & 31)) * 4 + wave];
    bool run = true;
    if (KSL > 1 && a.sk_tag != nullptr) {
        // Tagged hand-over: every slice but the LAST stores (value, tag) words and is done; the last slice polls them and adds
        // in slice order — one memory-side round trip behind the slowest slice instead of the three of the ticket below.
        // The poller is the last slice (blockIdx.y = KSL - 1): workgroups are dispatched in block order, so the writers it
        // waits for are placed before it and it can never hold a slot that one of them needs, whatever the occupancy.
        typedef unsigned long long u64;
        u64* slot0 = a.sk_tag + (size_t)tile * KSL * 128 + wave * 32 + (lane & 31);  // slice 0 of this tile
        if ((int)blockIdx.y != KSL - 1) {
            if (lane < 32)
                __hip_atomic_store(slot0 + blockIdx.y * 128, ((u64)tag << 32) | (u64)__float_as_uint(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            run = false;
        } else {
            u64 v[KSL];
            for (int spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int ks = 0; ks < KSL - 1; ++ks) {
                    v[ks] = __hip_atomic_load(slot0 + ks * 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(v[ks] >> 32) == tag;
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                if ((unsigned)spins >= a.spin_limit) {  // bounded: never hang the queue — and never pass silently
                    if (lane == 0 && a.fault) __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float ysum = __uint_as_float((unsigned)v[0]);
#pragma unroll
            for (int ks = 1; ks < KSL - 1; ++ks) ysum += __uint_as_float((unsigned)v[ks]);
            y = ysum + y;  // slice order 0, 1, .., KSL - 1
        }
    } else if constexpr (KSL > 1) {
        // Publish with write-through (agent-scope relaxed atomic = sc1) stores, drain them, take a ticket; the last arriver
        // reads all slices back with agent-scope loads.  No release / acquire fences (see gemv_mfma_kernel).
        typedef __attribute__((address_space(1))) unsigned gu32;
        gu32* slot = (gu32*)(a.sk_ws + ((size_t)tile * KSL + blockIdx.y) * 128) + wave * 32 + (lane & 31);
        if (lane < 32) __hip_atomic_store(slot, __float_as_uint(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(a.sk_cnt + tile * 4 + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        run = ticket == KSL - 1;
        if (run) {
            const gu32* base = (const gu32*)(a.sk_ws + (size_t)tile * KSL * 128) + wave * 32 + (lane & 31);
            unsigned bits[KSL];
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) bits[ks] = __hip_atomic_load(base + ks * 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            y = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) y += __uint_as_float(bits[ks]);
            if (lane == 0) __hip_atomic_store(a.sk_cnt + tile * 4 + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
        }
    }
    ZG_STAMP(5);
    float out = 0.0f;
    const bool valid = run && lane < 32 && n < N && m_out < M;
    if (valid) {
        if (ln) {  // y = r_m (S1 - mu_m c2_n) + c3_n
            float mu = st_mu, rs = st_rs;
            if (!st_tiles) {
                const float inv_k = 1.0f / (float)K;
                mu = s_stat[m_out * 2] * inv_k;
                rs = __builtin_amdgcn_rsqf(s_stat[m_out * 2 + 1] * inv_k - mu * mu + a.eps);
            }
            y = fmaf(rs, fmaf(-mu, e_c2, y), e_bias);
            e_bias = 0.0f;
        }
        Best nobest;
        out = epilogue_row(a, m_out, n, y, e_bias, e_res, T - 1, nobest);
        if (a.pl_out) {  // the next Linear reads this row as planes (of g * y when a LayerNorm follows)
            uint32_t hi, mid, lo;
            split3_pk(has_g ? out * e_g : out, 0.0f, hi, mid, lo);
            a.pl_out[plane_elem(0, m_out, n)] = (bf16_t)hi;
            a.pl_out[plane_elem(1, m_out, n)] = (bf16_t)mid;
            a.pl_out[plane_elem(2, m_out, n)] = (bf16_t)lo;
        }
    }
    if (a.st_out != nullptr && run) {  // tile sums of the rows written, for the LayerNorm of the next Linear (uniform branch)
        const float v = valid ? out : 0.0f;
        const float s1 = row16_allsum(v), s2 = row16_allsum(v * v);
        if (lane < 32 && brow == 0 && m_out < M) *reinterpret_cast<float2*>(a.st_out + ((size_t)m_out * ((N + 15) >> 4) + tile) * 2) = float2{s1, s2};
    }
    ZG_STAMP(6);
    ZG_STAMP(7);
    ZG_STAMP_FLUSH();
}

// ================================================================================================
// Vocabulary projection (ln_f + lm_head + greedy partial argmax) for 2..8 sequences: ONE WAVE PER 16-ROW TILE over the
// whole K, weights fetched as FULL 128-BYTE LINES.
//
// What bounded the K-split kernel above on this matrix (24.6 us per launch = 3.1 TB/s for the 77 MB of 124M's wte) was
// neither its per-tile exchange / barrier nor the MFMAs (the same kernel without them: 24.5 us) but the shape of its
// weight loads: the B fragment of v_mfma_f32_16x16x32_bf16 puts 16 different rows in the 16 lanes of a group, so one
// load instruction touches 16 rows x 64 B — half a line of each, the other half by the next instruction.  The same
// bytes fetched as 8 rows x 128 B per instruction stream at 17.3 us (4.5 TB/s).  So a wave loads line-shaped pieces
// (lane = row >> 3, 16-B piece & 7; two instructions cover 16 rows x 64 k), turns them into B fragments through a
// wave-private 2-KiB LDS slot (one ds_write_b128 and one ds_read_b128 per load, XOR-swizzled, conflict free; LDS
// operations of one wave complete in order, so no barrier), multiplies them with the three activation planes the
// workgroup built once, runs the epilogue on its own accumulators and moves on: no cross-wave exchange and no
// workgroup barrier inside the tile loop.  Same products as the K-split kernel, summed in two fp32 chains (even / odd
// 32-k steps).
template <int NS>  // 32-k steps per tile: K = 32 NS, NS even
__global__ __launch_bounds__(256) void lm_head_wpt_kernel(const bf16_t* __restrict__ W, const float* __restrict__ xin, int N,
                                                          int K, int M, int tiles_per_wg, const float* __restrict__ ln_g,
                                                          const float* __restrict__ ln_b, const GemvArgs a) {
    static_assert(NS % 2 == 0, "pairs of 32-k steps");
    extern __shared__ __attribute__((aligned(16))) char smem_mf[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = 2 * K + 16;
    char* planes = smem_mf;                                                       // [3][8][S]
    Best* s_best = reinterpret_cast<Best*>(smem_mf + (size_t)3 * kMfmaRows * S);  // [4 waves][8 rows]
    char* slot = smem_mf + (size_t)3 * kMfmaRows * S + 4 * kMfmaRows * sizeof(Best) + wave * 4096;  // 2 x 2 KiB per wave
    const int ntiles = (N + 15) >> 4;
    const int tile_begin = blockIdx.x * tiles_per_wg;
    const int tile_end = min(tile_begin + tiles_per_wg, ntiles);
    const int brow = lane & 15, bq = lane >> 4;  // B fragment: weight row within the tile, k quarter
    const int lrow = lane >> 3, lpc = lane & 7;  // load shape: row within the half tile, 16-B piece of the 128-B line

    // wq[2 j] = rows 0..7, wq[2 j + 1] = rows 8..15 of the k range [64 j, 64 j + 64)
    u32x4 wq[NS];
    auto load_tile = [&](int tile) {
        const int r0 = min(tile, ntiles - 1) * 16 + lrow;
        const bf16_t* p0 = W + (size_t)min(r0, N - 1) * K + lpc * 8;
        const bf16_t* p1 = W + (size_t)min(r0 + 8, N - 1) * K + lpc * 8;
#pragma unroll
        for (int j = 0; j < NS / 2; ++j) {
            wq[2 * j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p0 + j * 64));
            wq[2 * j + 1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p1 + j * 64));
        }
    };
    int tile = tile_begin + wave;
    load_tile(tile);  // independent of everything else
    const int T = a.ctrl ? a.ctrl->seq_len : 1;
    pf_count(a.progress);

    // ---- prologue: wave w normalises rows w and w + 4 (LayerNorm.forward, ops.zig:82-104: single pass sum / sum of
    // squares) and writes them as three bf16 planes; no cross-wave statistics
    constexpr int JT = (NS * 8 + 63) / 64;  // float4 per lane per row
    const int nq = K >> 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = wave + 4 * j;
        f32x4 v[JT], g4[JT], b4[JT];
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            const int ic = min(lane + 64 * t, nq - 1);
            g4[t] = reinterpret_cast<const f32x4*>(ln_g)[ic];
            b4[t] = reinterpret_cast<const f32x4*>(ln_b)[ic];
            v[t] = reinterpret_cast<const f32x4*>(xin + (size_t)min(m, M - 1) * a.x_stride)[ic];
        }
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            if (lane + 64 * t >= nq) v[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            t1 += (v[t].x + v[t].y) + (v[t].z + v[t].w);
            t2 = fmaf(v[t].x, v[t].x, fmaf(v[t].y, v[t].y, fmaf(v[t].z, v[t].z, fmaf(v[t].w, v[t].w, t2))));
        }
        t1 = wave_allsum(t1);
        t2 = wave_allsum(t2);
        const float inv_k = 1.0f / (float)K;
        const float mean = t1 * inv_k;
        const float rstd = __builtin_amdgcn_rsqf(t2 * inv_k - mean * mean + a.eps);
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            if (lane + 64 * t < nq) {
                f32x4 o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (m < M) {
                    o.x = fmaf((v[t].x - mean) * rstd, g4[t].x, b4[t].x);
                    o.y = fmaf((v[t].y - mean) * rstd, g4[t].y, b4[t].y);
                    o.z = fmaf((v[t].z - mean) * rstd, g4[t].z, b4[t].z);
                    o.w = fmaf((v[t].w - mean) * rstd, g4[t].w, b4[t].w);
                }
                store_split4(planes, S, m, (lane + 64 * t) * 4, o);
            }
        }
    }
    __syncthreads();

    Best best[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        best[r].val = -3.0e38f;
        best[r].idx = 0x7fffffff;
    }
    const int pos = T - 1;
    const size_t plane = (size_t)kMfmaRows * S;
    const char* arow = planes + (size_t)(lane & 7) * S + bq * 16;  // A fragment: batch row (lane & 15) & 7
    // transposing slot: row rho (0..15) x 8 pieces of 16 B, piece p stored at p ^ ((rho >> 1) & 7)
    const int wr0 = lrow * 128 + ((lpc ^ ((lrow >> 1) & 7)) << 4);              // rows 0..7
    const int wr1 = (lrow + 8) * 128 + ((lpc ^ (((lrow + 8) >> 1) & 7)) << 4);  // rows 8..15
    const int rd0 = brow * 128 + ((bq ^ ((brow >> 1) & 7)) << 4);        // step 2 j:     k = 64 j + 8 bq
    const int rd1 = brow * 128 + (((4 + bq) ^ ((brow >> 1) & 7)) << 4);  // step 2 j + 1: k = 64 j + 32 + 8 bq
    for (; tile < tile_end; tile += 4) {
        mf_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        // the A fragments are the same for every tile: an opaque offset keeps the compiler from hoisting all 3 NS of
        // them out of the tile loop (288 registers at K = 768)
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const char* ar = arow + opaque;
#pragma unroll
        for (int j = 0; j < NS / 2; ++j) {
            char* sl = slot + (j & 1) * 2048;
            *reinterpret_cast<u32x4*>(sl + wr0) = wq[2 * j];
            *reinterpret_cast<u32x4*>(sl + wr1) = wq[2 * j + 1];
            __builtin_amdgcn_wave_barrier();  // LDS operations of one wave complete in order: no s_barrier needed
            const mf_bf16x8 b0 = *reinterpret_cast<const mf_bf16x8*>(sl + rd0);
            const mf_bf16x8 b1 = *reinterpret_cast<const mf_bf16x8*>(sl + rd1);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 2; p >= 0; --p) {  // smallest plane first
                const mf_bf16x8 a0 = *reinterpret_cast<const mf_bf16x8*>(ar + p * plane + (2 * j) * 64);
                const mf_bf16x8 a1 = *reinterpret_cast<const mf_bf16x8*>(ar + p * plane + (2 * j + 1) * 64);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc1, 0, 0, 0);
            }
        }
        const mf_f32x4 acc = acc0 + acc1;
        const int n = tile * 16 + brow;
        if (tile + 4 < tile_end) load_tile(tile + 4);  // the next tile's weights fly under this tile's epilogue
        if (lane < 32 && n < N) {  // lanes 0..31 hold batch rows 0..7 (rows 8..15 of the tile alias them)
            const float bias_n = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bq * 4 + r;
                if (m < M) epilogue_row(a, m, n, acc[r], bias_n, 0.0f, pos, best[r]);
            }
        }
    }

    // ---- greedy partials: rows m = 4 bq + r live in the 16 lanes of DPP row bq; then across the four waves
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Best b = best[r];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            Best o;
            o.val = __shfl_xor(b.val, off, 64);
            o.idx = __shfl_xor(b.idx, off, 64);
            b = better(b, o);
        }
        if (brow == 0 && lane < 32) s_best[wave * kMfmaRows + bq * 4 + r] = b;
    }
    __syncthreads();
    if (tid < M) {
        Best b = s_best[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) b = better(b, s_best[w * kMfmaRows + tid]);
        a.part_val[(size_t)tid * gridDim.x + blockIdx.x] = b.val;
        a.part_idx[(size_t)tid * gridDim.x + blockIdx.x] = b.idx;
    }
}

// Which lm_head launches take the wave-per-tile kernel: the K values whose tile fits a wave's registers.
inline int lm_wpt_steps(const GemvArgs& a) {
    const int off = getenv("ZGPT2_NO_LM_WPT") ? atoi(getenv("ZGPT2_NO_LM_WPT")) : 0;  // read per call: tests flip it between handles
    if (off || a.epilogue != EPI_ARGMAX || a.prologue != PRO_LAYERNORM || a.M < 2 || a.M > kMfmaRows || a.K % 32 != 0) return 0;
    const int ns = a.K / 32;
    return (ns == 12 || ns == 24 || ns == 32) ? ns : 0;
}
inline int lm_wpt_tiles_per_wg() {  // a multiple of the four waves
    static const int v = getenv("ZGPT2_LM_WPT_TILES") ? atoi(getenv("ZGPT2_LM_WPT_TILES")) : 8;
    return v >= 4 ? (v / 4) * 4 : 4;
}

template <int NS>
int launch_lm_wpt(const GemvArgs& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)3 * kMfmaRows * (2 * a.K + 16) + 4 * kMfmaRows * sizeof(Best) + 4 * 4096;
    if (lds > 64 * 1024) {  // K = 1024 (NS = 32): 66,176 B — opt in once per instantiation, like every other launcher here
        static bool raised = false;
        if (!raised) {
            ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_head_wpt_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024));
            raised = true;
        }
    }
    note_kernel("lm_head_wpt_kernel<%d>", NS);
    hipLaunchKernelGGL((lm_head_wpt_kernel<NS>), dim3(grid), dim3(256), lds, s, reinterpret_cast<const bf16_t*>(a.W), a.x, a.N, a.K,
                       a.M, a.rows_per_wave, a.ln_g, a.ln_b, a);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

inline int gemv_mfma_waves(const GemvArgs& a) { return a.epilogue == EPI_ARGMAX ? 4 : 16; }

inline size_t gemv_mfma_lds(int K, int nw, bool alias_partial = false, bool line = false, bool gpl = false) {
    const size_t planes = gpl ? 0 : (size_t)3 * kMfmaRows * (2 * K + 16);
    return planes + 64 * sizeof(float) + (alias_partial ? 0 : (size_t)2 * nw * 64 * 4 * sizeof(float)) + (line ? (size_t)nw * 2048 : 0);
}
// full-line weight loads (LINE instantiations): whole pairs of 32-k steps and room for one 2-KiB slot per wave
inline bool gemv_mfma_line(int K, int nw, bool alias_partial, bool gpl = false) {
    const int off = getenv("ZGPT2_NO_LINE_LOADS") ? atoi(getenv("ZGPT2_NO_LINE_LOADS")) : 0;  // read per call: tests flip it between handles
    return !off && !alias_partial && K % 64 == 0 && gemv_mfma_lds(K, nw, false, true, gpl) <= 160 * 1024;
}

// single-tile workgroups may let the partial tiles alias the planes (see the kernel)
inline bool gemv_mfma_alias(const GemvArgs& a) {
    return a.kslices <= 1 && a.epilogue != EPI_ARGMAX && a.rows_per_wave == 1 && gemv_mfma_lds(a.K, 16) > 160 * 1024;
}

template <int KS, int NW, bool ARGMAX, int KSL, bool LINE, bool GPL = false>
int launch_mfma_inst2(const GemvArgs& a, int grid, bool alias, hipStream_t s) {
    const size_t lds = gemv_mfma_lds(a.K / KSL, NW, alias, LINE, GPL);
    GemvArgs b = a;
    b.waves_per_wg = alias ? -1 : NW;  // < 0: partial tiles alias the planes
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_mfma_kernel<KS, NW, ARGMAX, KSL, LINE, GPL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    note_kernel("gemv_mfma_kernel<%d, %d, %s, %d, %s, %s>", KS, NW, ARGMAX ? "true" : "false", KSL, LINE ? "true" : "false", GPL ? "true" : "false");
    hipLaunchKernelGGL((gemv_mfma_kernel<KS, NW, ARGMAX, KSL, LINE, GPL>), dim3(grid, KSL), dim3(NW * 64), lds, s,
                       reinterpret_cast<const bf16_t*>(a.W), a.x, a.N, a.K / KSL, a.M, a.rows_per_wave, a.prologue,
                       a.epilogue, a.ln_g, a.ln_b, b);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <int KS, int NW, bool ARGMAX, int KSL = 1>
int launch_mfma_inst(const GemvArgs& a, int grid, hipStream_t s) {
    if constexpr (!ARGMAX && NW == 16) {
        if (a.pl_in) {  // input planes in global memory: no LDS planes, nothing to alias
            if (gemv_mfma_line(a.K / KSL, NW, false, true)) return launch_mfma_inst2<KS, NW, ARGMAX, KSL, true, true>(a, grid, false, s);
            return launch_mfma_inst2<KS, NW, ARGMAX, KSL, false, true>(a, grid, false, s);
        }
    }
    const bool alias = KSL == 1 && gemv_mfma_alias(a);
    if (gemv_mfma_line(a.K / KSL, NW, alias)) return launch_mfma_inst2<KS, NW, ARGMAX, KSL, true>(a, grid, alias, s);
    return launch_mfma_inst2<KS, NW, ARGMAX, KSL, false>(a, grid, alias, s);
}

// Plane-fed Linears as four-wave workgroups (gemv_pl4_kernel): one tile per workgroup, whole 64-k pairs, at most five
// pairs per wave and slice (K <= 1280 per slice: every GPT-2 size but XL, which stays on the 16-wave kernel).
inline int pl4_pairs(const GemvArgs& a) {
    const int off = getenv("ZGPT2_NO_PL4") ? atoi(getenv("ZGPT2_NO_PL4")) : 0;  // read per call: tests flip it between handles
    if (off || a.pl_in == nullptr || a.epilogue == EPI_ARGMAX || a.rows_per_wave != 1) return 0;
    if (a.N > 0xffff || (a.prologue == PRO_LAYERNORM && a.x_stride != a.K) || (a.epilogue == EPI_RESIDUAL && a.resid_stride != a.N)) return 0;
    if (a.st_in != nullptr && a.K / 16 > 128) return 0;
    const int ksl = a.kslices > 1 ? a.kslices : 1;
    if (a.K % (64 * ksl) != 0) return 0;
    const int kp = (a.K / ksl / 64 + 3) / 4;
    return kp <= 5 ? kp : 0;
}

template <int KP>
int launch_pl4(const GemvArgs& a, int grid, hipStream_t s) {
    const bool ln = a.prologue == PRO_LAYERNORM;
    const int ksl = a.kslices == 4 ? 4 : 1;
    note_kernel("gemv_pl4_kernel<%d, %d>", KP, ksl);
    const unsigned nk = (unsigned)a.N | ((unsigned)(a.K / ksl) << 16);
    const float* e0 = ln ? a.ln_c2 : a.bias;
    const float* e1 = ln ? a.ln_c3 : (a.epilogue == EPI_RESIDUAL ? a.resid : nullptr);
    const float* xg = ln ? (a.st_in ? a.st_in : a.x) : (a.pl_out ? a.pl_g : nullptr);
    const unsigned flags = (unsigned)a.M | ((unsigned)a.prologue << 4) | ((unsigned)a.epilogue << 8) | ((e0 ? 1u : 0u) << 12) |
                           ((e1 ? 1u : 0u) << 13) | (((!ln && xg) ? 1u : 0u) << 14) | (((ln && a.st_in) ? 1u : 0u) << 15);
    if (!e0) e0 = a.zero;
    if (!e1) e1 = a.zero;
    if (!xg) xg = a.zero;
    const bf16_t* W = reinterpret_cast<const bf16_t*>(a.W);
    if (ksl == 4) {
        GemvArgs b = a;
        if (b.epoch == nullptr || b.launch_id == 0 || b.launch_id > 255) b.sk_tag = nullptr;  // tickets
        hipLaunchKernelGGL((gemv_pl4_kernel<KP, 4>), dim3(grid, 4), dim3(256), 0, s, W, a.pl_in, xg, nk, flags, e0, e1,
                           static_cast<const void*>(b.sk_tag ? b.epoch : reinterpret_cast<const unsigned*>(a.zero)), b);
    } else {
        hipLaunchKernelGGL((gemv_pl4_kernel<KP, 1>), dim3(grid), dim3(256), 0, s, W, a.pl_in, xg, nk, flags, e0, e1,
                           a.ctrl ? static_cast<const void*>(a.ctrl) : static_cast<const void*>(a.zero), a);
    }
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_gemv_mfma(const GemvArgs& a, int grid, hipStream_t s) {
    switch (pl4_pairs(a)) {
        case 1: return launch_pl4<1>(a, grid, s);
        case 2: return launch_pl4<2>(a, grid, s);
        case 3: return launch_pl4<3>(a, grid, s);
        case 4: return launch_pl4<4>(a, grid, s);
        case 5: return launch_pl4<5>(a, grid, s);
        default: break;
    }
    if (a.kslices == 4) {  // four K slices over four workgroups per tile (gemv_kslices)
        const int ks = (a.K / 4 / 32 + 15) / 16;
        if (ks <= 2) return launch_mfma_inst<2, 16, false, 4>(a, grid, s);
        if (ks <= 4) return launch_mfma_inst<4, 16, false, 4>(a, grid, s);
        return launch_mfma_inst<6, 16, false, 4>(a, grid, s);
    }
    const int steps = a.K / 32;
    switch (lm_wpt_steps(a)) {  // lm_head, one wave per tile
        case 12: return launch_lm_wpt<12>(a, grid, s);
        case 24: return launch_lm_wpt<24>(a, grid, s);
        case 32: return launch_lm_wpt<32>(a, grid, s);
        default: break;
    }
    if (a.epilogue == EPI_ARGMAX) {  // lm_head: 4 waves
        const int ks = (steps + 3) / 4;
        if (ks <= 3) return launch_mfma_inst<3, 4, true>(a, grid, s);
        if (ks <= 6) return launch_mfma_inst<6, 4, true>(a, grid, s);
        if (ks <= 13) return launch_mfma_inst<13, 4, true>(a, grid, s);
        return launch_mfma_inst<24, 4, true>(a, grid, s);
    }
    const int ks = (steps + 15) / 16;
    if (ks <= 2) return launch_mfma_inst<2, 16, false>(a, grid, s);
    if (ks <= 4) return launch_mfma_inst<4, 16, false>(a, grid, s);
    return launch_mfma_inst<6, 16, false>(a, grid, s);
}

template <typename WT, int MT, int LPR, int CPL, bool ARGMAX>
int launch_inst(const GemvArgs& a, int grid, hipStream_t s) {
    const int wpw = (MT == 1 && a.waves_per_wg >= 1 && a.waves_per_wg <= 4) ? a.waves_per_wg : 4;
    const size_t lds = ((size_t)(MT == 1 ? wpw : MT) * a.K + 4 * MT * 2 + 64) * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised = false;  // opt in once per instantiation to >64 KiB dynamic LDS
        if (!raised) {
            ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<WT, MT, LPR, CPL, ARGMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised = true;
        }
    }
    ZG_REQUIRE(lds <= 160 * 1024, ZG_ERR_UNSUPPORTED, "gemv: M=%d x K=%d does not fit LDS", a.M, a.K);
    note_kernel("gemv_kernel<%s, %d, %d, %d, %s>", sizeof(WT) == 2 ? "unsigned short" : "float", MT, LPR, CPL, ARGMAX ? "true" : "false");
    hipLaunchKernelGGL((gemv_kernel<WT, MT, LPR, CPL, ARGMAX>), dim3(grid), dim3(64 * wpw), lds, s, a.W, a.x, a.N, a.K,
                       (unsigned)a.M | ((unsigned)a.prologue << 4) | ((unsigned)a.epilogue << 8) | ((unsigned)wpw << 12), a.rows_per_wave,
                       a.ln_g, a.ln_b, a.ctrl ? reinterpret_cast<const int*>(a.ctrl) : reinterpret_cast<const int*>(a.zero), a);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <typename WT, int MT, int LPR, int CPL>
int launch_inst(const GemvArgs& a, int grid, hipStream_t s) {
    // The greedy-sampler epilogue is its own instantiation (distinct kernel name in profiles, and the
    // other kernels carry no argmax state).
    if (a.epilogue == EPI_ARGMAX) return launch_inst<WT, MT, LPR, CPL, true>(a, grid, s);
    return launch_inst<WT, MT, LPR, CPL, false>(a, grid, s);
}

template <typename WT, int MT>
int launch_mt(const GemvArgs& a, int grid, hipStream_t s) {
    const int nch = a.K / 8;
    if (nch <= 16 * 2) return launch_inst<WT, MT, 16, 2>(a, grid, s);
    if (nch <= 16 * 4) return launch_inst<WT, MT, 16, 4>(a, grid, s);
    if (nch <= 16 * 6) return launch_inst<WT, MT, 16, 6>(a, grid, s);
    if (nch <= 16 * 8) return launch_inst<WT, MT, 16, 8>(a, grid, s);
    if (nch <= 32 * 6) return launch_inst<WT, MT, 32, 6>(a, grid, s);
    if (nch <= 32 * 8) return launch_inst<WT, MT, 32, 8>(a, grid, s);
    if (nch <= 64 * 6) return launch_inst<WT, MT, 64, 6>(a, grid, s);
    if (nch <= 64 * 8) return launch_inst<WT, MT, 64, 8>(a, grid, s);
    if (nch <= 64 * 16) return launch_inst<WT, MT, 64, 16>(a, grid, s);
    zg::set_error("gemv: K=%d too large (max 8192)", a.K);
    return ZG_ERR_UNSUPPORTED;
}

template <typename WT>
int launch_wt(const GemvArgs& a, int grid, hipStream_t s) {
    if (a.K % 8 != 0) {
        ZG_REQUIRE(a.prologue == PRO_NONE && a.epilogue == EPI_STORE, ZG_ERR_UNSUPPORTED,
                   "gemv: K=%d not a multiple of 8 is only supported for plain Linear", a.K);
        hipLaunchKernelGGL((gemv_generic_kernel<WT>), dim3((a.N + 3) / 4), dim3(256), 0, s, a);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    if (a.M <= 1) return launch_mt<WT, 1>(a, grid, s);
    if (a.M <= 2) return launch_mt<WT, 2>(a, grid, s);
    if (a.M <= 4) return launch_mt<WT, 4>(a, grid, s);
    if (a.M <= 8) return launch_mt<WT, 8>(a, grid, s);
    zg::set_error("gemv: M=%d > 8 rows per launch", a.M);
    return ZG_ERR_UNSUPPORTED;
}

}  // namespace

int launch_ln_fold(const void* W, int weight_type, const float* g, const float* b, const float* bias, int N, int K, float* c2,
                   float* c3, hipStream_t s) {
    if (weight_type == WT_BF16)
        hipLaunchKernelGGL((ln_fold_kernel<bf16_t>), dim3((N + 3) / 4), dim3(256), 0, s, W, g, b, bias, N, K, c2, c3);
    else
        hipLaunchKernelGGL((ln_fold_kernel<float>), dim3((N + 3) / 4), dim3(256), 0, s, W, g, b, bias, N, K, c2, c3);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int gemv_lanes_per_row(int K) {
    const int nch = K / 8;
    if (nch <= 16 * 8) return 16;
    if (nch <= 32 * 8) return 32;
    return 64;
}

// Wide, thin, un-normalised Linears of the lock-step batch (mlp c_proj) are cut into four K slices over as many
// workgroups when the caller provided the combine workspace.
int gemv_kslices(const GemvArgs& a) {
    static const int off = getenv("ZGPT2_NO_SPLITK") ? atoi(getenv("ZGPT2_NO_SPLITK")) : 0;
    if (off || a.sk_ws == nullptr || a.sk_cnt == nullptr || a.M < 2 || a.M > kMfmaRows) return 1;
    if (a.prologue != PRO_NONE || a.epilogue == EPI_ARGMAX || a.epilogue == EPI_QKV) return 1;
    if (a.K < 2048 || a.K % 128 != 0 || a.K / 4 > 3072 || (a.N + 15) / 16 > a.sk_tiles) return 1;
    return 4;
}

// Rows per wave: enough waves to cover the chip (256 CUs x 4 SIMDs x 2) without dropping below
// one double pass (2 * 64/LPR rows) per wave.
// The matrix-core path serves the model tier's lock-step batch: bf16 weights, 2..8 rows, K a
// multiple of 32 whose three input planes fit in LDS, a fused LayerNorm no wider than 2048.
bool gemv_use_mfma(const GemvArgs& a, int weight_type) {
    static const int off = getenv("ZGPT2_NO_GEMV_MFMA") ? atoi(getenv("ZGPT2_NO_GEMV_MFMA")) : 0;
    if (off || weight_type != WT_BF16 || a.M < 2 || a.M > kMfmaRows) return false;
    if (gemv_kslices(a) > 1) return true;
    if (a.K % 32 != 0 || a.K / 32 < 4 || a.K / 32 > 96) return false;
    if (a.prologue == PRO_LAYERNORM && a.K > 2048) return false;
    if (gemv_mfma_lds(a.K, gemv_mfma_waves(a)) <= 160 * 1024) return true;
    // wide K: only as single-tile workgroups whose partial tiles alias the planes
    static const int wgs = getenv("ZGPT2_MFMA_WGS") ? atoi(getenv("ZGPT2_MFMA_WGS")) : 768;
    return a.epilogue != EPI_ARGMAX && (a.N + 15) / 16 <= wgs && gemv_mfma_lds(a.K, 16, true) <= 160 * 1024;
}

// Can a Linear with this M and K run at all?  (The batched kernels keep all M input rows in LDS.)
namespace {
inline size_t valu_lds(int mt, int K) { return ((size_t)mt * K + 4 * mt * 2 + 64) * sizeof(float); }
inline int valu_mt(int M) { return M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8)); }
// Rows of a plain (no prologue) Linear are independent: when M rows of K floats exceed the LDS the batch is
// run as row groups that fit (the weights are streamed once per group).
inline bool splittable(const GemvArgs& a) {
    return a.prologue == PRO_NONE && (a.epilogue == EPI_STORE || a.epilogue == EPI_RESIDUAL || a.epilogue == EPI_GELU);
}
inline int row_group(const GemvArgs& a) {
    int g = valu_mt(a.M);
    while (g > 1 && valu_lds(g, a.K) > 160 * 1024) g >>= 1;
    return g;
}
}  // namespace

bool gemv_supported(const GemvArgs& a, int weight_type) {
    if (a.M <= 1 || gemv_use_mfma(a, weight_type)) return true;
    if (valu_lds(valu_mt(a.M), a.K) <= 160 * 1024) return true;
    return splittable(a) && valu_lds(row_group(a), a.K) <= 160 * 1024;
}

int gemv_plan(GemvArgs& a, int weight_type) {
    if (gemv_use_mfma(a, weight_type)) {
        const int ntiles = (a.N + 15) / 16;
        a.kslices = gemv_kslices(a);
        if (a.kslices > 1) {  // one tile per workgroup and slice
            a.rows_per_wave = 1;
            return ntiles;
        }
        if (lm_wpt_steps(a) > 0) {  // one wave per tile: every wave of a workgroup gets the same number of tiles
            a.rows_per_wave = lm_wpt_tiles_per_wg();
            return (ntiles + a.rows_per_wave - 1) / a.rows_per_wave;
        }
        static const int wgs = getenv("ZGPT2_MFMA_WGS") ? atoi(getenv("ZGPT2_MFMA_WGS")) : 768;
        int tpw = (ntiles + wgs - 1) / wgs;  // at most ~4 workgroups per CU for the widest matrices
        if (tpw < 1) tpw = 1;
        a.rows_per_wave = tpw;            // tiles per workgroup on this path
        return (ntiles + tpw - 1) / tpw;
    }
    const int rpp2 = 2 * (64 / gemv_lanes_per_row(a.K));
    const int target_waves = 256 * 4 * 2;
    int rpw = (a.N + target_waves - 1) / target_waves;
    rpw = ((rpw + rpp2 - 1) / rpp2) * rpp2;
    if (rpw < rpp2) rpw = rpp2;
    if (const char* e = getenv("ZGPT2_RPW")) {  // tuning experiments only
        const int v = atoi(e);
        if (v > 0 && a.N > 20000) rpw = v;
    }
    a.rows_per_wave = rpw;
    const int waves = (a.N + rpw - 1) / rpw;
    // M == 1 without the argmax tail: one-wave workgroups while the matrix has at most ~8 waves per CU
    static const int wpw_env = getenv("ZGPT2_WPW") ? atoi(getenv("ZGPT2_WPW")) : 0;
    int wpw = 4;
    if (a.M == 1 && a.epilogue != EPI_ARGMAX) wpw = wpw_env > 0 ? wpw_env : (waves <= 2048 ? 1 : 4);
    static const int share_k = getenv("ZGPT2_SHARE_K") ? atoi(getenv("ZGPT2_SHARE_K")) : 2048;
    static const int share_wpw = getenv("ZGPT2_SHARE_WPW") ? atoi(getenv("ZGPT2_SHARE_WPW")) : 0;
    if (a.M == 1 && a.epilogue != EPI_ARGMAX && a.prologue == PRO_NONE && a.K >= share_k && a.K <= 8192)
        wpw = share_wpw > 0 ? share_wpw : 2;  // measured in situ: 2 >= 4 at K = 3072 (124M) and K = 6400 (XL)
    static const int share_merge = getenv("ZGPT2_SHARE_MERGE") ? atoi(getenv("ZGPT2_SHARE_MERGE")) : 4;
    if (a.M == 1 && a.epilogue != EPI_ARGMAX && a.prologue == PRO_ATTN_MERGE && share_merge > 1) wpw = share_merge;
    a.waves_per_wg = wpw;
    return (waves + wpw - 1) / wpw;
}

// Mirrors the dispatch of launch_gemv below for a planned launch (prefetch.hip follows the same tiles).
int gemv_rows_per_wg(const GemvArgs& a, int weight_type) {
    if (gemv_use_mfma(a, weight_type)) return a.kslices > 1 ? 0 : 16 * a.rows_per_wave;
    const int nchq = a.K / 32;
    if (gemv_use_ksplit(a)) return 2 * (64 / (nchq <= 32 ? 16 : (nchq <= 224 ? 32 : 64)));  // launch_ksplit
    if (gemv_use_lnk(a)) return 4 * (64 / (nchq <= 32 ? 16 : (nchq <= 96 ? 32 : 64)));      // launch_lnk
    if (a.M > 1) return 0;
    return a.waves_per_wg * a.rows_per_wave;
}

bool gemv_planes_ok(const GemvArgs& a, int weight_type) {
    if (a.epilogue == EPI_ARGMAX || !gemv_use_mfma(a, weight_type)) return false;
    return a.prologue == PRO_NONE || (a.prologue == PRO_LAYERNORM && a.ln_c2 != nullptr && a.ln_c3 != nullptr && a.K <= 2048);
}

bool gemv_pl4_ok(const GemvArgs& a, int weight_type) {
    if (!gemv_planes_ok(a, weight_type)) return false;
    GemvArgs b = a;
    (void)gemv_plan(b, weight_type);
    if (b.pl_in == nullptr) b.pl_in = reinterpret_cast<const bf16_t*>(a.W);  // any non-null: only the shape is judged
    return pl4_pairs(b) > 0;
}

bool gemv_planes_producer_ok(const GemvArgs& a, int weight_type) { return a.epilogue != EPI_ARGMAX && gemv_use_mfma(a, weight_type); }

// Whether this M == 1 launch runs on one of the two kernels that know x as granules (GemvArgs.xg): the K-split kernel
// (residual / output) or the linearised-LayerNorm kernel (input).
bool gemv_xg_ok(const GemvArgs& a, int weight_type) {
    if (a.M != 1 || gemv_use_mfma(a, weight_type)) return false;
    return gemv_use_ksplit(a) || gemv_use_lnk(a);
}

int launch_gemv(const GemvArgs& a, int weight_type, int grid, hipStream_t s) {
    ZG_REQUIRE((a.xg == nullptr && a.yg == nullptr && a.in_g == nullptr) || gemv_xg_ok(a, weight_type), ZG_ERR_UNSUPPORTED, "gemv: granule input / output asked of a launch outside the K-split kernels");
    ZG_REQUIRE(a.pl_in == nullptr || gemv_planes_ok(a, weight_type), ZG_ERR_UNSUPPORTED, "gemv: input planes given to a launch outside the matrix-core path");
    ZG_REQUIRE(a.pl_out == nullptr || gemv_use_mfma(a, weight_type), ZG_ERR_UNSUPPORTED, "gemv: output planes asked of a launch outside the matrix-core path");
    if (gemv_use_mfma(a, weight_type)) return launch_gemv_mfma(a, grid, s);
    if (gemv_use_ksplit(a)) return weight_type == WT_BF16 ? launch_ksplit<bf16_t>(a, s) : launch_ksplit<float>(a, s);
    if (gemv_use_lnk(a)) return weight_type == WT_BF16 ? launch_lnk<bf16_t>(a, s) : launch_lnk<float>(a, s);
    if (a.M > 1 && valu_lds(valu_mt(a.M), a.K) > 160 * 1024 && splittable(a)) {
        const int g = row_group(a);
        for (int m0 = 0; m0 < a.M; m0 += g) {
            GemvArgs b = a;
            b.M = a.M - m0 < g ? a.M - m0 : g;
            b.x = a.x + (size_t)m0 * a.x_stride;
            b.y = a.y + (size_t)m0 * a.y_stride;
            if (a.resid) b.resid = a.resid + (size_t)m0 * a.resid_stride;
            GemvArgs p = b;
            const int gb = gemv_plan(p, weight_type);  // M == 1 groups are planned differently
            ZG_TRY(weight_type == WT_BF16 ? launch_wt<bf16_t>(p, gb, s) : launch_wt<float>(p, gb, s));
        }
        return ZG_OK;
    }
    return weight_type == WT_BF16 ? launch_wt<bf16_t>(a, grid, s) : launch_wt<float>(a, grid, s);
}

}  // namespace zg
