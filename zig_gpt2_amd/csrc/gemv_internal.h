// gemv_internal.h — what the GEMV translation units share (gemv.hip: planning and dispatch; gemv_valu.hip: the general VALU
// kernel; gemv_ksplit.hip: the two M = 1 K-split kernels; gemv_mfma16.hip: the 16-wave matrix-core kernel; gemv_pl4.hip: the
// four-wave plane-fed kernel and the wave-per-tile lm_head).  Device helpers live in an anonymous namespace: every unit gets
// its own copy.
#pragma once
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
            if (a.y) a.y[(size_t)m * a.y_stride + n] = v;  // null: the output leaves as planes only (GemvArgs.pl_out)
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
                if (a.kv_mode == 1) kv_store<_Float16>(cache, off, v);
                else if (a.kv_mode == 2) b24_store(cache, a.kv_lo, off, v);
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

// ---- host-side shape tests shared by the launchers and the planner
inline int lm_wpt_steps(const GemvArgs& a) {
    const int off = decode_paths_off() & 32;  // read per call: tests flip it between handles
    if (off || a.epilogue != EPI_ARGMAX || a.prologue != PRO_LAYERNORM || a.M < 2 || a.M > kMfmaRows || a.K % 32 != 0) return 0;
    const int ns = a.K / 32;
    return (ns == 12 || ns == 24 || ns == 32) ? ns : 0;
}
inline int lm_wpt_tiles_per_wg() {  // a multiple of the four waves
    constexpr int v = 8;  // (sweep: profiles/round4_lm_head_tiles_sweep.txt)
    return v >= 4 ? (v / 4) * 4 : 4;
}

inline int gemv_mfma_waves(const GemvArgs& a) { return a.epilogue == EPI_ARGMAX ? 4 : 16; }

inline size_t gemv_mfma_lds(int K, int nw, bool alias_partial = false, bool line = false, bool gpl = false) {
    const size_t planes = gpl ? 0 : (size_t)3 * kMfmaRows * (2 * K + 16);
    return planes + 64 * sizeof(float) + (alias_partial ? 0 : (size_t)2 * nw * 64 * 4 * sizeof(float)) + (line ? (size_t)nw * 2048 : 0);
}
// full-line weight loads (LINE instantiations): whole pairs of 32-k steps and room for one 2-KiB slot per wave
inline bool gemv_mfma_line(int K, int nw, bool alias_partial, bool gpl = false) {
    const int off = decode_paths_off() & 16;  // read per call: tests flip it between handles
    return !off && !alias_partial && K % 64 == 0 && gemv_mfma_lds(K, nw, false, true, gpl) <= 160 * 1024;
}

// single-tile workgroups may let the partial tiles alias the planes (see the kernel)
inline bool gemv_mfma_alias(const GemvArgs& a) {
    return a.kslices <= 1 && a.epilogue != EPI_ARGMAX && a.rows_per_wave == 1 && gemv_mfma_lds(a.K, 16) > 160 * 1024;
}

// Plane-fed Linears as four-wave workgroups (gemv_pl4_kernel): one tile per workgroup, whole 64-k pairs, at most five
// pairs per wave and slice (K <= 1280 per slice: every GPT-2 size but XL, which stays on the 16-wave kernel).
inline int pl4_pairs(const GemvArgs& a) {
    const int off = decode_paths_off() & 2;  // read per call: tests flip it between handles
    if (off || a.pl_in == nullptr || a.epilogue == EPI_ARGMAX || a.rows_per_wave != 1) return 0;
    if (a.N > 0xffff || (a.prologue == PRO_LAYERNORM && a.x_stride != a.K) || (a.epilogue == EPI_RESIDUAL && a.resid_stride != a.N)) return 0;
    if (a.st_in != nullptr && a.K / 16 > 128) return 0;
    const int ksl = a.kslices > 1 ? a.kslices : 1;
    if (a.K % (64 * ksl) != 0) return 0;
    const int kp = (a.K / ksl / 64 + 3) / 4;
    return kp <= 5 ? kp : 0;
}

}  // namespace

// ---- launchers of the units (gemv.hip decides which one a launch takes)
bool gemv_use_ksplit(const GemvArgs& a);
bool gemv_use_lnk(const GemvArgs& a);
int gemv_launch_ksplit(const GemvArgs& a, int weight_type, hipStream_t s);
int gemv_launch_lnk(const GemvArgs& a, int weight_type, hipStream_t s);
int gemv_launch_valu(const GemvArgs& a, int weight_type, int grid, hipStream_t s);
int gemv_launch_mfma16(const GemvArgs& a, int grid, hipStream_t s);            // 16-wave kernel (4 waves for lm_head), K slices
int gemv_launch_pl4(const GemvArgs& a, int pairs, int grid, hipStream_t s);    // pairs = pl4_pairs(a) in 1..5
int gemv_launch_lm_wpt(const GemvArgs& a, int steps, int grid, hipStream_t s);  // steps = lm_wpt_steps(a) in {12, 24, 32}

}  // namespace zg
