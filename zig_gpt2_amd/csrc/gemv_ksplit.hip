// gemv_ksplit.hip — M == 1: the K-split kernel (wide plain inputs, the head-merging c_proj) and the linearised-LayerNorm kernel; see gemv.hip
#include "gemv_internal.h"

namespace zg {

namespace {

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
    // 14 preloaded dwords: Wv, xin, N, K, em = epilogue | merge_splits << 8 | has_bias << 16 | has_resid << 17, the attention
    // partials and their split stride, bias and residual (a few zero floats when absent: read at index 0)
    const int epilogue = (int)(em & 0xffu), merge_splits = (int)((em >> 8) & 0xffu);
    const int has_bias = (int)((em >> 16) & 1u), has_resid = (int)((em >> 17) & 1u);
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
    } else {
#pragma unroll
        for (int i = 0; i < CPL; ++i) xr[i] = load_x8(xin + (size_t)wave * Kq + (size_t)min(lr + LPR * i, nchq - 1) * 8);
    }
    float bias_n = 0.0f, resid_n = 0.0f;
    // (absent operands are a few zero words read at index 0: the loads are unconditional — a load inside a uniform branch
    // whose result is merged with a constant makes the compiler wait at the join for every load issued before it)
    if (tid < ROWS) {
        const int n = min(row0 + tid, N - 1);
        bias_n = bias[n * has_bias];
        resid_n = resid[n * has_resid];
    }
    ZG_STAMP(1);
    ZG_PIN(a.y);  // the tail's argument-block fields, fetched under the vector loads (zg_common.h ZG_PIN)
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
        a.y[n] = epilogue == EPI_RESIDUAL ? v + resid_n : (epilogue == EPI_GELU ? gelu_ref(v) : v);
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
    const unsigned em = (unsigned)a.epilogue | ((unsigned)merge_splits << 8) | ((a.bias ? 1u : 0u) << 16) | (has_resid << 17);
    // two passes of 64 / LPR rows per workgroup (four measured slower: 2.65 -> 3.3 us for mlp c_proj)
#define ZG_KS(LPR_, CPL_)                                                                                                \
    {                                                                                                                    \
        constexpr int rows = 2 * (64 / LPR_);                                                                            \
        note_kernel("gemv_ksplit_kernel<%s, %d, %d, 2>", sizeof(WT) == 2 ? "unsigned short" : "float", LPR_, CPL_);         \
        hipLaunchKernelGGL((gemv_ksplit_kernel<WT, LPR_, CPL_, 2>), dim3((a.N + rows - 1) / rows), dim3(256), 0, s, a.W,     \
                           a.x,                                                                                         \
                           a.N, a.K, em, a.part ? a.part : a.zero, a.max_splits, a.bias ? a.bias : a.zero,              \
                           has_resid ? a.resid : a.zero, a);                                                             \
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
template <typename WT, int LPR, int CPL, int NP = 2>  // NP passes of 64 / LPR rows per workgroup
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
        xr[i] = load_x8(xin + off);
        gr[i] = load_x8(ln_g + off);
    }
    float c2n = 0.0f, c3n = 0.0f;
    if (tid < ROWS) {
        const int n = min(row0 + tid, N - 1);
        c2n = c2[n];
        c3n = c3[n];
    }
    const int T = max(cw[1], 1);  // KV append position (EPI_QKV)
    ZG_STAMP(1);
    {   // the argument-block fields of the tail, fetched under the vector loads (zg_common.h ZG_PIN)
        ZG_PIN(a.progress); ZG_PIN(a.y); ZG_PIN(a.y_stride); ZG_PIN(a.epilogue); ZG_PIN(__float_as_uint(a.eps));
        if (epilogue == EPI_QKV) {
            ZG_PIN(a.q); ZG_PIN(a.k_cache); ZG_PIN(a.v_cache); ZG_PIN(a.N); ZG_PIN(a.head_dim); ZG_PIN(a.n_heads); ZG_PIN(a.ctx); ZG_PIN(a.kv_mode); ZG_PIN(a.kv_lo);
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
        note_kernel("gemv_lnk_kernel<%s, %d, %d, 4>", sizeof(WT) == 2 ? "unsigned short" : "float", LPR_, CPL_);       \
        hipLaunchKernelGGL((gemv_lnk_kernel<WT, LPR_, CPL_, 4>), dim3((a.N + rows - 1) / rows), dim3(256), 0, s, a.W, a.x, \
                           (unsigned)a.N | ((unsigned)a.epilogue << 24), a.K, a.ln_g, a.ln_c2, a.ln_c3,                 \
                           a.ctrl ? reinterpret_cast<const int*>(a.ctrl) : reinterpret_cast<const int*>(a.zero), a);   \
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


// Slow generic fallback for K % 8 != 0 (op tier only): one wave per row, scalar loads.

}  // namespace

// M == 1 plain Linear over a wide input: the K-split kernel (measured against the shared-strip form in situ)
bool gemv_use_ksplit(const GemvArgs& a) {
    if (a.M != 1) return false;
    if (a.epilogue != EPI_STORE && a.epilogue != EPI_RESIDUAL && a.epilogue != EPI_GELU) return false;
    if (a.prologue == PRO_ATTN_MERGE)  // head merge folded into the lanes' own chunks: model tier, <= 4 splits known at launch
        return a.head_dim == 64 && a.t_hi > 0 && (a.t_hi + kAttnChunk - 1) / kAttnChunk <= 4 && a.K % 32 == 0 && a.K <= 1024;  // wider rows (XL, K = 1600: three chunks per lane) measured slower than the shared strip
    if (a.prologue != PRO_NONE) return false;
    return a.K >= 2048 && a.K % 32 == 0 && a.K / 32 <= 256;
}

bool gemv_use_lnk(const GemvArgs& a) {
    if (a.M != 1 || a.prologue != PRO_LAYERNORM || a.ln_c2 == nullptr || a.ln_c3 == nullptr) return false;
    if (a.epilogue != EPI_STORE && a.epilogue != EPI_GELU && a.epilogue != EPI_QKV) return false;
    return a.K % 32 == 0 && a.K / 32 <= 128 && a.N <= 16384;
}

int gemv_launch_ksplit(const GemvArgs& a, int weight_type, hipStream_t s) {
    return weight_type == WT_BF16 ? launch_ksplit<bf16_t>(a, s) : launch_ksplit<float>(a, s);
}
int gemv_launch_lnk(const GemvArgs& a, int weight_type, hipStream_t s) {
    return weight_type == WT_BF16 ? launch_lnk<bf16_t>(a, s) : launch_lnk<float>(a, s);
}

int launch_ln_fold(const void* W, int weight_type, const float* g, const float* b, const float* bias, int N, int K, float* c2,
                   float* c3, hipStream_t s) {
    if (weight_type == WT_BF16)
        hipLaunchKernelGGL((ln_fold_kernel<bf16_t>), dim3((N + 3) / 4), dim3(256), 0, s, W, g, b, bias, N, K, c2, c3);
    else
        hipLaunchKernelGGL((ln_fold_kernel<float>), dim3((N + 3) / 4), dim3(256), 0, s, W, g, b, bias, N, K, c2, c3);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
