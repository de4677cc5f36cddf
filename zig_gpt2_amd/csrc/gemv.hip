// gemv.hip — the decode-regime Linear: y[M,N] = x[M,K] * W[N,K]^T (+ bias) for M <= 8.
//
// Replaces the cblas_sgemm call of Linear.forward (reference src/ops.zig:21-46) when the batch
// is tiny (decode: M = number of lock-step sequences).  HBM-bandwidth bound: every weight byte is
// read exactly once, 16 B per lane, K-contiguous rows ([out,in] layout of ops.Linear.weight), so
// one row is read by a group of LPR lanes with fully coalesced 16-B loads and reduced with DPP
// row rotations (no LDS, no barriers in the M == 1 path).
//
// Fused around the dot products (the reference does these as separate host loops / ops):
//   prologue  PRO_LAYERNORM   LayerNorm.forward of the input row   (src/ops.zig:82-104)
//             PRO_ATTN_MERGE  combine split-KV attention partials  (src/ops.zig:284-305 tail)
//   epilogue  EPI_RESIDUAL    state.o + state.x residual adds      (src/main.zig:136-145)
//             EPI_GELU        ops.gelu                             (src/ops.zig:221-228)
//             EPI_QKV         split_qkv + KV-cache append          (src/ops.zig:146-157)
//             EPI_ARGMAX      greedy sampler partial argmax        (replaces src/main.zig:198-207)
#include "zg_kernels.h"

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

// Merged attention output for elements [e0, e0+8) of sequence m (all inside one head):
// out = sum_s exp(m_s - M) o_s / sum_s exp(m_s - M) l_s over the splits that saw keys.
__device__ __forceinline__ W8 merge_attn8(const GemvArgs& a, int m, int e0, int nsplit) {
    const int h = e0 / a.head_dim, d0 = e0 % a.head_dim;
    const float* p = a.part + ((size_t)(m * a.n_heads + h) * a.max_splits) * kPartStride;
    float mx = -1e30f;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, p[s * kPartStride + 64]);
    W8 o = zero_w8();
    float l = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = __expf(p[s * kPartStride + 64] - mx);
        l = fmaf(w, p[s * kPartStride + 65], l);
        const float* os = p + s * kPartStride + d0;  // 8-B aligned (kPartStride is even)
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = fmaf(w, os[j], o.v[j]);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] *= inv;
    return o;
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
    reinterpret_cast<KV*>(cache)[off] = (KV)v;
}

__device__ __forceinline__ void epilogue_row(const GemvArgs& a, int m, int n, float acc, int pos, Best& best) {
    float v = acc + (a.bias ? a.bias[n] : 0.0f);
    switch (a.epilogue) {
        case EPI_STORE:
            a.y[(size_t)m * a.y_stride + n] = v;
            break;
        case EPI_RESIDUAL:
            a.y[(size_t)m * a.y_stride + n] = v + a.resid[(size_t)m * a.resid_stride + n];
            break;
        case EPI_GELU:
            a.y[(size_t)m * a.y_stride + n] = gelu_ref(v);
            break;
        case EPI_QKV: {
            const int E = a.N / 3;
            if (n < E) {
                a.q[(size_t)m * E + n] = v;
            } else {
                const int which = n >= 2 * E;
                const int e = n - (which ? 2 * E : E);
                const int h = e / a.head_dim, d = e % a.head_dim;
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
}

// One workgroup = 4 waves; each wave owns rows [gw * rows_per_wave, +rows_per_wave).
// LPR lanes share one row (64 / LPR rows per pass, two passes in flight), CPL 16-B chunks per lane.
template <typename WT, int MT, int LPR, int CPL, bool ARGMAX>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RPP = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane % LPR, rsub = lane / LPR;
    const int K = a.K, N = a.N, nch = K >> 3;
    const WT* W = reinterpret_cast<const WT*>(a.W);
    const int T = a.ctrl ? a.ctrl->seq_len : 1;
    const int pos = T - 1;

    const int gw = blockIdx.x * 4 + wave;
    const int row_begin = gw * a.rows_per_wave;
    const int row_end = min(row_begin + a.rows_per_wave, N);

    // ---------------------------------------------------------------- prologue
    constexpr bool XREG = (MT == 1) && (CPL <= 8);  // input row cached in registers
    W8 xr[XREG ? CPL : 1];
    if constexpr (XREG) {
        // Register path: every LPR-lane group builds the (transformed) input row in the same
        // chunk layout it will use against the weights.  No LDS, no barrier.
        if (a.prologue == PRO_ATTN_MERGE) {
            const int nsplit = (T + kAttnChunk - 1) / kAttnChunk;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = lr + LPR * i;
                xr[i] = (c < nch) ? merge_attn8(a, 0, c * 8, nsplit) : zero_w8();
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = lr + LPR * i;
                xr[i] = (c < nch) ? load_x8(a.x + c * 8) : zero_w8();
            }
            if (a.prologue == PRO_LAYERNORM) {
                // single pass sum / sum of squares, std = sqrt(E[x^2] - mean^2 + eps): ops.zig:88-95
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < CPL; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s1 += xr[i].v[j];
                        s2 = fmaf(xr[i].v[j], xr[i].v[j], s2);
                    }
                s1 = group_allsum<LPR>(s1);
                s2 = group_allsum<LPR>(s2);
                const float n = (float)K;
                const float mean = s1 / n;
                const float rstd = 1.0f / sqrtf(s2 / n - mean * mean + a.eps);
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const int c = lr + LPR * i;
                    if (c < nch) {
                        const W8 g = load_x8(a.ln_g + c * 8), b = load_x8(a.ln_b + c * 8);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            xr[i].v[j] = fmaf((xr[i].v[j] - mean) * rstd, g.v[j], b.v[j]);
                    }
                }
            }
        }
    } else {
        // LDS path: xs[m][k] for all MT rows; wave w prepares rows w, w+4, ...
        for (int m = wave; m < MT; m += 4) {
            float* xs = smem + (size_t)m * K;
            if (m >= a.M) {
                for (int k = lane; k < K; k += 64) xs[k] = 0.0f;
                continue;
            }
            if (a.prologue == PRO_ATTN_MERGE) {
                const int nsplit = (T + kAttnChunk - 1) / kAttnChunk;
                for (int c = lane; c < nch; c += 64) {
                    const W8 o = merge_attn8(a, m, c * 8, nsplit);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs[c * 8 + j] = o.v[j];
                }
            } else {
                const float* x = a.x + (size_t)m * a.x_stride;
                float s1 = 0.0f, s2 = 0.0f;
                for (int k = lane; k < K; k += 64) {
                    const float v = x[k];
                    xs[k] = v;
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
                if (a.prologue == PRO_LAYERNORM) {
                    s1 = wave_allsum(s1);
                    s2 = wave_allsum(s2);
                    const float n = (float)K;
                    const float mean = s1 / n;
                    const float rstd = 1.0f / sqrtf(s2 / n - mean * mean + a.eps);
                    for (int k = lane; k < K; k += 64)
                        xs[k] = fmaf((xs[k] - mean) * rstd, a.ln_g[k], a.ln_b[k]);
                }
            }
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- rows
    Best best[ARGMAX ? MT : 1];
#pragma unroll
    for (int m = 0; m < (ARGMAX ? MT : 1); ++m) {
        best[m].val = -3.0e38f;
        best[m].idx = 0x7fffffff;
    }

    for (int rb = row_begin; rb < row_end; rb += 2 * RPP) {
        const int r0 = rb + rsub, r1 = rb + RPP + rsub;
        const bool v0 = r0 < row_end, v1 = r1 < row_end;
        const WT* w0p = W + (size_t)(v0 ? r0 : row_begin) * K;
        const WT* w1p = W + (size_t)(v1 ? r1 : row_begin) * K;
        Raw<WT> w0[CPL], w1[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lr + LPR * i;
            if (c < nch) {
                w0[i] = load_raw(w0p, c);
                w1[i] = load_raw(w1p, c);
            } else {
                zero_raw(w0[i]);
                zero_raw(w1[i]);
            }
        }
        float acc0[MT], acc1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc0[m] = acc1[m] = 0.0f;
        if constexpr (XREG) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                acc0[0] = dot8(unpack(w0[i]), xr[i], acc0[0]);
                acc1[0] = dot8(unpack(w1[i]), xr[i], acc1[0]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = lr + LPR * i;
                if (c < nch) {
                    const W8 u0 = unpack(w0[i]), u1 = unpack(w1[i]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const W8 x = load_x8(smem + (size_t)m * K + c * 8);
                        acc0[m] = dot8(u0, x, acc0[m]);
                        acc1[m] = dot8(u1, x, acc1[m]);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc0[m] = group_allsum<LPR>(acc0[m]);
            acc1[m] = group_allsum<LPR>(acc1[m]);
        }
        if (lr == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < a.M) {
                    if (v0) epilogue_row(a, m, r0, acc0[m], pos, best[ARGMAX ? m : 0]);
                    if (v1) epilogue_row(a, m, r1, acc1[m], pos, best[ARGMAX ? m : 0]);
                }
            }
        }
    }

    // ---------------------------------------------------------------- argmax partials
    if constexpr (ARGMAX) {
        __shared__ float s_val[4 * 8];
        __shared__ int s_idx[4 * 8];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const Best b = wave_best(best[m]);
            if (lane == 0) {
                s_val[wave * 8 + m] = b.val;
                s_idx[wave * 8 + m] = b.idx;
            }
        }
        __syncthreads();
        if (threadIdx.x < MT && (int)threadIdx.x < a.M) {
            const int m = threadIdx.x;
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

template <typename WT, int MT, int LPR, int CPL, bool ARGMAX>
int launch_inst(const GemvArgs& a, int grid, hipStream_t s) {
    const size_t lds = (MT == 1 && CPL <= 8) ? 0 : (size_t)MT * a.K * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised = false;  // opt in once per instantiation to >64 KiB dynamic LDS
        if (!raised) {
            ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<WT, MT, LPR, CPL, ARGMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised = true;
        }
    }
    ZG_REQUIRE(lds <= 160 * 1024, ZG_ERR_UNSUPPORTED, "gemv: M=%d x K=%d does not fit LDS", a.M, a.K);
    hipLaunchKernelGGL((gemv_kernel<WT, MT, LPR, CPL, ARGMAX>), dim3(grid), dim3(256), lds, s, a);
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

int gemv_lanes_per_row(int K) {
    const int nch = K / 8;
    if (nch <= 16 * 8) return 16;
    if (nch <= 32 * 8) return 32;
    return 64;
}

// Rows per wave: enough waves to cover the chip (256 CUs x 4 SIMDs x 2) without dropping below
// one double pass (2 * 64/LPR rows) per wave.
int gemv_plan(GemvArgs& a) {
    const int rpp2 = 2 * (64 / gemv_lanes_per_row(a.K));
    const int target_waves = 256 * 4 * 2;
    int rpw = (a.N + target_waves - 1) / target_waves;
    rpw = ((rpw + rpp2 - 1) / rpp2) * rpp2;
    if (rpw < rpp2) rpw = rpp2;
    a.rows_per_wave = rpw;
    const int waves = (a.N + rpw - 1) / rpw;
    return (waves + 3) / 4;
}

int launch_gemv(const GemvArgs& a, int weight_type, int grid, hipStream_t s) {
    return weight_type == WT_BF16 ? launch_wt<bf16_t>(a, grid, s) : launch_wt<float>(a, grid, s);
}

}  // namespace zg
