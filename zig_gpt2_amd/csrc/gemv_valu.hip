// gemv_valu.hip — the general GEMV kernel (M <= 8 on the vector ALUs, every prologue / epilogue) and its launch ladder; see gemv.hip
#include "gemv_internal.h"

namespace zg {

namespace {

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

int gemv_launch_valu(const GemvArgs& a, int weight_type, int grid, hipStream_t s) {
    return weight_type == WT_BF16 ? launch_wt<bf16_t>(a, grid, s) : launch_wt<float>(a, grid, s);
}

}  // namespace zg
