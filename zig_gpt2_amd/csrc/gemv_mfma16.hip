// gemv_mfma16.hip — the lock-step batch (2..8 sequences) on the matrix cores, 16-wave form: XL shapes, the op tier, K slices by tickets; see gemv.hip
#include "gemv_internal.h"

namespace zg {

namespace {

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


}  // namespace

int gemv_launch_mfma16(const GemvArgs& a, int grid, hipStream_t s) {
    if (a.kslices == 4) {  // four K slices over four workgroups per tile (gemv_kslices)
        const int ks = (a.K / 4 / 32 + 15) / 16;
        if (ks <= 2) return launch_mfma_inst<2, 16, false, 4>(a, grid, s);
        if (ks <= 4) return launch_mfma_inst<4, 16, false, 4>(a, grid, s);
        return launch_mfma_inst<6, 16, false, 4>(a, grid, s);
    }
    const int steps = a.K / 32;
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

}  // namespace zg
