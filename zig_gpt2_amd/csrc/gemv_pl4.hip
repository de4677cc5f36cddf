// gemv_pl4.hip — the lock-step batch on the matrix cores, four-wave plane-fed form (the per-layer Linears), and the wave-per-tile lm_head; see gemv.hip
#include "gemv_internal.h"

namespace zg {

namespace {

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
            ZG_PIN(a.q); ZG_PIN(a.k_cache); ZG_PIN(a.v_cache); ZG_PIN(a.N); ZG_PIN(a.head_dim); ZG_PIN(a.n_heads); ZG_PIN(a.ctx); ZG_PIN(a.kv_mode); ZG_PIN(a.kv_lo);
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
        // the next tile of this wave streams in behind the current one: each pair of weight registers is reloaded as soon as
        // its contents sit in the LDS slot, so a wave always has a whole tile (24 KB at K = 768) of loads in flight
        const bool has_next = tile + 4 < tile_end;  // (wave-uniform)
        const int nr0 = min(tile + 4, ntiles - 1) * 16 + lrow;
        const bf16_t* np0 = W + (size_t)min(nr0, N - 1) * K + lpc * 8;
        const bf16_t* np1 = W + (size_t)min(nr0 + 8, N - 1) * K + lpc * 8;
#pragma unroll
        for (int j = 0; j < NS / 2; ++j) {
            char* sl = slot + (j & 1) * 2048;
            *reinterpret_cast<u32x4*>(sl + wr0) = wq[2 * j];
            *reinterpret_cast<u32x4*>(sl + wr1) = wq[2 * j + 1];
            if (has_next) {
                wq[2 * j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(np0 + j * 64));
                wq[2 * j + 1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(np1 + j * 64));
            }
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


}  // namespace

int gemv_launch_pl4(const GemvArgs& a, int pairs, int grid, hipStream_t s) {
    switch (pairs) {
        case 1: return launch_pl4<1>(a, grid, s);
        case 2: return launch_pl4<2>(a, grid, s);
        case 3: return launch_pl4<3>(a, grid, s);
        case 4: return launch_pl4<4>(a, grid, s);
        case 5: return launch_pl4<5>(a, grid, s);
        default: break;
    }
    ZG_REQUIRE(false, ZG_ERR_ARG, "gemv_pl4: %d pairs per wave", pairs);
}
int gemv_launch_lm_wpt(const GemvArgs& a, int steps, int grid, hipStream_t s) {
    switch (steps) {
        case 12: return launch_lm_wpt<12>(a, grid, s);
        case 24: return launch_lm_wpt<24>(a, grid, s);
        case 32: return launch_lm_wpt<32>(a, grid, s);
        default: break;
    }
    ZG_REQUIRE(false, ZG_ERR_ARG, "lm_head_wpt: %d steps per tile", steps);
}

}  // namespace zg
