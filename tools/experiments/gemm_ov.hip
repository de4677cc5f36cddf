// gemm_ov.hip — the large-batch Linear with a bf16 result on the CDNA4 matrix cores, third generation:
//   C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]) rounded to bf16, bf16 operands, fp32 accumulate, act = GELU or identity.
// Linear.forward of the reference (src/ops.zig:21-46: cblas_sgemm RowMajor/NoTrans/Trans over a bias-prefilled output)
// followed by gelu (src/ops.zig:221-228) for M >> 1 — BASELINE's 768 x 3072 point is c_fc + GELU at M = 8192.
//
// What this generation adds to gemm_s4.hip (same 256 x 192 tile, four waves, one per SIMD, v_mfma_f32_32x32x16_bf16, same
// LDS ring of two K-steps filled by buffer_load ... lds): THE EPILOGUE OF A TILE RUNS UNDER THE NEXT TILE'S MAIN LOOP.
// In gemm_s4 the matrix pipe idles for 10.6 k of every 32.5 k cycles a tile takes (profiles/round4_gemm_a.txt) while the
// wave reads its 192 accumulators, applies bias / GELU, rounds, stages rows through LDS and stores.  Here
//   * the accumulators of the tile being multiplied live in a[0:191]; at the end of a tile they are copied to v[64:255]
//     (192 v_accvgpr_read) and the accumulators restart at the NEXT tile's bias row — 12 MFMAs of a bias fragment (the fp32
//     bias as three bf16 terms hi + mid + lo in three k positions) times a fragment of ones: exact, and it is also the zeroing;
//   * the drain of v[64:255] — GELU, bf16 rounding, a wave-private LDS image of each 32-row m-tile, 16-byte row stores — is a
//     PROGRAM of ~120 single-instruction ops per 32 x 32 accumulator block, dealt 2-4 per MFMA shadow over one K-step of the
//     next tile: twelve blocks = the first twelve K-steps.  The K-step body is the same code for every block (its 16 values
//     are moved into compiler registers by a block-specific prelude of 16 moves), so the kernel has six bodies — (none,
//     block, block + m-tile store tail) x what the K-step before carried — and fits the instruction cache; the same
//     programs run back to back after the last tile of a workgroup (the only exposed drain);
//   * with both register files spoken for — accumulators and ALL fragments (a[192:255]: B of a whole K-step in ONE rolling
//     buffer, A double-buffered) in the accumulation file, the drain copy in v[64:255] — the compiler is held to v[0:63]
//     (amdgpu_num_vgpr) for addresses and the drain's temporaries; every register above is named by hand in inline asm.
//   * barriers sit one step BEHIND the last fragment read of the region they release, so no LDS latency is exposed in front
//     of them, and fragments stay valid across a tile boundary (nothing is re-read).
// Audit after every edit (tools/README.md): no compiler-generated v_accvgpr_* and no scratch in the -S output.
#include <stdlib.h>

#include <type_traits>

#include "gemm_ov.h"

namespace zg {

// diagnostic (dbg bit 256), layout as g_s4_stamps: [0] grid; [1 + 2 b ..] shader-clock {start, end} of workgroup b's wave 0;
// [513 + 2 b ..] the same on the 100 MHz clock; [1025 ..] workgroup 0's phase stamps
__device__ unsigned long long g_ov_stamps[1 + 4 * 256 + 16];

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define ZG_SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ lds_ptr_t to_lds(unsigned byte_addr) { return (lds_ptr_t)(size_t)byte_addr; }

template <int I>
using Ic = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(Ic<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------------------------------- geometry
struct OV {
    static constexpr int BN = 192;  // (256-row tiles)
    static constexpr int A_SLOT = 256 * 128, B_SLOT = BN * 128;   // one K-step (64 of K) of the A / B tile
    static constexpr int A_OFF = 0, B_OFF = 2 * A_SLOT;           // [A slot 0][A slot 1][B slot 0][B slot 1]
    static constexpr int BIAS_OFF = 2 * A_SLOT + 2 * B_SLOT;      // two tiles' bias rows (fp32), by tile parity
    // bf16 staging, per wave: one 32-row m-tile x the wave's 96 columns, rows padded by 16 B
    static constexpr int ST_ROW = 192 + 16, ST_WAVE = 32 * ST_ROW, ST_OFF = BIAS_OFF + 2 * 1024;
    static constexpr int LDS = ST_OFF + 4 * ST_WAVE;              // 140 KiB
    // hand-owned registers
    static constexpr int FB0 = 192, FA0 = 240, DR0 = 64;
};
constexpr int acc_reg(int b) { return 16 * b; }                         // accumulator block b = 3 i + j: a[16 b : 16 b + 15]
constexpr int fb_reg(int ks, int j) { return OV::FB0 + 4 * (3 * ks + j); }  // B fragment (k-slice ks, column tile j)
constexpr int fa_reg(int buf, int i2) { return OV::FA0 + 4 * (2 * buf + i2); }
constexpr int dr_reg(int b, int r) { return OV::DR0 + 16 * b + r; }     // drain copy of accumulator register r of block b

// ---------------------------------------------------------------------------------------------------- the drain program
// One op = (about) one instruction.  Per accumulator block (i, j), 16 values x_r (GELU):
//   A  t_r = x_r^2            B  t_r = t_r k2 + k1      C  t_r = x_r t_r (= -2 log2e u)   D  t_r = 2^t_r
//   E  t_r = t_r + 1          F  t_r = 1 / t_r          G  t_r = x_r t_r (= gelu)          H_g  round 4 -> ds_write_b64 (x4)
// and, in the K-step of the third block of an m-tile (variant kTail): R_k  ds_read_b128 of a 16-byte row piece (x6), seven
// idle slots (the reads return meanwhile), W  wait, S_k  store (x6).  Without GELU a block is H_0..H_3 alone.
enum { kNone = 0, kBlk = 1, kTail = 2 };             // what a K-step carries besides its MFMAs
template <bool GELU>
struct Prog {
    static constexpr int BLK = GELU ? 7 * 16 + 4 : 4;   // ops per block
    static constexpr int IDLE = 7;
    static constexpr int len(int V) { return V == kNone ? 0 : V == kBlk ? BLK : BLK + 6 + IDLE + 1 + 6; }
    enum { oA, oB, oC, oD, oE, oF, oG, oH, oR, oW, oS, oNop };
    static constexpr int kind(int V, int q) {
        if (q < 0 || q >= len(V)) return oNop;
        if (q >= BLK) {
            const int tq = q - BLK;
            return tq < 6 ? oR : tq < 6 + IDLE ? oNop : tq == 6 + IDLE ? oW : oS;
        }
        if (!GELU) return oH;
        return q < 112 ? oA + q / 16 : oH;
    }
    static constexpr int sub(int V, int q) {             // r (A..G), g (H), k (R, S)
        if (q >= BLK) {
            const int tq = q - BLK;
            return tq < 6 ? tq : tq - (6 + IDLE + 1);
        }
        if (!GELU) return q;
        return q < 112 ? q % 16 : q - 112;
    }
    static constexpr bool is_lds(int V, int q) { const int k = kind(V, q); return k == oH || k == oR; }
    static constexpr bool is_store(int V, int q) { return kind(V, q) == oS; }
};

// ---------------------------------------------------------------------------------------------------- the K-step schedule
// Step J = 4 h + ks (A half h, k-slice ks) has six MFMAs m = 3 i2 + j; behind MFMA m sits gap (J, m) with one main item:
//   m = 0, 1          fragment reads A (next step, m-tile i2 = m)
//   m = 3, 4, 5       J >= 4: fragment reads B of the NEXT K-step (k-slice J - 4, column tile m - 3), straight over the
//                     fragment the MFMA in front of the gap used last
//   m = 2 .. 5        J = 1..3: ten DMA pieces — A half 1 of K-step t + 1 (4), B of K-step t + 2 (6)
//   m = 2             J >= 4: DMA piece J - 4 of A half 0 of K-step t + 2
// and a number of drain ops that depends on what else the gap carries.
constexpr bool main_is_lds(int J, int m) { return m <= 1 || (J >= 4 && m >= 3); }
constexpr bool main_is_dma(int J, int m) { return (J >= 1 && J <= 3 && m >= 2 && (J - 1) * 4 + (m - 2) < 10) || (J >= 4 && m == 2); }
constexpr int gap_cap(int J, int m) { return main_is_dma(J, m) ? 2 : main_is_lds(J, m) ? 3 : 4; }
constexpr int kstep_cap() {
    int c = 0;
    for (int J = 0; J < 8; ++J)
        for (int m = 0; m < 6; ++m) c += gap_cap(J, m);
    return c;
}
constexpr int gap_q0(int J, int m) {                 // first drain op of gap (J, m)
    int c = 0;
    for (int jj = 0; jj < 8; ++jj)
        for (int mm = 0; mm < 6; ++mm) {
            if (jj == J && mm == m) return c;
            c += gap_cap(jj, mm);
        }
    return c;
}
template <bool GELU>
constexpr int drain_count(int V, int J, int m, bool lds) {  // drain ops of one kind in gap (J, m) of a K-step of variant V
    int n = 0;
    const int q0 = gap_q0(J, m);
    for (int q = q0; q < q0 + gap_cap(J, m); ++q) n += lds ? Prog<GELU>::is_lds(V, q) : Prog<GELU>::is_store(V, q);
    return n;
}
// LDS operations issued behind the main item of gap (J, m0) up to the end of the step (its own drain ops included)
template <bool GELU>
constexpr int lds_after(int V, int J, int m0) {
    int n = drain_count<GELU>(V, J, m0, true);
    for (int m = m0 + 1; m < 6; ++m) n += (main_is_lds(J, m) ? 1 : 0) + drain_count<GELU>(V, J, m, true);
    return n;
}
template <bool GELU>
constexpr int lds_upto(int V, int J, int m1) {       // LDS operations of gaps 0 .. m1 of step J
    int n = 0;
    for (int m = 0; m <= m1; ++m) n += (main_is_lds(J, m) ? 1 : 0) + drain_count<GELU>(V, J, m, true);
    return n;
}
template <bool GELU>
constexpr int stores_in(int V, int J0, int m0, int J1, int m1) {  // drain stores in gaps (J0, m0) .. (J1, m1) of one K-step
    int n = 0;
    for (int J = J0; J <= J1; ++J)
        for (int m = (J == J0 ? m0 : 0); m <= (J == J1 ? m1 : 5); ++m) n += drain_count<GELU>(V, J, m, false);
    return n;
}

__device__ __forceinline__ void tile_of(int idx, int tiles_m, int tiles_n, int gw, int& tm, int& tn) {
    const int band = idx / (tiles_m * gw), full = tiles_n / gw;
    if (band < full) {
        const int r = idx - band * tiles_m * gw;
        tm = r / gw;
        tn = band * gw + r % gw;
    } else {
        const int w = tiles_n - full * gw, r = idx - full * tiles_m * gw;
        tm = r / w;
        tn = full * gw + r % w;
    }
}

struct Ahead {  // where a K-step of the DMA stream comes from
    unsigned baseA, baseB, kbA, kbB;
};

// ---------------------------------------------------------------------------------------------------- hand-issued instructions
// Under amdgpu_num_vgpr(64) the compiler still considers a[0:63] its own (it parks long-lived values there instead of
// recomputing them); every statement that writes accumulators names those 64 as clobbered, so nothing of the compiler's can
// live in them across the main loop.  The registers above v63 / a63 are reserved by the attribute: it never touches them.
#define ZG_A64_CLOBBERS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
template <int DST, int OFF>
__device__ __forceinline__ void ds_read_frag(unsigned addr) {  // 16 bytes per lane -> a[DST : DST + 3]
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%3" ::"v"(addr), "i"(DST), "i"(DST + 3), "i"(OFF) : "memory");
}
template <int ACC, int SRC0, int SRC1>
__device__ __forceinline__ void mfma_hand(void) {  // a[ACC..] += a[SRC0..] (weight rows -> output columns) x a[SRC1..] (activation rows)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], a[%c2:%c3], a[%c4:%c5], a[%c0:%c1]" ::"i"(ACC), "i"(ACC + 15), "i"(SRC0), "i"(SRC0 + 3),
                 "i"(SRC1), "i"(SRC1 + 3)
                 : ZG_A64_CLOBBERS);
}
template <int ACC>
__device__ __forceinline__ void mfma_init(const u32x4& wfrag, const u32x4& ones) {  // a[ACC..] = wfrag x ones + 0
    asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[%c0:%c1], %2, %3, 0" ::"i"(ACC), "i"(ACC + 15), "v"(wfrag), "v"(ones) : ZG_A64_CLOBBERS);  // (VALU-written operands)
}
template <int N>
__device__ __forceinline__ void acc_to_drain(void) {  // v[64 + N] = a[N]
    asm volatile("v_accvgpr_read_b32 v%c0, a%c1" ::"i"(OV::DR0 + N), "i"(N) : ZG_A64_CLOBBERS);
}

template <bool GELU, int ST_AUX, int ABL>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(64))) void gemm_ov_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias, bf16_t* __restrict__ C, int M, int N,
    unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
    // all 14 argument dwords are preloaded into SGPRs (zg_common.h ZG_PIN); packing as in gemm_s4_kernel:
    // p0 = lda | ldb << 16; p1 = ldc | K-steps per plane << 20 | plane pairs << 28; p2 = A planes | B planes << 12 | band width << 24;
    // p3 = workgroups | dbg << 10
    using P = OV;
    using PG = Prog<GELU>;
    // diagnostic ablations, compile time (ZGPT2_OV_ABL; timing only, wrong results): 1 no DMA, 2 no barriers, 4 no fragment reads,
    // 8 no waits on the fragment reads, 16 no arithmetic drain ops, 32 no drain stores, 64 no block prelude, 128 no drain staging (LDS)
    constexpr bool abl_dma = ABL & 1, abl_bar = ABL & 2, abl_rd = ABL & 4, abl_lgkm = ABL & 8, abl_valu = ABL & 16, abl_st = ABL & 32, abl_lb = ABL & 64, abl_lds = ABL & 128;
    GemmPlanes pl;
    pl.lda = (int)(p0 & 0xffffu);
    pl.ldb = (int)(p0 >> 16);
    const int ldc = (int)(p1 & 0xfffffu);
    pl.kpp = (int)((p1 >> 20) & 0xffu);
    pl.npairs = (int)(p1 >> 28);
    pl.pa_bits = p2 & 0xfffu;
    pl.pb_bits = (p2 >> 12) & 0xfffu;
    const int gw = (int)(p2 >> 24);
    const int dbg = (int)(p3 >> 10);
    const int tiles_m = (M + 255) >> 8, tiles_n = (N + P::BN - 1) / P::BN;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)lds;

    // ---- this workgroup's tiles: XCD x = bid % 8 owns a contiguous range of the banded order
    const int n_tiles = tiles_m * tiles_n, G = (int)(p3 & 0x3ffu), bid = blockIdx.x;
    const int nx = G < 8 ? G : 8;
    const int xcd = bid % nx, loc = bid / nx;
    const int gx = G / nx + (xcd < G % nx ? 1 : 0);
    const int q8 = n_tiles / nx, r8 = n_tiles % nx;
    const int t_begin = xcd * q8 + min(xcd, r8), t_end = t_begin + q8 + (xcd < r8 ? 1 : 0);
    int idx = t_begin + loc;
    if (idx >= t_end) return;
    const bool stamp = (dbg & 256) && wave == 0 && bid < 256;
    unsigned long long t_start = 0, w_start = 0;
    int n_phase = 0;
    if (stamp) {
        t_start = __builtin_readcyclecounter();
        w_start = __builtin_amdgcn_s_memrealtime();
    }
    auto phase_stamp = [&]() __attribute__((always_inline)) {
        if (stamp && bid == 0 && n_phase < 16) {
            if (lane == 0) g_ov_stamps[1025 + n_phase] = __builtin_readcyclecounter();
            ++n_phase;
        }
    };
    phase_stamp();
    // (the kernel descriptor's register counts come from what the compiler sees: name the top of both hand-owned files once)
    asm volatile("" ::: "v255", "a255");

    // ---- DMA sources (as gemm_s4_kernel).  A piece = 8 unit rows x 128 B, written lane-linearly (lane -> row lane / 8, 16-B
    // position lane % 8); position p of LDS row R holds source chunk p ^ ((R >> 1) & 7) (the swizzle the fragment reads undo).
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * pl.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)N * pl.ldb * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias != nullptr ? (unsigned)N * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (unsigned)((size_t)M * ldc * 2), 0x00020000);
    int tile_par = 0;  // parity of the current tile: its bias row is in buffer tile_par, the next tile's in the other one
    auto fetch_bias = [&](int n0_, int par) __attribute__((always_inline)) {
        if (wave == 0) {
#pragma unroll
            for (int p = 0; p < P::BN / 64; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, to_lds(lds_base + P::BIAS_OFF + par * 1024 + p * 256), 4, (unsigned)lane * 4u,
                                                         (unsigned)(n0_ + 64 * p) * 4u, 0, 0);
        }
    };
    const unsigned lda2 = (unsigned)pl.lda * 2u, ldb2 = (unsigned)pl.ldb * 2u;
    unsigned relA, relB;
    {
        const int csrc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
        relA = (unsigned)(wave * 8 + (lane >> 3)) * lda2 + (unsigned)csrc * 16u;
        relB = (unsigned)(wave * 8 + (lane >> 3)) * ldb2 + (unsigned)csrc * 16u;
    }

    // ---- fragment reads: lane -> row lane & 31 of a 32-row MFMA tile, k-chunk 2 ks + (lane >> 5) of the row's 8.
    // a_addr points into the slot the CURRENT K-step's A reads come from (flipped in front of step 7, whose reads are the
    // next K-step's), b_addr into the slot of the NEXT K-step's B (all the B reads of a K-step are that).
    unsigned a_addr[4], b_addr[4];
    {
        const int l31 = lane & 31, hh = lane >> 5;
        const unsigned swz = (unsigned)((hh ^ ((l31 >> 1) & 7)) << 4);
        const unsigned a0 = lds_base + P::A_OFF + (unsigned)(wr * 64 + l31) * 128u + swz;
        const unsigned b0 = lds_base + P::B_OFF + (unsigned)(wc * (P::BN / 2) + l31) * 128u + swz;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a_addr[k] = a0 ^ (32u * k);
            b_addr[k] = b0 ^ (32u * k);
        }
    }
    int slot = 0;  // LDS slot of the current K-step

    // ---- K-steps walk the plane pairs: step kt = pair * kpp + kk multiplies A plane pa[pair] with B plane pb[pair]
    const int kpp = pl.kpp;
    int pi_cur = 0, kk_cur = 0;
    int tm, tn;
    tile_of(idx, tiles_m, tiles_n, gw, tm, tn);
    const unsigned strideA = 256u * lda2, strideB = (unsigned)P::BN * ldb2;
    unsigned curA = (unsigned)tm * strideA, curB = (unsigned)tn * strideB;
    int m0 = tm * 256, n0 = tn * P::BN;
    constexpr unsigned kOob = 0x80000000u;  // tile base of "no next tile": every lane out of range -> zero fill
    unsigned nxtA = kOob, nxtB = kOob;
    int nidx = idx + gx, ntm = 0, ntn = 0;
    if (nidx < t_end) {
        tile_of(nidx, tiles_m, tiles_n, gw, ntm, ntn);
        nxtA = (unsigned)ntm * strideA;
        nxtB = (unsigned)ntn * strideB;
    }
    auto ahead = [&](int d) __attribute__((always_inline)) {  // K-step t + d of the stream (d <= 2 <= kpp); runs on into the next tile
        int kk = kk_cur + d, pi = pi_cur;
        if (kk >= kpp) {
            kk -= kpp;
            ++pi;
        }
        const bool in_cur = pi < pl.npairs;
        if (!in_cur) pi = 0;
        const unsigned pa = (pl.pa_bits >> (2 * pi)) & 3u, pb = (pl.pb_bits >> (2 * pi)) & 3u;
        // (wave-uniform by construction; said so explicitly: an offset the compiler cannot prove uniform turns every DMA
        // piece into a waterfall loop over the lanes' "different" values)
        Ahead s;
        s.kbA = (unsigned)__builtin_amdgcn_readfirstlane((int)((pa * (unsigned)kpp + (unsigned)kk) * 128u));
        s.kbB = (unsigned)__builtin_amdgcn_readfirstlane((int)((pb * (unsigned)kpp + (unsigned)kk) * 128u));
        s.baseA = (unsigned)__builtin_amdgcn_readfirstlane((int)(in_cur ? curA : nxtA));
        s.baseB = (unsigned)__builtin_amdgcn_readfirstlane((int)(in_cur ? curB : nxtB));
        return s;
    };
    // piece i (0..3) of A half h of stream position s -> slot X; piece i (0..5) of B
    auto dma_a = [&](int X, int h, int i, const Ahead& s) __attribute__((always_inline)) {
        if constexpr (abl_dma) return;
        const unsigned rowd = (unsigned)((i >> 1) * 128 + h * 64 + (i & 1) * 32);
        const unsigned dst = lds_base + P::A_OFF + X * P::A_SLOT + h * 16384 + (wave + 4 * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, to_lds(dst), 16, relA, s.baseA + rowd * lda2 + s.kbA, 0, 0);
    };
    auto dma_b = [&](int X, int i, const Ahead& s) __attribute__((always_inline)) {
        if constexpr (abl_dma) return;
        const unsigned dst = lds_base + P::B_OFF + X * P::B_SLOT + (wave + 4 * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, to_lds(dst), 16, relB, s.baseB + (unsigned)(i * 32) * ldb2 + s.kbB, 0, 0);
    };
    auto bar = [&]() __attribute__((always_inline)) {
        ZG_SB();
        if constexpr (!abl_bar) __builtin_amdgcn_s_barrier();
        ZG_SB();
    };

    // ---- a tile starts at its bias row (the reference pre-fills the output with the bias, src/ops.zig:24-29): bias[n] as
    // three bf16 terms in k positions 0..2 of a "weight" fragment, ones in the same positions of the other operand
    auto init_acc_from_bias = [&](int par) __attribute__((always_inline)) {
        int lane_b = lane;  // opaque copy: nothing derived from it is hoisted out of (and kept live across) the main loop
        asm volatile("" : "+v"(lane_b));
        const int l31 = lane_b & 31, hh = lane_b >> 5;
        const unsigned baddr = lds_base + P::BIAS_OFF + par * 1024 + (unsigned)(wc * (P::BN / 2) + l31) * 4u;
        float bv[3];
        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:128\n\tds_read_b32 %2, %3 offset:256\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2])
                     : "v"(baddr)
                     : "memory");
        u32x4 ones = {hh ? 0u : 0x3F803F80u, hh ? 0u : 0x00003F80u, 0u, 0u};
        static_for<3>([&](auto JT) {
            constexpr int j = decltype(JT)::value;
            unsigned hi, mid, lo;
            split3_pk(hh ? 0.0f : bv[j], 0.0f, hi, mid, lo);
            u32x4 wf = {(hi & 0xffffu) | (mid << 16), lo & 0xffffu, 0u, 0u};
            mfma_init<acc_reg(0 * 3 + j)>(wf, ones);
            mfma_init<acc_reg(1 * 3 + j)>(wf, ones);
            mfma_init<acc_reg(2 * 3 + j)>(wf, ones);
            mfma_init<acc_reg(3 * 3 + j)>(wf, ones);
        });
        ZG_SB();
    };

    // ---- the drain: state of the tile whose copy sits in v[64:255]
    float k1v = -2.0f * 1.4426950408889634f * 0.7978845608f, k2v = k1v * 0.044715f;  // gelu: x / (1 + 2^(x (k1 + k2 x^2)))
    asm volatile("" : "+v"(k1v), "+v"(k2v));
    unsigned st_w, st_r, g_off3[3], row16;  // staging write / read addresses, store offsets of the three 64-byte column groups
    {
        const int l31 = lane & 31, hh = lane >> 5;
        st_w = lds_base + P::ST_OFF + wave * P::ST_WAVE + (unsigned)l31 * P::ST_ROW + (unsigned)hh * 8u;
        st_r = lds_base + P::ST_OFF + wave * P::ST_WAVE + (unsigned)(lane >> 2) * P::ST_ROW + (unsigned)(lane & 3) * 16u;
        row16 = (unsigned)ldc * 32u;  // bytes of 16 output rows
    }
    const unsigned no_store_mask = (dbg & 1) ? 0x80000000u : 0u;
    auto set_drain_tile = [&](int dm0, int dn0) __attribute__((always_inline)) {
        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int col = dn0 + wc * (P::BN / 2) + ((lane_b & 3) + 4 * kk) * 8;
            const unsigned off = (unsigned)(dm0 + wr * 128 + (lane_b >> 2)) * (unsigned)(ldc * 2) + (unsigned)col * 2u;
            g_off3[kk] = (col < N ? off : 0x80000000u) | no_store_mask;  // (the output is under 2 GiB: launcher)
        }
    };
    // The 16 values of block u out of the hand-owned copy into compiler registers: the only block-specific code of the drain.
    float xx[16];
    auto load_block = [&](int u) __attribute__((always_inline)) {
        if constexpr (abl_lb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xx[r] = 1.0f;
            return;
        }
        switch (u) {
#define ZG_LB8(B, R)                                                                                                                       \
    asm volatile("v_mov_b32 %0, v%c8\n\tv_mov_b32 %1, v%c9\n\tv_mov_b32 %2, v%c10\n\tv_mov_b32 %3, v%c11\n\tv_mov_b32 %4, v%c12\n\t"        \
                 "v_mov_b32 %5, v%c13\n\tv_mov_b32 %6, v%c14\n\tv_mov_b32 %7, v%c15"                                                      \
                 : "=&v"(xx[R]), "=&v"(xx[R + 1]), "=&v"(xx[R + 2]), "=&v"(xx[R + 3]), "=&v"(xx[R + 4]), "=&v"(xx[R + 5]), "=&v"(xx[R + 6]),  \
                   "=&v"(xx[R + 7])                                                                                                        \
                 : "i"(dr_reg(B, R)), "i"(dr_reg(B, R + 1)), "i"(dr_reg(B, R + 2)), "i"(dr_reg(B, R + 3)), "i"(dr_reg(B, R + 4)),           \
                   "i"(dr_reg(B, R + 5)), "i"(dr_reg(B, R + 6)), "i"(dr_reg(B, R + 7)))
#define ZG_LB(B) \
    case B:      \
        ZG_LB8(B, 0); \
        ZG_LB8(B, 8); \
        break;
            ZG_LB(0) ZG_LB(1) ZG_LB(2) ZG_LB(3) ZG_LB(4) ZG_LB(5) ZG_LB(6) ZG_LB(7) ZG_LB(8) ZG_LB(9) ZG_LB(10) ZG_LB(11)
#undef ZG_LB
#undef ZG_LB8
            default: break;
        }
    };
    // Per K-step: block u = 3 i + j -> where its bf16 runs go in the staged m-tile, and where the m-tile's rows go in C
    unsigned st_wj = 0, soff_i = 0;
    auto set_block = [&](int u) __attribute__((always_inline)) {
        const int i = u / 3, j = u - 3 * i;
        st_wj = st_w + (unsigned)j * 64u;
        soff_i = (unsigned)(2 * i) * row16;
    };
    // one op of the program of variant V.  WCNT: fragment reads issued between the staged reads R and the wait W
    auto drain_op = [&](auto VT, auto QT, auto WCNT, float (&t)[16], u32x4 (&o)[6]) __attribute__((always_inline)) {
        constexpr int V = decltype(VT)::value, q = decltype(QT)::value;
        constexpr int kind = PG::kind(V, q), r = PG::sub(V, q);
        constexpr bool is_valu = kind <= PG::oG;
        if constexpr (kind == PG::oNop || (abl_valu && is_valu) || (abl_st && kind == PG::oS) || (abl_lds && !is_valu && kind != PG::oS)) {
            if constexpr (abl_valu && kind == PG::oA) t[r] = 1.0f;
        } else if constexpr (kind == PG::oR) {
            const unsigned st_r_l = st_r;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o[r]) : "v"(st_r_l), "i"((r % 3) * 64 + (r / 3) * 16 * P::ST_ROW) : "memory");
        } else if constexpr (kind == PG::oW) {
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]) : "i"(decltype(WCNT)::value));
        } else if constexpr (kind == PG::oS) {
            __builtin_amdgcn_raw_buffer_store_b128(o[r], rc, g_off3[r % 3], soff_i + (unsigned)(r / 3) * row16, ST_AUX);
        } else if constexpr (kind == PG::oA) {
            // (every op is a volatile asm statement: plain arithmetic common to several K-step bodies is hoisted by the compiler
            // in front of the branch between them — out of the MFMA shadows it was dealt to, and into 40 more live registers.
            // A transcendental's consumer is never the next instruction: stages run over all 16 values.)
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t[r]) : "v"(xx[r]));
        } else if constexpr (kind == PG::oB) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t[r]) : "v"(k2v), "v"(k1v));
        } else if constexpr (kind == PG::oC || kind == PG::oG) {
            asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t[r]) : "v"(xx[r]));
        } else if constexpr (kind == PG::oD) {
            asm volatile("v_exp_f32 %0, %0" : "+v"(t[r]));
        } else if constexpr (kind == PG::oE) {
            asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t[r]));
        } else if constexpr (kind == PG::oF) {
            asm volatile("v_rcp_f32 %0, %0" : "+v"(t[r]));
        } else {  // H_g: columns 32 j + 8 g + 4 hh + {0..3} of row l31 -> 8 bytes of the staged m-tile
            constexpr int g = r;
            const unsigned st_l = st_wj;
            unsigned p0_, p1_;
            if constexpr (GELU) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p0_) : "v"(t[4 * g]), "v"(t[4 * g + 1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1_) : "v"(t[4 * g + 2]), "v"(t[4 * g + 3]));
            } else {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p0_) : "v"(xx[4 * g]), "v"(xx[4 * g + 1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1_) : "v"(xx[4 * g + 2]), "v"(xx[4 * g + 3]));
            }
            const u32x2 pk = {p0_, p1_};
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(st_l), "v"(pk), "i"(g * 16) : "memory");
        }
    };
    // the drain ops of gap (J, m) of a K-step of variant V
    auto drain_gap = [&](auto VT, auto JT, auto MT, float (&t)[16], u32x4 (&o)[6]) __attribute__((always_inline)) {
        constexpr int V = decltype(VT)::value, J = decltype(JT)::value, m = decltype(MT)::value;
        if constexpr (V != kNone) {
            constexpr int q0 = gap_q0(J, m), n = gap_cap(J, m);
            // a wait W in this gap: main LDS items issued since the last staged read R (the idle slots between them are empty)
            constexpr int wcnt = [] {
                int c = 0;
                constexpr int q_w = PG::BLK + 6 + PG::IDLE, q_r = PG::BLK + 5;
                if (V == kTail && q_w >= q0 && q_w < q0 + n) {
                    int jj = J, mm = m;
                    for (;;) {
                        if (gap_q0(jj, mm) <= q_r) break;   // the last R sits in this gap: its main item came first
                        c += main_is_lds(jj, mm) ? 1 : 0;
                        if (mm == 0) {
                            if (jj == 0) break;
                            --jj;
                            mm = 5;
                        } else
                            --mm;
                    }
                }
                return c;
            }();
            static_for<n>([&](auto KT) { drain_op(VT, Ic<q0 + decltype(KT)::value>{}, Ic<wcnt>{}, t, o); });
        }
    };
    // the program of variant V back to back (no main loop around it): the exposed drain
    auto drain_all = [&](auto VT) __attribute__((always_inline)) {
        constexpr int V = decltype(VT)::value;
        float t[16];
        u32x4 o[6];
        static_for<PG::len(V)>([&](auto KT) {
            drain_op(VT, KT, Ic<0>{}, t, o);
            if constexpr ((decltype(KT)::value & 15) == 15) ZG_SB();
        });
        ZG_SB();
    };
    static_assert(PG::len(kTail) <= kstep_cap(), "one block's drain program must fit one K-step");

    Ahead s1, s2;  // stream positions t + 1 / t + 2 of the current K-step t
    // ---- one K-step (slot `slot`) of variant V behind a K-step of variant PV
    auto kstep = [&](auto VT, auto PVT) __attribute__((always_inline)) {
        constexpr int V = decltype(VT)::value, PV = decltype(PVT)::value;
        const unsigned X = (unsigned)slot, XO = X ^ 1u;
        float t[16];
        u32x4 o[6];
        if constexpr (abl_lds)
            for (int k = 0; k < 6; ++k) o[k] = u32x4{0u, 0u, 0u, 0u};
        static_for<8>([&](auto JT) {
            constexpr int J = decltype(JT)::value;
            constexpr int h = J >> 2, ks = J & 3, cb = J & 1, nb = cb ^ 1;
            constexpr int JN = (J + 1) & 7, hn = JN >> 2, ksn = JN & 3;
            constexpr int JP = (J + 7) & 7, VP = J == 0 ? PV : V;   // the step before, and the variant of its K-step
            if constexpr (J == 0) s2 = ahead(2);
            // ---- waits and barriers in front of the step's first MFMA
            ZG_SB();
            // A fragment tile 0 of this step: the main item of gap 0 of the step before
            if constexpr (!abl_lgkm) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(lds_after<GELU>(VP, JP, 0) < 15 ? lds_after<GELU>(VP, JP, 0) : 15) : "memory");
            if constexpr (J == 1) {
                // barrier 2: A half 1 of slot X ^ 1 (read in steps 3-6 of the K-step before) and B of slot X (read in its steps
                // 4-7) are read out; A half 1 of THIS K-step and B of the next have landed — only the four pieces of A half 0
                // of K-step t + 1 may still be in flight, and the drain stores issued BEHIND the last piece that must have
                // landed (step 3, gap 3 of the K-step before)
                constexpr int st = stores_in<GELU>(PV, 3, 3, 7, 5) + stores_in<GELU>(V, 0, 0, 0, 5);
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 + st) : "memory");
                bar();
            }
            if constexpr (J == 4) {
                // barrier 1: A half 0 of slot X (last read in step 2) is read out; A half 0 of K-step t + 1 has landed — the ten
                // pieces of steps 1-3 (and the drain stores since step 7, gap 2 of the K-step before) may still be in flight
                constexpr int st = stores_in<GELU>(PV, 7, 2, 7, 5) + stores_in<GELU>(V, 0, 0, 3, 5);
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(10 + st) : "memory");
                bar();
            }
            if constexpr (J == 7) {  // the A reads from here on are the next K-step's: other slot
#pragma unroll
                for (int k = 0; k < 4; ++k) a_addr[k] ^= (unsigned)P::A_SLOT;
            }
            ZG_SB();
            static_for<6>([&](auto MT) {
                constexpr int m = decltype(MT)::value, i2 = m / 3, j = m % 3;
                if constexpr (m == 3) {  // A fragment tile 1: the main item of gap 1 of the step before
                    constexpr int c = lds_after<GELU>(VP, JP, 1) + lds_upto<GELU>(V, J, 2);
                    if constexpr (!abl_lgkm) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(c < 15 ? c : 15) : "memory");
                    ZG_SB();
                }
                mfma_hand<acc_reg((2 * h + i2) * 3 + j), fb_reg(ks, j), fa_reg(cb, i2)>();
                ZG_SB();
                // ---- the gap behind MFMA m
                if constexpr (m <= 1) {
                    if constexpr (!abl_rd) ds_read_frag<fa_reg(nb, m), hn * 16384 + m * 4096>(a_addr[ksn]);
                } else if constexpr (J >= 4 && m >= 3) {
                    if constexpr (!abl_rd) ds_read_frag<fb_reg(J - 4, m - 3), (m - 3) * 4096>(b_addr[J - 4]);
                } else if constexpr (J >= 4 && m == 2) {
                    dma_a(X, 0, J - 4, s2);                                   // A half 0 of K-step t + 2 -> this slot
                } else if constexpr (J >= 1 && J <= 3 && m >= 2) {
                    constexpr int p = (J - 1) * 4 + (m - 2);
                    if constexpr (p < 4) dma_a(XO, 1, p, s1);                 // A half 1 of K-step t + 1 -> other slot
                    else if constexpr (p < 10) dma_b(X, p - 4, s2);           // B of K-step t + 2 -> this slot
                }
                drain_gap(VT, JT, MT, t, o);
                ZG_SB();
            });
        });
        // the next K-step's B reads come from this K-step's slot
        const unsigned db = X ? (unsigned)(-P::B_SLOT) : (unsigned)P::B_SLOT;  // b_addr points at slot X ^ 1 now
#pragma unroll
        for (int k = 0; k < 4; ++k) b_addr[k] -= db;
        slot ^= 1;
        s1 = s2;
    };

    auto next_tile = [&]() __attribute__((always_inline)) {
        idx = nidx;
        m0 = ntm * 256;
        n0 = ntn * P::BN;
        tile_par ^= 1;
        curA = nxtA;
        curB = nxtB;
        nidx = idx + gx;
        nxtA = kOob;
        nxtB = kOob;
        if (nidx < t_end) {
            tile_of(nidx, tiles_m, tiles_n, gw, ntm, ntn);
            nxtA = (unsigned)ntm * strideA;
            nxtB = (unsigned)ntn * strideB;
            fetch_bias(ntn * P::BN, tile_par ^ 1);  // a whole tile ahead of its use
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {  // K-step t -> t + 1; true at the end of the tile
        if (++kk_cur == kpp) {
            kk_cur = 0;
            ++pi_cur;
        }
        if (pi_cur < pl.npairs) return false;
        pi_cur = 0;
        return true;
    };

    // ---- prologue: the stream in steady-state order up to the start of K-step 0.  Per K-step t the stream carries
    // [A half 1 of t + 1, B of t + 2, A half 0 of t + 2]; before K-step 0 that is [B 0, A0 0] and [A1 0, B 1, A0 1].
    {
        fetch_bias(n0, 0);
        if (nidx < t_end) fetch_bias(ntn * P::BN, 1);
        const Ahead s0 = ahead(0);
        s1 = ahead(1);
        s2 = s1;
#pragma unroll
        for (int i = 0; i < 6; ++i) dma_b(0, i, s0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(0, 0, i, s0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(0, 1, i, s0);
#pragma unroll
        for (int i = 0; i < 6; ++i) dma_b(1, i, s1);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(1, 0, i, s1);
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // the bias rows, B and A half 0 of K-step 0 are in
        bar();
        init_acc_from_bias(0);
        // all B fragments of K-step 0 and the A fragments of its step 0
        static_for<4>([&](auto KT) {
            constexpr int ks = decltype(KT)::value;
            ds_read_frag<fb_reg(ks, 0), 0>(b_addr[ks]);
            ds_read_frag<fb_reg(ks, 1), 4096>(b_addr[ks]);
            ds_read_frag<fb_reg(ks, 2), 8192>(b_addr[ks]);
        });
        ds_read_frag<fa_reg(0, 0), 0>(a_addr[0]);
        ds_read_frag<fa_reg(0, 1), 4096>(a_addr[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ZG_SB();
#pragma unroll
        for (int k = 0; k < 4; ++k) b_addr[k] += (unsigned)P::B_SLOT;  // K-step 0's B reads are K-step 1's fragments: slot 1
        phase_stamp();
    }

    // The drain of the previous tile takes the first twelve K-steps of a tile, block u in K-step u; `unit` is the block of the
    // next K-step (12: none — the whole first tile of a workgroup), `pv` the variant of the K-step before.
    int unit = 12, pv = kNone;
    bool last = false;  // the last tile has been copied: one more trip for its drain (ONE site of the exposed drain in the code)
    for (;;) {
        if (!last) {
            if (unit < 12) {
                load_block(unit);
                set_block(unit);
                if (unit % 3 == 2) {
                    kstep(Ic<kTail>{}, Ic<kBlk>{});
                    pv = kTail;
                } else {
                    if (pv == kBlk) kstep(Ic<kBlk>{}, Ic<kBlk>{});
                    else if (pv == kTail) kstep(Ic<kBlk>{}, Ic<kTail>{});
                    else kstep(Ic<kBlk>{}, Ic<kNone>{});  // behind a tile boundary (nothing counted as issued later: stricter waits)
                    pv = kBlk;
                }
                ++unit;
            } else {
                if (pv == kTail) kstep(Ic<kNone>{}, Ic<kTail>{});
                else kstep(Ic<kNone>{}, Ic<kNone>{});
                pv = kNone;
            }
            if (!advance()) continue;
            phase_stamp();
        }
        // ---- end of a tile (or the trip behind the last one): what is left of the drain runs back to back — the rest of the
        // previous tile's when a tile has fewer than twelve K-steps, all of the last tile's
        for (; unit < 12; ++unit) {
            load_block(unit);
            set_block(unit);
            if (unit % 3 == 2) drain_all(Ic<kTail>{});
            else drain_all(Ic<kBlk>{});
        }
        if (last) break;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs retire before their accumulators are read
        static_for<192>([&](auto NT) { acc_to_drain<decltype(NT)::value>(); });
        ZG_SB();
        set_drain_tile(m0, n0);
        unit = 0;
        pv = kNone;
        if (idx + gx >= t_end) {
            last = true;
            continue;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the fragment reads of the last step 7: nothing of ours in flight below)
        init_acc_from_bias(tile_par ^ 1);
        next_tile();
        phase_stamp();
    }
    phase_stamp();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // zero-fill pieces of "no next tile" still write this workgroup's LDS
    if (stamp && lane == 0) {
        g_ov_stamps[0] = gridDim.x;
        g_ov_stamps[1 + 2 * bid] = t_start;
        g_ov_stamps[2 + 2 * bid] = __builtin_readcyclecounter();
        g_ov_stamps[513 + 2 * bid] = w_start;
        g_ov_stamps[514 + 2 * bid] = __builtin_amdgcn_s_memrealtime();
    }
}

template <bool GELU, int ST_AUX, int ABL = 0>
int launch_ov(const bf16_t* A, const bf16_t* B, const float* bias, bf16_t* C, int M, int N, const GemmPlanes& pl, int ldc, hipStream_t s) {
    using P = OV;
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ov_kernel<GELU, ST_AUX, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS));
        raised = true;
    }
    const int tiles_m = (M + 255) / 256, tiles_n = (N + P::BN - 1) / P::BN, n_tiles = tiles_m * tiles_n;
    const int gw_env = getenv("ZGPT2_GW") ? atoi(getenv("ZGPT2_GW")) : 0;
    int gw = gw_env > 0 ? gw_env : 8;
    if (gw > tiles_n) gw = tiles_n;
    const int cus_env = getenv("ZGPT2_GEMM_WGS") ? atoi(getenv("ZGPT2_GEMM_WGS")) : 0;  // tests: few workgroups, many tiles each
    const int cus = cus_env > 0 ? cus_env : 256;
    const int grid = n_tiles < cus ? n_tiles : cus;
    const unsigned dbg = (unsigned)(getenv("ZGPT2_GEMM_DBG") ? atoi(getenv("ZGPT2_GEMM_DBG")) : 0);
    unsigned pa2 = 0, pb2 = 0;
    for (int i = 0; i < pl.npairs; ++i) {
        pa2 |= ((pl.pa_bits >> (4 * i)) & 3u) << (2 * i);
        pb2 |= ((pl.pb_bits >> (4 * i)) & 3u) << (2 * i);
    }
    hipLaunchKernelGGL((gemm_ov_kernel<GELU, ST_AUX, ABL>), dim3(grid), dim3(256), P::LDS, s, A, B, bias, C, M, N,
                       (unsigned)pl.lda | ((unsigned)pl.ldb << 16), (unsigned)ldc | ((unsigned)pl.kpp << 20) | ((unsigned)pl.npairs << 28),
                       pa2 | (pb2 << 12) | ((unsigned)gw << 24), (unsigned)grid | (dbg << 10));
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace

// what the packed kernel arguments and the 2^31 "column out of range" store offset can express
bool gemm_ov_args_ok(int M, const GemmPlanes& pl, int ldc) {
    return pl.lda > 0 && pl.ldb > 0 && pl.lda < 65536 && pl.ldb < 65536 && ldc < (1 << 20) && pl.kpp >= 2 && pl.kpp < 256 && pl.npairs <= 6 &&
           (size_t)(M + 256) * (size_t)ldc * 2 < ((size_t)1 << 31);
}

int gemm_ov_stamps(unsigned long long* out, size_t n_words) {
    if (n_words > 1 + 4 * 256 + 16) n_words = 1 + 4 * 256 + 16;
    ZG_HIP(hipDeviceSynchronize());
    ZG_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ov_stamps), n_words * sizeof(unsigned long long)));
    return ZG_OK;
}

int launch_gemm_ov(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc, bool gelu,
                   hipStream_t s) {
    ZG_REQUIRE(gemm_ov_args_ok(M, pl, ldc), ZG_ERR_UNSUPPORTED, "gemm_ov: lda %d / ldb %d / ldc %d / K beyond the packed kernel arguments", pl.lda,
               pl.ldb, ldc);
    const int aux = getenv("ZGPT2_OV_AUX") ? atoi(getenv("ZGPT2_OV_AUX")) : 16;  // A/B of the store policy: 0 plain, 16 sc1 = write-through
    bf16_t* c = reinterpret_cast<bf16_t*>(C);
    if (gelu && aux == 16) {
        switch (getenv("ZGPT2_OV_ABL") ? atoi(getenv("ZGPT2_OV_ABL")) : 0) {
#define ZG_ABL(K) case K: return launch_ov<true, 16, K>(A, B, bias, c, M, N, pl, ldc, s);
            ZG_ABL(1) ZG_ABL(2) ZG_ABL(4) ZG_ABL(8) ZG_ABL(12) ZG_ABL(16) ZG_ABL(32) ZG_ABL(48) ZG_ABL(64) ZG_ABL(112) ZG_ABL(15) ZG_ABL(127) ZG_ABL(128) ZG_ABL(160) ZG_ABL(240)
#undef ZG_ABL
            default: break;
        }
    }
    if (gelu) return aux == 16 ? launch_ov<true, 16>(A, B, bias, c, M, N, pl, ldc, s) : launch_ov<true, 0>(A, B, bias, c, M, N, pl, ldc, s);
    return aux == 16 ? launch_ov<false, 16>(A, B, bias, c, M, N, pl, ldc, s) : launch_ov<false, 0>(A, B, bias, c, M, N, pl, ldc, s);
}

}  // namespace zg
