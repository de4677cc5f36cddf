// gemm_s4.hip — the large-batch Linear on the CDNA4 matrix cores, second generation:
//   C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (optionally GELU), bf16 operands, fp32 accumulate.
// Linear.forward of the reference (src/ops.zig:21-46: cblas_sgemm RowMajor/NoTrans/Trans, bias pre-filled) for
// M >> 1: both operands K-contiguous, exactly ops.Linear's layouts (x is [M, in], weight is [out, in]).
//
// Structure (what the measurements of round 3 asked for, tools/microbench/dma_intake.hip):
//   * FOUR waves per workgroup, one per SIMD, each with the whole 512-register file: a 256 x BN tile (BN = 192 or
//     256) as 2 x 2 wave tiles of 128 x BN/2, v_mfma_f32_32x32x16_bf16 (twice the flops per operand register of the
//     16x16x32 form), operands SWAPPED (A operand = weight rows) so that a lane's accumulator registers are
//     runs of 4 consecutive output columns of one output row.
//   * Every wave runs ONE software-pipelined instruction stream: the fragment reads of k-slice j + 1 are issued
//     in front of the MFMAs of k-slice j (two fragment buffers), the LDS-DMA pieces (buffer_load ... lds) of the
//     K-step two ahead are spread between the MFMAs, and the workgroup meets at only TWO barriers per K-step
//     (the first-generation kernel: eight, each behind an exposed fragment-read latency).
//   * LDS: two K-step slots; a K-step is three DMA units — A half 0 (rows 0-63 of every wave row), B, A half 1 —
//     consumed as two phases of four k-slices (half 0 x B, half 1 x B), so that a unit is free, and its
//     replacement in flight, more than a K-step before it is needed.
//   * Persistent: <= 256 workgroups walk XCD-contiguous ranges of a column-banded tile order; the DMA stream
//     continues across tile boundaries (the next tile's first K-steps land under the epilogue).
#include <stdlib.h>

#include <type_traits>

#include "prefill_epi.h"
#include "zg_kernels.h"

namespace zg {

// diagnostic (dbg bit 256): [0] grid; [1 + 2 b ..] shader-clock {start, end} of workgroup b's wave 0; [513 + 2 b ..] the same two moments on the
// constant 100 MHz clock (s_memrealtime: wall time inside the launch, whatever the shader clock does); [1025 ..] workgroup 0's phase
// stamps (shader clock): start, prologue done, then {main loop done, epilogue done} per tile
__device__ unsigned long long g_s4_stamps[1 + 4 * 256 + 16];
__device__ unsigned g_s4_fault;  // a stream-K consumer ran out of polls (gemm_s4_fault())

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16v;
typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(2))) float f32x2v;

// gelu(x) = x / (1 + exp(-2u)), u = x * 0.7978845608 * (1 + 0.044715 x^2)  (src/ops.zig:225); the -2 log2(e)
// factor is folded into the polynomial so that the exponential is a bare v_exp_f32
__device__ __forceinline__ float gelu1(float x) {
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
    const float arg = x * fmaf(x * x, k2, k1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
}

__device__ __forceinline__ f32x2v gelu2(f32x2v x) {
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
    const f32x2v p = __builtin_elementwise_fma(x * x, f32x2v{k2, k2}, f32x2v{k1, k1});
    const f32x2v arg = x * p;
    const f32x2v d = f32x2v{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)} + f32x2v{1.0f, 1.0f};
    return x * f32x2v{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

template <int NT_>
struct S4 {
    static constexpr int NT = NT_;              // 32-column MFMA tiles per wave: 3 or 4
    static constexpr int BM = 256, BN = 64 * NT;
    static constexpr int A_SLOT = 256 * 128, B_SLOT = BN * 128;
    static constexpr int A_OFF = 0, B_OFF = 2 * A_SLOT;  // [A slot 0][A slot 1][B slot 0][B slot 1]
    static constexpr int BIAS_OFF = 2 * A_SLOT + 2 * B_SLOT;  // two tiles' bias rows (fp32), by tile parity
    // bf16 epilogue staging, per wave: one 32-row m-tile x the wave's BN / 2 columns, rows padded by 16 B
    static constexpr int ST_ROW = NT * 64 + 16, ST_WAVE = 32 * ST_ROW, ST_OFF = BIAS_OFF + 2 * 1024;
    static constexpr int LDS = ST_OFF + 4 * ST_WAVE;      // 140 / 164 KiB
    static constexpr int PB = BN / 32;          // B pieces (8 rows x 128 B) per wave per K-step: 6 or 8
    static constexpr int W3 = 8 + PB;           // pieces that may still be in flight when A half 1 of this K-step must have landed
};

#define ZG_SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ lds_ptr_t to_lds(unsigned byte_addr) { return (lds_ptr_t)(size_t)byte_addr; }

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

struct Ahead {  // where a K-step of the DMA stream comes from
    unsigned baseA, baseB, kbA, kbB;
};

// Fragment reads are hand-issued so that they stay where they are written (nothing orders the compiler's own LDS loads
// against s_barrier or an LDS-DMA).  J = 4 h + ks: A half h, k-slice ks; I: 32-row tile.
template <int NT, int X, int J, int I>
__device__ __forceinline__ void read_frag_a(bf16x8& f, const unsigned (&a_addr)[4]) {
    using P = S4<NT>;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a_addr[J & 3]), "i"(X * P::A_SLOT + (J >> 2) * 16384 + I * 4096));
}
template <int NT, int X, int J, int I>
__device__ __forceinline__ void read_frag_b(bf16x8& f, const unsigned (&b_addr)[4]) {
    using P = S4<NT>;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(b_addr[J & 3]), "i"(X * P::B_SLOT + I * 4096));
}

// ---- The accumulators are OWNED BY HAND: tile (i, j) of a wave lives in a[16 (i NT + j) : +15], written only by the inline-asm
// MFMAs below and read / zeroed only by the inline-asm moves of the epilogue.  Left to the register allocator, a change in
// the epilogue's register pressure made it park accumulators in VGPRs and shuffle them with v_accvgpr_mov inside the K loop
// (dependent on the MFMAs: +14k cycles per launch).  Every statement that touches them names all of them as clobbered, so
// the compiler keeps nothing of its own there across any of them; what remains to audit after an edit is that no
// compiler-generated v_accvgpr_* names a0..a191 (tools/README: grep recipe).  Hazards are ours too: MFMA result ->
// v_accvgpr_read needs the pipeline drained (s_nop in acc_settle), v_accvgpr_write -> MFMA source C two wait states.
#define ZG_ACC_CLOBBERS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191"
template <int BASE>
__device__ __forceinline__ void mfma_acc(const bf16x8& a_op, const bf16x8& b_op) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a_op), "v"(b_op), "i"(BASE), "i"(BASE + 15) : ZG_ACC_CLOBBERS);
}
template <int BASE>
__device__ __forceinline__ void acc_read8(float* v) {  // eight accumulator registers, one statement
    asm volatile(
        "v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c9]\n\tv_accvgpr_read_b32 %2, a[%c10]\n\tv_accvgpr_read_b32 %3, a[%c11]\n\t"
        "v_accvgpr_read_b32 %4, a[%c12]\n\tv_accvgpr_read_b32 %5, a[%c13]\n\tv_accvgpr_read_b32 %6, a[%c14]\n\tv_accvgpr_read_b32 %7, a[%c15]"
        : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
        : "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3), "i"(BASE + 4), "i"(BASE + 5), "i"(BASE + 6), "i"(BASE + 7)
        : ZG_ACC_CLOBBERS);
}
template <int BASE>
__device__ __forceinline__ void acc_read16(float (&v)[16]) {  // the 16 registers of one 32 x 32 tile
    acc_read8<BASE>(v);
    acc_read8<BASE + 8>(v + 8);
}
template <int BASE>
__device__ __forceinline__ void acc_write16(const u32x4 (&v)[4]) {  // sixteen accumulator registers from four loaded quads (SK)
    asm volatile(
        "v_accvgpr_write_b32 a[%c16], %0\n\tv_accvgpr_write_b32 a[%c17], %1\n\tv_accvgpr_write_b32 a[%c18], %2\n\tv_accvgpr_write_b32 a[%c19], %3\n\t"
        "v_accvgpr_write_b32 a[%c20], %4\n\tv_accvgpr_write_b32 a[%c21], %5\n\tv_accvgpr_write_b32 a[%c22], %6\n\tv_accvgpr_write_b32 a[%c23], %7\n\t"
        "v_accvgpr_write_b32 a[%c24], %8\n\tv_accvgpr_write_b32 a[%c25], %9\n\tv_accvgpr_write_b32 a[%c26], %10\n\tv_accvgpr_write_b32 a[%c27], %11\n\t"
        "v_accvgpr_write_b32 a[%c28], %12\n\tv_accvgpr_write_b32 a[%c29], %13\n\tv_accvgpr_write_b32 a[%c30], %14\n\tv_accvgpr_write_b32 a[%c31], %15"
        :
        : "v"(v[0][0]), "v"(v[0][1]), "v"(v[0][2]), "v"(v[0][3]), "v"(v[1][0]), "v"(v[1][1]), "v"(v[1][2]), "v"(v[1][3]), "v"(v[2][0]), "v"(v[2][1]),
          "v"(v[2][2]), "v"(v[2][3]), "v"(v[3][0]), "v"(v[3][1]), "v"(v[3][2]), "v"(v[3][3]), "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3),
          "i"(BASE + 4), "i"(BASE + 5), "i"(BASE + 6), "i"(BASE + 7), "i"(BASE + 8), "i"(BASE + 9), "i"(BASE + 10), "i"(BASE + 11), "i"(BASE + 12),
          "i"(BASE + 13), "i"(BASE + 14), "i"(BASE + 15)
        : ZG_ACC_CLOBBERS);
}
// A tile starts at its bias (the reference pre-fills the output with the bias, src/ops.zig:24-29).  The four 32 x 32 tiles of
// one column tile j share the bias pattern of their 16 registers: it goes into the scratch accumulator a[192:207] once and
// four MFMAs of zero operands copy it (0 x 0 + C; destination and C of an MFMA must both be accumulator registers) — on
// the idle matrix pipe, instead of 64 v_accvgpr_writes now and 32 packed adds in the epilogue.  One statement, so that the
// compiler can hold nothing in a[192:207] across it.
template <int B0, int B1, int B2, int B3>
__device__ __forceinline__ void acc_init4(const bf16x8& zero_op, const f32x4v& g0, const f32x4v& g1, const f32x4v& g2, const f32x4v& g3) {
    asm volatile(
        "v_accvgpr_write_b32 a192, %1\n\tv_accvgpr_write_b32 a193, %2\n\tv_accvgpr_write_b32 a194, %3\n\tv_accvgpr_write_b32 a195, %4\n\t"
        "v_accvgpr_write_b32 a196, %5\n\tv_accvgpr_write_b32 a197, %6\n\tv_accvgpr_write_b32 a198, %7\n\tv_accvgpr_write_b32 a199, %8\n\t"
        "v_accvgpr_write_b32 a200, %9\n\tv_accvgpr_write_b32 a201, %10\n\tv_accvgpr_write_b32 a202, %11\n\tv_accvgpr_write_b32 a203, %12\n\t"
        "v_accvgpr_write_b32 a204, %13\n\tv_accvgpr_write_b32 a205, %14\n\tv_accvgpr_write_b32 a206, %15\n\tv_accvgpr_write_b32 a207, %16\n\t"
        "s_nop 3\n\t"
        "v_mfma_f32_32x32x16_bf16 a[%c17:%c18], %0, %0, a[192:207]\n\t"
        "v_mfma_f32_32x32x16_bf16 a[%c19:%c20], %0, %0, a[192:207]\n\t"
        "v_mfma_f32_32x32x16_bf16 a[%c21:%c22], %0, %0, a[192:207]\n\t"
        "v_mfma_f32_32x32x16_bf16 a[%c23:%c24], %0, %0, a[192:207]\n\t"
        "s_nop 7"
        :
        : "v"(zero_op), "v"(g0[0]), "v"(g0[1]), "v"(g0[2]), "v"(g0[3]), "v"(g1[0]), "v"(g1[1]), "v"(g1[2]), "v"(g1[3]), "v"(g2[0]), "v"(g2[1]),
          "v"(g2[2]), "v"(g2[3]), "v"(g3[0]), "v"(g3[1]), "v"(g3[2]), "v"(g3[3]), "i"(B0), "i"(B0 + 15), "i"(B1), "i"(B1 + 15), "i"(B2),
          "i"(B2 + 15), "i"(B3), "i"(B3 + 15)
        : ZG_ACC_CLOBBERS, "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206",
          "a207");
}
template <int REG>
__device__ __forceinline__ void acc_zero(void) {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(REG) : ZG_ACC_CLOBBERS);
}
__device__ __forceinline__ void acc_settle(void) {  // after the last MFMA, before the first read: the 8-pass pipeline drains
    asm volatile("s_nop 15\n\ts_nop 15" ::: ZG_ACC_CLOBBERS);
}

template <int NT, int X>
__device__ __forceinline__ void read_b_all(bf16x8 (&fb)[4][NT], const unsigned (&b_addr)[4]) {
    using P = S4<NT>;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[ks][j]) : "v"(b_addr[ks]), "i"(X * P::B_SLOT + j * 4096));
}

// KIND: what a finished tile becomes.
//   S4_PLAIN    bias (+ GELU), fp32 or bf16 rows of C[M][ldc]                                     (Linear.forward, src/ops.zig:21-46)
//   S4_PARTIAL  fp32 partial sums of K slice s into slab s of C[slices][M][ldc], no bias: the whole-prompt Linears whose N gives
//               too few tiles (N = n_embed) slice K over the tile list; prefill.hip's reduce kernels sum the slabs in fixed
//               order and finish (bias, residual, LayerNorm + plane split)
//   S4_QKV      bias, fp32 rows of qkv[M][3E] AND the cache append of src/ops.zig:152-157 for the K / V columns
//   S4_SPLIT3   bias + GELU, then the exact three-term bf16 split as planes C[M][3N] = [hi | mid | lo]  (src/main.zig:79-80 feeding
//               the next Linear's A operand)

template <int NT, int KIND, bool GELU, bool OUT_BF16, bool SK = false>
__global__ __launch_bounds__(256, 1) void gemm_s4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         const float* __restrict__ bias, void* __restrict__ C, int M, int N,
                                                         unsigned p0, unsigned p1, unsigned p2, unsigned p3, const PrefillQkv qa) {
    // All 14 dwords of the arguments are preloaded into SGPRs at wave launch (zg_common.h ZG_PIN): the first version passed the
    // plane description and the tile counts behind them and began with a scalar round trip to the kernarg segment — cold for
    // every launch — before its first DMA.  p0 = lda | ldb << 16; p1 = ldc | K-steps per plane << 20 | plane pairs << 28;
    // p2 = A plane of pair i in bits [2i, 2i + 2) | B planes << 12 | band width << 24; p3 = workgroups | dbg << 10 | K slices << 20 |
    // B planes are matrices << 31.
    // (qa — the cache description of S4_QKV — lies behind the preloaded dwords: its fields are fetched under the first DMA)
    using P = S4<NT>;
    GemmPlanes pl;
    pl.lda = (int)(p0 & 0xffffu);
    pl.ldb = (int)(p0 >> 16);
    const int ldc = (int)(p1 & 0xfffffu);
    pl.kpp = (int)((p1 >> 20) & 0xffu);
    pl.npairs = (int)(p1 >> 28);
    pl.pa_bits = p2 & 0xfffu;
    pl.pb_bits = (p2 >> 12) & 0xfffu;
    const int gw = (int)(p2 >> 24);
    const int dbg = (int)((p3 >> 10) & 0x3ffu);
    const int n_sl = KIND == S4_PARTIAL ? (int)((p3 >> 20) & 0xffu) : 1;  // K slices: slice s of a tile walks K-steps [s kps, (s + 1) kps) of every plane
    const bool b_major = (p3 >> 31) != 0u;  // B = three plane matrices [3][N][ldb] (fp32 weights of the model) instead of planes side by side in a row
    const int tiles_m = (M + 255) >> 8, tiles_n = (N + P::BN - 1) / P::BN;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)lds;

    // ---- this workgroup's tiles: XCD x = bid % 8 owns a contiguous range of the banded order
    const int tmn = tiles_m * tiles_n, n_tiles = tmn * n_sl, G = (int)(p3 & 0x3ffu), bid = blockIdx.x;
    const int nx = G < 8 ? G : 8;
    const int xcd = bid % nx, loc = bid / nx;
    const int gx = G / nx + (xcd < G % nx ? 1 : 0);
    const int q8 = n_tiles / nx, r8 = n_tiles % nx;
    const int t_begin = xcd * q8 + min(xcd, r8), t_end = t_begin + q8 + (xcd < r8 ? 1 : 0);
    // The workgroup's work items: item k is tile t_begin + loc + k gx of its XCD's range.  SK (stream-K hand-over, S4_QKV when the
    // tiles are 1.5 rounds of the workgroups — c_attn of eight 1023-token prompts: 384 tiles on 256 CUs, the second round half
    // empty): G workgroups, R G + G / 2 tiles.  Every workgroup walks R whole tiles and HALF of one of the last G / 2: the K-steps
    // [0, kpp / 2) of every plane (the producer: even place in its XCD, item 0 — it then spills its 192 accumulators per lane to
    // sk_ws and raises a per-wave flag) or [kpp / 2, kpp) (the consumer: odd place — dispatched behind its producer — last item:
    // its accumulators START at the producer's partial, which left more than a whole tile earlier, and it runs the epilogue).
    int it = 0;
    const int sk_R = SK ? tmn / G : 0, sk_gx = G >> 3;  // (SK: G % 16 == 0, tmn = sk_R G + G / 2: the launcher's conditions)
    const bool sk_cons = SK && (loc & 1);
    const int sk_sh = xcd * (sk_gx >> 1) + (loc >> 1);  // which of the G / 2 shared tiles
    auto item = [&](int k, int& tile, int& half, int& role) {  // role 0 whole tile, 1 producer half, 2 consumer half; false: no item k
        if constexpr (!SK) {
            tile = t_begin + loc + k * gx;
            half = 0;
            role = 0;
            return tile < t_end;
        } else {
            if (k > sk_R) return false;
            if (sk_cons ? k == sk_R : k == 0) {
                tile = sk_R * G + sk_sh;
                half = sk_cons ? 1 : 0;
                role = sk_cons ? 2 : 1;
            } else {
                tile = xcd * (sk_R * sk_gx) + loc + (sk_cons ? k : k - 1) * sk_gx;
                half = 0;
                role = 0;
            }
            return true;
        }
    };
    int idx, half_cur = 0, role_cur = 0;
    if (!item(0, idx, half_cur, role_cur)) return;
    const bool stamp = (dbg & 256) && wave == 0 && bid < 256;
    unsigned long long t_start = 0, w_start = 0;
    int n_phase = 0;
    if (stamp) {
        t_start = __builtin_readcyclecounter();
        w_start = __builtin_amdgcn_s_memrealtime();
    }
    auto phase_stamp = [&]() {
        if (stamp && bid == 0 && n_phase < 16) {
            if (lane == 0) g_s4_stamps[1025 + n_phase] = __builtin_readcyclecounter();
            ++n_phase;
        }
    };
    phase_stamp();

    // ---- DMA sources.  A piece = 8 unit rows x 128 B, written lane-linearly (lane -> row lane / 8, 16-B position
    // lane % 8); position p of LDS row R holds source chunk p ^ ((R >> 1) & 7) (the swizzle the fragment reads undo).
    // A wave issues pieces q = wave + 4 i of every unit; (R >> 1) & 7 = ((q & 1) << 2) | (lane >> 4) does not depend on i.
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * pl.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)N * pl.ldb * 2) * (b_major ? 3u : 1u), 0x00020000);
    // The tile's bias row travels through LDS (wave 0 fetches it by LDS-DMA at the start of the tile; columns past N and a
    // null bias read as zero): the epilogue must not issue a global load — the compiler would wait for it with vmcnt(0),
    // i.e. drain the stores and the next tile's DMA — and 48 registers of bias held across the epilogue make the register
    // allocator shuffle accumulators inside the K loop.
    const __amdgpu_buffer_rsrc_t rbias =
        __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias != nullptr ? (unsigned)N * 4u : 0u, 0x00020000);
    int tile_par = 0;  // parity of the current tile: its bias row is in buffer tile_par, the next tile's in the other one
    auto fetch_bias = [&](int n0_, int par) {
        if (wave == 0) {
#pragma unroll
            for (int p = 0; p < P::BN / 64; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, to_lds(lds_base + P::BIAS_OFF + par * 1024 + p * 256), 4, (unsigned)lane * 4u,
                                                         (unsigned)(n0_ + 64 * p) * 4u, 0, 0);
        }
    };
    const unsigned lda2 = (unsigned)pl.lda * 2u, ldb2 = (unsigned)pl.ldb * 2u;
    const int csrc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const unsigned relA = (unsigned)(wave * 8 + (lane >> 3)) * lda2 + (unsigned)csrc * 16u;
    const unsigned relB = (unsigned)(wave * 8 + (lane >> 3)) * ldb2 + (unsigned)csrc * 16u;

    // ---- fragment reads: lane -> row lane & 31 of a 32-row MFMA tile, k-chunk 2 ks + (lane >> 5) of the row's 8
    const int l31 = lane & 31, hh = lane >> 5;
    const unsigned swz = (unsigned)((hh ^ ((l31 >> 1) & 7)) << 4);
    const unsigned a_addr0 = lds_base + P::A_OFF + (unsigned)(wr * 64 + l31) * 128u + swz;
    const unsigned b_addr0 = lds_base + P::B_OFF + (unsigned)(wc * (P::BN / 2) + l31) * 128u + swz;
    const unsigned a_addr[4] = {a_addr0, a_addr0 ^ 32u, a_addr0 ^ 64u, a_addr0 ^ 96u};
    const unsigned b_addr[4] = {b_addr0, b_addr0 ^ 32u, b_addr0 ^ 64u, b_addr0 ^ 96u};

    // ---- K-steps walk the plane pairs: step kt = pair * kpp + kk multiplies A plane pa[pair] with B plane pb[pair]
    const int kpp = pl.kpp;                                   // K-steps per plane (the plane stride of both operands)
    const int kps_full = KIND == S4_PARTIAL ? kpp / n_sl : kpp;  // ... and those one tile walks (SK: a half item walks kpp / 2)
    int kps = (SK && role_cur != 0) ? kpp >> 1 : kps_full;
    int pi_cur = 0, kk_cur = 0;
    int tm, tn, sl = 0;
    auto locate = [&](int i, int& tm_, int& tn_, int& sl_) {  // the tile list: slice-major, then the banded order
        if constexpr (KIND == S4_PARTIAL) {
            sl_ = i / tmn;
            i -= sl_ * tmn;
        }
        tile_of(i, tiles_m, tiles_n, gw, tm_, tn_);
    };
    locate(idx, tm, tn, sl);
    const unsigned strideA = 256u * lda2, strideB = (unsigned)P::BN * ldb2, strideK = (unsigned)kps_full * 128u;
    const unsigned halfK = (unsigned)(kpp >> 1) * 128u;  // SK: where the consumer's half of a plane begins
    const unsigned plane_b = b_major ? (unsigned)N * ldb2 : (unsigned)kpp * 128u;  // from one B plane to the next
    unsigned curA = (unsigned)tm * strideA + (unsigned)sl * strideK + (unsigned)half_cur * halfK;
    unsigned curB = (unsigned)tn * strideB + (unsigned)sl * strideK + (unsigned)half_cur * halfK;
    int m0 = tm * 256, n0 = tn * P::BN;
    constexpr unsigned kOob = 0x80000000u;  // tile base of "no next tile": every lane out of range -> zero fill
    unsigned nxtA = kOob, nxtB = kOob;
    int nidx = 0, ntm = 0, ntn = 0, nsl = 0, half_nxt = 0, role_nxt = 0;
    bool has_nxt = item(1, nidx, half_nxt, role_nxt);
    if (has_nxt) {
        locate(nidx, ntm, ntn, nsl);
        nxtA = (unsigned)ntm * strideA + (unsigned)nsl * strideK + (unsigned)half_nxt * halfK;
        nxtB = (unsigned)ntn * strideB + (unsigned)nsl * strideK + (unsigned)half_nxt * halfK;
    }
    auto ahead = [&](int d) {  // K-step t + d of the stream (d <= 2 <= kps); runs on into the next tile
        int kk = kk_cur + d, pi = pi_cur;
        if (kk >= kps) {
            kk -= kps;
            ++pi;
        }
        const bool in_cur = pi < pl.npairs;
        if (!in_cur) pi = 0;
        const unsigned pa = (pl.pa_bits >> (2 * pi)) & 3u, pb = (pl.pb_bits >> (2 * pi)) & 3u;
        Ahead s;
        s.kbA = (pa * (unsigned)kpp + (unsigned)kk) * 128u;
        s.kbB = pb * plane_b + (unsigned)kk * 128u;
        s.baseA = in_cur ? curA : nxtA;
        s.baseB = in_cur ? curB : nxtB;
        return s;
    };
    // piece i (0..3) of A half h of stream position s -> slot X
    // Cache policy of the bf16 output stores: sc1 = write-through.  With plain stores the launch ends with the L2s writing
    // their dirty lines back — 4.2 us between the last workgroup's end and the next launch's first workgroup against 2.0 us
    // with write-through stores, at +0.6 us inside the launch (41.5 -> 40.2 us at M = 8192, profiles/round4_gemm_a.txt; nt: 40.7).
    constexpr int ST_AUX = 16;
    auto dma_a = [&](int X, int h, int i, const Ahead& s) {
        const unsigned rowd = (unsigned)((i >> 1) * 128 + h * 64 + (i & 1) * 32);
        const unsigned dst = lds_base + P::A_OFF + X * P::A_SLOT + h * 16384 + (wave + 4 * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, to_lds(dst), 16, relA, s.baseA + rowd * lda2 + s.kbA, 0, 0);
    };
    auto dma_b = [&](int X, int i, const Ahead& s) {
        const unsigned dst = lds_base + P::B_OFF + X * P::B_SLOT + (wave + 4 * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, to_lds(dst), 16, relB, s.baseB + (unsigned)(i * 32) * ldb2 + s.kbB, 0, 0);
    };
    // accumulators: a[0 : 16 * 4 * NT - 1], owned by hand (see mfma_acc); start at zero
    static_assert(NT == 3, "the hand-owned accumulator file is laid out for NT = 3 (192 registers)");
    // every tile starts at its bias row (buffer `par` of the LDS bias area): 12 MFMAs, see acc_init
    auto init_acc_from_bias = [&](int par) {
        const unsigned baddr = lds_base + P::BIAS_OFF + par * 1024 + (unsigned)(wc * (P::BN / 2) + 4 * (lane >> 5)) * 4u;
        bf16x8 zop;
#pragma unroll
        for (int e = 0; e < 8; ++e) zop[e] = (__bf16)0.0f;
        static_for<NT>([&](auto JT) {
            constexpr int j = decltype(JT)::value;
            const unsigned baddr_l = baddr;
            f32x4v b4[4];  // columns 32 j + 8 g + 4 hh + {0..3}: registers 4 g .. 4 g + 3 of the tile
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b4[g]) : "v"(baddr_l), "i"((j * 32 + 8 * g) * 4));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
            acc_init4<16 * (0 * NT + j), 16 * (1 * NT + j), 16 * (2 * NT + j), 16 * (3 * NT + j)>(zop, b4[0], b4[1], b4[2], b4[3]);
        });
    };
    // Fragments.  A: two buffers by step parity (2 tiles of 32 rows, one k-slice).  B: the WHOLE K-step of this wave's
    // columns (4 k-slices x NT tiles), two buffers by K-step parity — B is read from LDS once per K-step (both
    // phases multiply the same B fragments), during steps 3..6 of the K-step before.
    bf16x8 fa[2][2], fb[2][4][NT];

    auto bar = [&]() {
        ZG_SB();
        __builtin_amdgcn_s_barrier();
        ZG_SB();
    };
    // all B fragments + the A fragments of step 0 of the K-step in slot X (kernel start: nothing else is in flight)
    auto read_kstep_head = [&](auto XT) {
        constexpr int X = decltype(XT)::value;
        read_b_all<NT, X>(fb[X], b_addr);
        read_frag_a<NT, X, 0, 0>(fa[0][0], a_addr);
        read_frag_a<NT, X, 0, 1>(fa[0][1], a_addr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ZG_SB();
    };
    // End of a tile: the fragment reads of steps 3-7 of its last K-step (the next tile's first fragments) are in flight.
    // Their registers must not be live across the epilogue — its register pressure would make the compiler move them,
    // possibly before the data has landed, and 112 live fragment registers on top of the epilogue's own make it spill.
    // So: wait until they have all landed (in registers nobody reads), run the epilogue, and read the next K-step's
    // fragments again afterwards (read_kstep_head: ~14 reads and one exposed LDS latency per tile).
    auto settle = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ZG_SB();
    };

    // ---- deferred stores (bf16 output).  All 256 CUs reach their epilogue together, and 98 KB per CU in one burst is
    // bounded by the chip's write bandwidth (~4 us per tile), with the next tile's DMA queued behind it in the same memory
    // pipeline.  So only the first half of a tile's stores leaves from the epilogue (hidden under its arithmetic); the
    // second half waits in registers and leaves one store per K-step of the NEXT tile, behind barrier 1 (no hand-issued
    // LDS read is in flight there).  Whatever is still pending at the next epilogue, or at the end, is flushed.
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    // MEASURED AND SWITCHED OFF (NPEND = 0): a store issued inside the K loop sits among the DMA pieces on the wave's vmcnt
    // counter, completes late under the chip-wide write burst, and every counted wait behind it stalls (50.4 against 45.2 us).
    constexpr int NPEND = 0;  // OUT_BF16 ? 4 * NT : 0
    u32x4 pend[NPEND > 0 ? NPEND : 1];
    unsigned pend_row = 0, no_store_mask = (dbg & 1) ? 0xFFFFFFFFu : 0u;  // byte offset of (row l31 of m-tile 0, wave's column 0)
    int pend_col = 0, pend_n = NPEND;                                       // wave's first column + 8 hh; stores issued so far
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(C, 0, (unsigned)((size_t)n_sl * M * ldc * ESZ), 0x00020000);
    auto store_q = [&](auto QT, const u32x4& o, unsigned row0, int col0) {  // store q = (j * 4 + i) * 2 + p of a tile
        constexpr int q = decltype(QT)::value, i = (q / 2) % 4, j = (q / 2) / 4, pp = q % 2;
        const int col = col0 + j * 32 + 16 * pp;
        const unsigned off = (col < N ? row0 + (unsigned)(i * 32) * (unsigned)(ldc * ESZ) + (unsigned)(col * ESZ) : 0xFFFFFFFFu) | no_store_mask;
        __builtin_amdgcn_raw_buffer_store_b128(o, rc, off, 0, 0);
    };
    auto emit_pending = [&](int k) {  // pending store k (uniform)
        if constexpr (NPEND > 0) {
            switch (k) {
#define ZG_PEND_CASE(K) \
    case K: \
        if constexpr (K < NPEND) store_q(Ic<NPEND + (K < NPEND ? K : 0)>{}, pend[K < NPEND ? K : 0], pend_row, pend_col); \
        break;
                ZG_PEND_CASE(0) ZG_PEND_CASE(1) ZG_PEND_CASE(2) ZG_PEND_CASE(3) ZG_PEND_CASE(4) ZG_PEND_CASE(5) ZG_PEND_CASE(6) ZG_PEND_CASE(7)
                ZG_PEND_CASE(8) ZG_PEND_CASE(9) ZG_PEND_CASE(10) ZG_PEND_CASE(11) ZG_PEND_CASE(12) ZG_PEND_CASE(13) ZG_PEND_CASE(14) ZG_PEND_CASE(15)
#undef ZG_PEND_CASE
                default: break;
            }
        }
    };
    auto flush_pending = [&]() {
        if constexpr (NPEND > 0)
            for (; pend_n < NPEND; ++pend_n) emit_pending(pend_n);
    };

    Ahead s1, s2;  // stream positions t + 1 / t + 2 of the current K-step t
    // One step = one k-slice (16 of K) of one phase: 2 x NT MFMAs, and behind each MFMA at most ONE other item
    // (a fragment read or a DMA piece: two ds_read_b128 in one MFMA shadow do not fit — tools/microbench/mfma_lds.hip).
    // J = 4 h + ks.  Per K-step t (slot X):
    //   steps 0-2   DMA: B of K-step t + 2 -> slot X, A half 1 of K-step t + 1 -> slot X ^ 1
    //   step 3      barrier 1 (A half 0 of slot X read out; A half 1 of t, B and A half 0 of t + 1 landed)
    //   steps 3-6   DMA: A half 0 of K-step t + 2 -> slot X (one piece per step); B fragments of K-step t + 1 -> registers
    //   step 7      barrier 2 (A half 1 of slot X, B of slot X ^ 1 read out)
    auto step = [&](auto XT, auto JT) {
        constexpr int X = decltype(XT)::value, J = decltype(JT)::value;
        constexpr int h = J >> 2, ks = J & 3, cb = J & 1, nb = cb ^ 1;
        constexpr int NW0 = P::PB + 4;                       // pieces of steps 0-2: 10 or 12
        constexpr int n0 = (NW0 + 2) / 3, n1 = (NW0 - n0 + 1) / 2, n2 = NW0 - n0 - n1;  // 4,3,3 / 4,4,4
        constexpr int nd = J == 0 ? n0 : J == 1 ? n1 : J == 2 ? n2 : J <= 6 ? 1 : 0;
        constexpr int p0 = J == 0 ? 0 : J == 1 ? n0 : J == 2 ? n0 + n1 : 0;
        constexpr bool b_reads = J >= 3 && J <= 6;           // this step carries NT B reads (k-slice J - 3 of K-step t + 1)
        constexpr bool prev_b = J >= 4 && J <= 7;            // ... and so did the step before
        // Order of a step's LDS reads (they return in order): [A tile 0, A tile 1] of the next step, with the NT B reads
        // between them in steps 3-6 — behind MFMAs 0 .. NT + 1; the step's DMA pieces follow (steps 0-2: several).
        constexpr int lenp = prev_b ? NT + 2 : 2;            // reads the step before issued; its A tile 1 was the last one
        constexpr int mine = b_reads ? NT : 2;               // reads this step has issued before its MFMA NT
        if constexpr (J == 3) {  // barrier 1: A half 0 of this slot is read out (the two reads of step 2 were its last)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (P::PB == 6) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            bar();
            if constexpr (NPEND > 0) {
                if (pend_n < NPEND) {
                    emit_pending(pend_n);
                    ++pend_n;
                }
                ZG_SB();
            }
        }
        if constexpr (J == 7) {  // barrier 2: A half 1 of this slot and B of the other are read out (step 6 issued the last reads)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (P::PB == 6) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            bar();
        }
        // (Fragment reads run on across the end of a tile — the next tile's first K-step is in LDS by then; the tile loop
        // waits for them before the epilogue, see there.)
        auto gap = [&](auto MT) {  // the one item behind MFMA m
            constexpr int m = decltype(MT)::value;
            constexpr int JN = (J + 1) & 7, XN = (J == 7) ? (X ^ 1) : X;
            // where the next step's A tile 1 is read: last, behind the B reads — except in step 6, whose reads must ALL be
            // back before barrier 2 (they are the last ones of A half 1): the A tiles lead there
            constexpr int a1_gap = (b_reads && J != 6) ? NT + 1 : 1;
            constexpr int b_gap0 = J == 6 ? 2 : 1;           // first of the NT gaps that carry the B reads
            constexpr int d_gap0 = b_reads ? NT + 2 : 2;     // first gap that carries a DMA piece
            ZG_SB();
            if constexpr (m == 0) {
                if constexpr (J == 0) s2 = ahead(2);
                read_frag_a<NT, XN, JN, 0>(fa[nb][0], a_addr);
            } else if constexpr (m == a1_gap) {
                read_frag_a<NT, XN, JN, 1>(fa[nb][1], a_addr);
            } else if constexpr (b_reads && m >= b_gap0 && m < b_gap0 + NT) {
                read_frag_b<NT, X ^ 1, J - 3, m - b_gap0>(fb[X ^ 1][J - 3][m - b_gap0], b_addr);
            } else if constexpr (m >= d_gap0 && m - d_gap0 < nd) {
                constexpr int d = m - d_gap0;
                if constexpr (J <= 2) {
                    constexpr int p = p0 + d;
                    if constexpr (p < P::PB) dma_b(X, p, s2);         // B of K-step t + 2 -> this slot
                    else dma_a(X ^ 1, 1, p - P::PB, s1);              // A half 1 of K-step t + 1 -> other slot
                } else
                    dma_a(X, 0, J - 3, s2);                            // A half 0 of K-step t + 2 -> this slot
            }
            ZG_SB();
        };
        auto mma = [&](auto MT) {
            constexpr int m = decltype(MT)::value, i = m / NT, j = m % NT;
            // A operands: tile 0 was the FIRST read of the step before, tile 1 its LAST (steps 3 and 7 start behind lgkmcnt(0))
            if constexpr (J != 3 && J != 7) {
                if constexpr (m == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(lenp - 1) : "memory");
                if constexpr (m == NT) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(mine) : "memory");
                if constexpr (m == 0 || m == NT) ZG_SB();
            }
            mfma_acc<16 * ((h * 2 + i) * NT + j)>(fb[X][ks][j], fa[cb][i]);
            gap(MT);
        };
        mma(Ic<0>{});
        mma(Ic<1>{});
        mma(Ic<2>{});
        mma(Ic<3>{});
        mma(Ic<4>{});
        mma(Ic<5>{});
        if constexpr (NT == 4) {
            mma(Ic<6>{});
            mma(Ic<7>{});
        }
    };
    auto kstep = [&](auto XT) {
        step(XT, Ic<0>{});
        step(XT, Ic<1>{});
        step(XT, Ic<2>{});
        step(XT, Ic<3>{});
        step(XT, Ic<4>{});
        step(XT, Ic<5>{});
        step(XT, Ic<6>{});
        step(XT, Ic<7>{});
        s1 = s2;
    };

    // ---- epilogue of one tile, straight from the registers: lane (l31, hh) holds output row l31 of each 32 x 32
    // tile and columns 8 g + 4 hh + {0..3} (register 4 g + e).  bf16: v_permlane32_swap makes 8 consecutive
    // columns (16 B) per lane out of the two half-waves' runs of 4.
    // ---- SK: the producer's half tile leaves as raw accumulators, [shared tile][wave][48 register quads][lane] x 16 B, write-
    // through; then, its stores drained, one flag word per wave = this launch's epoch (the consumer's wave w polls wave w's flag
    // and loads exactly what that wave stored: no workgroup barrier on either side).  The poll is bounded: a consumer that
    // ran out raises g_s4_fault and carries on (the host checks the word where it drains the stream).
    auto sk_spill = [&]() {
        if constexpr (SK) {
            acc_settle();
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(qa.sk_ws, 0, (unsigned)(G >> 1) * 196608u, 0x00020000);
            const unsigned base = (unsigned)(sk_sh * 4 + wave) * 49152u + (unsigned)lane * 16u;
            static_for<12>([&](auto QT) {
                constexpr int q = decltype(QT)::value;
                float av[16];
                acc_read16<16 * q>(av);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4v v4 = {av[4 * e], av[4 * e + 1], av[4 * e + 2], av[4 * e + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), rw, base + (unsigned)(q * 4 + e) * 1024u, 0, 16);
                }
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (also the next item's DMA pieces in flight: they are due anyway)
            if (lane == 0) __hip_atomic_store(qa.sk_flags + sk_sh * 4 + wave, qa.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto sk_init_from_partial = [&]() {
        if constexpr (SK) {
            unsigned spins = 0;
            while (__hip_atomic_load(qa.sk_flags + sk_sh * 4 + wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != qa.sk_epoch) {
                if (++spins > (1u << 22)) {
                    if (lane == 0) g_s4_fault = 1u;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            // the partial's loads must not be hoisted above the flag: agent-scope acquire (the poll itself is relaxed)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(qa.sk_ws, 0, (unsigned)(G >> 1) * 196608u, 0x00020000);
            const unsigned base = (unsigned)(sk_sh * 4 + wave) * 49152u + (unsigned)lane * 16u;
            static_for<12>([&](auto QT) {
                constexpr int q = decltype(QT)::value;
                u32x4 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rw, base + (unsigned)(q * 4 + e) * 1024u, 0, 16);
                acc_write16<16 * q>(v);
            });
            asm volatile("s_nop 7" ::: ZG_ACC_CLOBBERS);  // v_accvgpr_write -> MFMA source C
        }
    };
    auto epilogue = [&]() {
        if (SK && role_cur == 1) {  // a producer's half tile: no epilogue, the accumulators go to its consumer
            sk_spill();
            if (has_nxt) init_acc_from_bias(tile_par ^ 1);
            ZG_SB();
            return;
        }
        acc_settle();
        if (dbg & 4) {  // diagnostic: no epilogue at all (the accumulators just keep running)
            return;
        }
        int lane_e = lane;  // opaque copy: nothing derived from it can be hoisted into (and kept across) the main loop
        asm volatile("" : "+v"(lane_e));
        const int l31e = lane_e & 31, hhe = lane_e >> 5;
        // Stores go through a buffer descriptor: a row past M lies past the descriptor's end and a column past N gets an
        // out-of-range offset, so the hardware drops them — no exec masking, one straight block of code per tile.
        flush_pending();  // (a tile of fewer K-steps than pending stores)
        const unsigned row0 = (unsigned)(m0 + wr * 128 + l31e) * (unsigned)(ldc * ESZ);
        const int col0 = n0 + wc * (P::BN / 2) + 8 * hhe;
        // bf16: a lane's runs of 4 columns go through a wave-private LDS image of the m-tile and leave as 16-B pieces of
        // whole rows (a store instruction that scatters 32-B pieces over 32 rows costs one L2 request per piece: the
        // epilogue was bounded by the request rate of the L2s, 6.8k cycles per tile)
        constexpr int CPR = NT * 4, NCH = 2 * NT;   // 16-B chunks per staged row; chunks (= stores) per lane per m-tile
        const unsigned st_w = lds_base + P::ST_OFF + wave * P::ST_WAVE + (unsigned)l31e * P::ST_ROW + (unsigned)hhe * 8u;
        unsigned st_r[NCH], g_off[NCH];
        if constexpr (OUT_BF16) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int c = lane_e + 64 * k, row = c / CPR, ch = c % CPR;
                st_r[k] = lds_base + P::ST_OFF + wave * P::ST_WAVE + (unsigned)row * P::ST_ROW + (unsigned)ch * 16u;
                const int col = n0 + wc * (P::BN / 2) + ch * 8;
                g_off[k] = (col < N ? (unsigned)(m0 + wr * 128 + row) * (unsigned)(ldc * 2) + (unsigned)col * 2u : 0xFFFFFFFFu) | no_store_mask;
            }
        }
        if constexpr (KIND == S4_SPLIT3) {
            // GELU, then the three planes of an m-tile one after the other through the wave's staging image (its LDS accesses
            // complete in order: plane p + 1 is written behind the reads of plane p without a wait)
            static_for<4>([&](auto IT) {
                constexpr int i = decltype(IT)::value;
                u32x2 w[3][NT][4];
                static_for<NT>([&](auto JT) {
                    constexpr int j = decltype(JT)::value;
                    asm volatile("s_nop 3" ::: "memory");
                    float av[16];
                    acc_read16<16 * (i * NT + j)>(av);
                    f32x2v x[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[k] = gelu2(f32x2v{av[2 * k], av[2 * k + 1]});
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint32_t a[3], b[3];
                        split3_pk(x[2 * g].x, x[2 * g].y, a[0], a[1], a[2]);
                        split3_pk(x[2 * g + 1].x, x[2 * g + 1].y, b[0], b[1], b[2]);
#pragma unroll
                        for (int p = 0; p < 3; ++p) w[p][j][g] = u32x2{a[p], b[p]};
                    }
                });
                static_for<3>([&](auto PT) {
                    constexpr int p = decltype(PT)::value;
                    static_for<NT>([&](auto JT) {
                        constexpr int j = decltype(JT)::value;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const unsigned st_w_l = st_w;
                            const u32x2 pk = w[p][j][g];
                            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(st_w_l), "v"(pk), "i"(j * 64 + g * 16) : "memory");
                        }
                    });
                    u32x4 o[NCH];
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        const unsigned st_r_l = st_r[k];
                        asm volatile("ds_read_b128 %0, %1" : "=v"(o[k]) : "v"(st_r_l) : "memory");
                    }
                    static_assert(NCH == 6, "staging read-back written for NT = 3");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]));
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {  // plane p of row m: columns [p N, (p + 1) N) of the 3 N-wide row
                        const unsigned off = g_off[k] == 0xFFFFFFFFu ? g_off[k] : g_off[k] + (unsigned)(i * 32) * (unsigned)(ldc * 2) + (unsigned)(p * N) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(o[k], rc, off, 0, ST_AUX);
                    }
                });
            });
        } else if constexpr (KIND == S4_PARTIAL || KIND == S4_QKV) {
            // fp32 rows of the whole-prompt Linears (N % 64 == 0), every 32 x 32 MFMA tile through the wave's staging image [32][32]
            // (pitch 144 B) so that a store instruction covers 8 rows x 128 contiguous bytes (straight from the registers it scatters
            // 32-byte pieces over 32 rows: one L2 request per piece, and the tile's epilogue took 65 k cycles with the cache append).
            // Write-through: the next kernel reads the rows from every XCD.  A row past M would land in the next slab, so rows are
            // masked here instead of by the descriptor's end.
            // S4_QKV: a 32-column tile is half of ONE head of q, k or v (E = 64 H).  q columns go to qkv[M][3E]; k / v go to the
            // head-major caches [b][h][ctx][64] (src/ops.zig:152-157) — and to qkv as well only when the cache is not fp32 (the
            // prompt attention reads an fp32 cache directly, prefill.hip).
            const unsigned sw = lds_base + P::ST_OFF + wave * P::ST_WAVE + (unsigned)l31e * 144u + (unsigned)hhe * 16u;
            const unsigned sr = lds_base + P::ST_OFF + wave * P::ST_WAVE + (unsigned)(lane_e >> 3) * 144u + (unsigned)(lane_e & 7) * 16u;
            [[maybe_unused]] float rcp_p = 0.0f;
            if constexpr (KIND == S4_QKV) rcp_p = 1.0f / (float)qa.P;
            static_for<4>([&](auto IT) {
                constexpr int i = decltype(IT)::value;
                const int mrow = m0 + wr * 128 + i * 32 + (lane_e >> 3);  // the lane's rows of this m-tile: mrow + 8 k
                [[maybe_unused]] int sb[4], st[4];  // S4_QKV: (sequence, position) of those rows
                if constexpr (KIND == S4_QKV) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = mrow + 8 * k;
                        int q = (int)((float)row * rcp_p), r = row - q * qa.P;  // row < 2^24: the estimate is off by at most one
                        if (r < 0) { --q; r += qa.P; }
                        if (r >= qa.P) { ++q; r -= qa.P; }
                        sb[k] = q;
                        st[k] = r;
                    }
                }
                static_for<NT>([&](auto JT) {
                    constexpr int j = decltype(JT)::value;
                    const int gcol = n0 + wc * (P::BN / 2) + j * 32;
                    asm volatile("s_nop 3" ::: "memory");
                    float av[16];
                    acc_read16<16 * (i * NT + j)>(av);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4v v4 = {av[4 * g], av[4 * g + 1], av[4 * g + 2], av[4 * g + 3]};
                        const unsigned sw_l = sw;
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sw_l), "v"(v4), "i"(g * 32) : "memory");
                    }
                    f32x4v o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned sr_l = sr;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o[k]) : "v"(sr_l), "i"(k * 8 * 144) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
                    const int col = gcol + 4 * (lane_e & 7);
                    bool rows_too = true;
                    [[maybe_unused]] void* cache = nullptr;
                    [[maybe_unused]] int e0 = 0;
                    if constexpr (KIND == S4_QKV) {
                        if (gcol >= qa.E) {
                            const int which = gcol >= 2 * qa.E;
                            e0 = col - (which ? 2 * qa.E : qa.E);
                            cache = which ? qa.v_cache : qa.k_cache;
                            rows_too = qa.kv_mode != 0;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = mrow + 8 * k;
                        const bool ok = row < M && gcol < N;
                        const unsigned off = (ok && rows_too) ? (unsigned)(sl * M + row) * (unsigned)(ldc * 4) + (unsigned)col * 4u : 0xFFFFFFFFu;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[k]), rc, off, 0, ST_AUX);
                        if constexpr (KIND == S4_QKV) {
                            if (cache != nullptr && ok)
                                kv_cache_store4(qa, cache, (((size_t)sb[k] * qa.H + (e0 >> 6)) * qa.ctx + st[k]) * 64 + (e0 & 63),
                                                f32x4{o[k][0], o[k][1], o[k][2], o[k][3]});
                        }
                    }
                });
            });
        } else
        static_for<4>([&](auto IT) {
            constexpr int i = decltype(IT)::value;
            static_for<NT>([&](auto JT) {
                constexpr int j = decltype(JT)::value;
                const int gcol = n0 + wc * (P::BN / 2) + j * 32;
                // (a 16-byte store reads its data registers a little after it issues; the compiler pads that hazard for its own
                // instructions only, and the moves below are ours: without the pad they overwrote the previous m-tile's store
                // data now and then)
                asm volatile("s_nop 3" ::: "memory");
                float av[16];
                acc_read16<16 * (i * NT + j)>(av);
                // eight pairs, every stage over all of them: independent chains side by side
                f32x2v x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = f32x2v{av[2 * k], av[2 * k + 1]};  // (the bias is in: the tile started there)
                if constexpr (GELU) {
                    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
                    f32x2v t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = x[k] * x[k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = __builtin_elementwise_fma(t[k], f32x2v{k2, k2}, f32x2v{k1, k1});
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = x[k] * t[k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = f32x2v{__builtin_amdgcn_exp2f(t[k].x), __builtin_amdgcn_exp2f(t[k].y)};
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = t[k] + f32x2v{1.0f, 1.0f};
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = f32x2v{__builtin_amdgcn_rcpf(t[k].x), __builtin_amdgcn_rcpf(t[k].y)};
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[k] = x[k] * t[k];
                }
                if constexpr (OUT_BF16) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {  // columns 32 j + 8 g + 4 hh + {0..3} of row l31: 8 bytes
                        const u32x2 pk = {cvt_pk_bf16(x[2 * g].x, x[2 * g].y), cvt_pk_bf16(x[2 * g + 1].x, x[2 * g + 1].y)};
                        const unsigned st_w_l = st_w;  // (a generic lambda does not capture what only an asm operand names)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(st_w_l), "v"(pk), "i"(j * 64 + g * 16) : "memory");
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = gcol + 8 * g + 4 * hhe;
                        const unsigned base_off = row0 + (unsigned)(i * 32) * (unsigned)(ldc * 4) + (unsigned)col * 4u;
                        const unsigned off = (col + 4 <= N ? base_off : 0xFFFFFFFFu) | no_store_mask;
                        const f32x4v o = {x[2 * g].x, x[2 * g].y, x[2 * g + 1].x, x[2 * g + 1].y};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rc, off, 0, 0);
                        if (N & 3) {  // a row that does not end on a run of four (lm_head: 50257 columns): its last 1-3 columns one by one
                            const bool tail = col < N && col + 4 > N;
                            const float oe[3] = {x[2 * g].x, x[2 * g].y, x[2 * g + 1].x};
#pragma unroll
                            for (int e = 0; e < 3; ++e) {
                                const unsigned offe = ((tail && col + e < N) ? base_off + 4u * e : 0xFFFFFFFFu) | no_store_mask;
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oe[e]), rc, offe, 0, 0);
                            }
                        }
                    }
                }
            });
            if constexpr (OUT_BF16) {  // the m-tile back out as rows (LDS serves a wave's accesses in order: no wait between write and read)
                u32x4 o[NCH];
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const unsigned st_r_l = st_r[k];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(o[k]) : "v"(st_r_l) : "memory");
                }
                if constexpr (NCH == 6)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]));
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const unsigned off = g_off[k] == 0xFFFFFFFFu ? g_off[k] : g_off[k] + (unsigned)(i * 32) * (unsigned)(ldc * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(o[k], rc, off, 0, ST_AUX);
                }
            }
        });
        if (has_nxt) {  // the next tile starts at its bias row — or, a consumer's half tile, at its producer's partial
            if (SK && role_nxt == 2) sk_init_from_partial();
            else init_acc_from_bias(tile_par ^ 1);
        }
        if constexpr (NPEND > 0) {
            pend_row = row0;
            pend_col = col0;
            pend_n = 0;
        }
        ZG_SB();
    };
    auto next_tile = [&]() {
        idx = nidx;
        ++it;
        m0 = ntm * 256;
        n0 = ntn * P::BN;
        sl = nsl;
        half_cur = half_nxt;
        role_cur = role_nxt;
        if constexpr (SK) kps = role_cur != 0 ? kpp >> 1 : kps_full;
        tile_par ^= 1;
        curA = nxtA;
        curB = nxtB;
        nxtA = kOob;
        nxtB = kOob;
        has_nxt = item(it + 1, nidx, half_nxt, role_nxt);
        if (has_nxt) {
            locate(nidx, ntm, ntn, nsl);
            nxtA = (unsigned)ntm * strideA + (unsigned)nsl * strideK + (unsigned)half_nxt * halfK;
            nxtB = (unsigned)ntn * strideB + (unsigned)nsl * strideK + (unsigned)half_nxt * halfK;
            fetch_bias(ntn * P::BN, tile_par ^ 1);  // a whole tile ahead of its use (its buffer was last read before this tile began)
        }
    };
    auto advance = [&]() {  // K-step t -> t + 1; true at the end of the tile
        if (++kk_cur == kps) {
            kk_cur = 0;
            ++pi_cur;
        }
        if (pi_cur < pl.npairs) return false;
        pi_cur = 0;
        return true;
    };

    // ---- prologue: the stream in steady-state order up to the start of K-step 0 — B, A half 0 of K-step 0 (the A half 1
    // of "K-step -1" is skipped), then B, A half 1 of K-step 0 ... wait: per K-step the stream carries
    // [B of t + 2, A half 1 of t + 1, A half 0 of t + 2]; before K-step 0 that is [B 0, A0 0] and [B 1, A1 0, A0 1].
    {
        fetch_bias(n0, 0);
        if (has_nxt) fetch_bias(ntn * P::BN, 1);
        const Ahead s0 = ahead(0);
        s1 = ahead(1);
        s2 = s1;
#pragma unroll
        for (int i = 0; i < P::PB; ++i) dma_b(0, i, s0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(0, 0, i, s0);
#pragma unroll
        for (int i = 0; i < P::PB; ++i) dma_b(1, i, s1);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(0, 1, i, s0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(1, 0, i, s1);
        if constexpr (KIND == S4_QKV) {  // the epilogue's argument-block fields, fetched under the first DMA
            ZG_PIN(qa.P); ZG_PIN(qa.E); ZG_PIN(qa.H); ZG_PIN(qa.ctx); ZG_PIN(qa.kv_mode); ZG_PIN(qa.kv_lo); ZG_PIN(qa.k_cache); ZG_PIN(qa.v_cache);
        }
        if constexpr (P::PB == 6) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // B and A half 0 of K-step 0 are in
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        bar();
        init_acc_from_bias(0);  // (wave 0's bias pieces were the first of the stream: landed with the wait above)
        read_kstep_head(Ic<0>{});
        phase_stamp();
    }

    // two K-steps per trip so that the slot is a compile-time constant; a tile may end after either
    for (;;) {
        kstep(Ic<0>{});
        if (advance()) {
            settle();
            phase_stamp();
            epilogue();
            phase_stamp();
            if (!has_nxt) break;
            next_tile();
            read_kstep_head(Ic<1>{});
            bar();  // every wave has re-read the new tile's first B fragments: steps 0-2 may now overwrite that B region
        }
        kstep(Ic<1>{});
        if (advance()) {
            settle();
            phase_stamp();
            epilogue();
            phase_stamp();
            if (!has_nxt) break;
            next_tile();
            read_kstep_head(Ic<0>{});
            bar();
        }
    }
    flush_pending();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // zero-fill pieces of "no next tile" still write this workgroup's LDS
    if (stamp && lane == 0) {
        g_s4_stamps[0] = gridDim.x;
        g_s4_stamps[1 + 2 * bid] = t_start;
        g_s4_stamps[2 + 2 * bid] = __builtin_readcyclecounter();
        g_s4_stamps[513 + 2 * bid] = w_start;
        g_s4_stamps[514 + 2 * bid] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int NT, int KIND, bool GELU, bool OUT_BF16, bool SK = false>
int launch_s4_kind(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc, int n_sl,
                  const PrefillQkv& qa, hipStream_t s) {
    using P = S4<NT>;
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s4_kernel<NT, KIND, GELU, OUT_BF16, SK>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS));
        raised = true;
    }
    const int tiles_m = (M + 255) / 256, tiles_n = (N + P::BN - 1) / P::BN, n_tiles = tiles_m * tiles_n * n_sl;
    int gw = 8;  // tile-order band width (sweep 1 .. 16: profiles/NOTEBOOK.md)
    if (gw > tiles_n) gw = tiles_n;
    const int cus_env = getenv("ZGPT2_GEMM_WGS") ? atoi(getenv("ZGPT2_GEMM_WGS")) : 0;  // tests: few workgroups, many tiles each
    const int cus = cus_env > 0 ? cus_env : 256;
    const int grid = n_tiles < cus ? n_tiles : cus;
    const unsigned dbg = (unsigned)(getenv("ZGPT2_GEMM_DBG") ? atoi(getenv("ZGPT2_GEMM_DBG")) : 0);
    ZG_REQUIRE(gemm_s4_args_ok(pl, ldc) && gw < 256 && grid < 1024 && dbg < (1u << 10) && n_sl >= 1 && n_sl < 256, ZG_ERR_UNSUPPORTED,
               "gemm: lda %d / ldb %d / ldc %d / K beyond the packed kernel arguments", pl.lda, pl.ldb, ldc);
    unsigned pa2 = 0, pb2 = 0;
    for (int i = 0; i < pl.npairs; ++i) {
        pa2 |= ((pl.pa_bits >> (4 * i)) & 3u) << (2 * i);
        pb2 |= ((pl.pb_bits >> (4 * i)) & 3u) << (2 * i);
    }
    hipLaunchKernelGGL((gemm_s4_kernel<NT, KIND, GELU, OUT_BF16, SK>), dim3(grid), dim3(256), P::LDS, s, A, B, bias, C, M, N,
                       (unsigned)pl.lda | ((unsigned)pl.ldb << 16), (unsigned)ldc | ((unsigned)pl.kpp << 20) | ((unsigned)pl.npairs << 28),
                       pa2 | (pb2 << 12) | ((unsigned)gw << 24), (unsigned)grid | (dbg << 10) | ((unsigned)n_sl << 20) | (pl.b_plane_major ? 0x80000000u : 0u), qa);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <int NT, bool GELU, bool OUT_BF16>
int launch_s4(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc, hipStream_t s) {
    const PrefillQkv none{};
    return launch_s4_kind<NT, S4_PLAIN, GELU, OUT_BF16>(A, B, bias, C, M, N, pl, ldc, 1, none, s);
}

template <int NT>
int launch_s4_nt(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc,
                 bool gelu, bool out_bf16, hipStream_t s) {
    if (gelu) return out_bf16 ? launch_s4<NT, true, true>(A, B, bias, C, M, N, pl, ldc, s)
                              : launch_s4<NT, true, false>(A, B, bias, C, M, N, pl, ldc, s);
    return out_bf16 ? launch_s4<NT, false, true>(A, B, bias, C, M, N, pl, ldc, s)
                    : launch_s4<NT, false, false>(A, B, bias, C, M, N, pl, ldc, s);
}

}  // namespace

// what the 14 packed argument dwords of gemm_s4_kernel can express (the dispatcher sends anything else that is not ragged
// to the eight-wave kernel, whose arguments are not packed; api_ops.hip keeps ragged Linears beyond it on the GEMV path)
bool gemm_s4_args_ok(const GemmPlanes& pl, int ldc) {
    return pl.lda > 0 && pl.ldb > 0 && pl.lda < 65536 && pl.ldb < 65536 && ldc < (1 << 20) && pl.kpp < 256 && pl.npairs <= 6;
}

int gemm_s4_fault(unsigned* out) {  // stream-K hand-over timed out since the last call?  (drains the device; clears the word)
    ZG_HIP(hipDeviceSynchronize());
    ZG_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s4_fault), sizeof(unsigned)));
    if (*out) {
        const unsigned zero = 0;
        ZG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_s4_fault), &zero, sizeof zero));
    }
    return ZG_OK;
}

int gemm_s4_stamps(unsigned long long* out, size_t n_words) {
    if (n_words > 1 + 4 * 256 + 16) n_words = 1 + 4 * 256 + 16;
    ZG_HIP(hipDeviceSynchronize());
    ZG_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s4_stamps), n_words * sizeof(unsigned long long)));
    return ZG_OK;
}

int launch_gemm_s4(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc,
                   bool gelu, bool out_bf16, int bn, hipStream_t s) {
    (void)bn;  // 192-wide tiles only: 256 x 256 needs all 256 accumulator registers plus 128 of B fragments
    return launch_s4_nt<3>(A, B, bias, C, M, N, pl, ldc, gelu, out_bf16, s);
}

// The whole-prompt Linears on the same kernel (prefill.hip decides when): A = the activation planes [M][nplanes K] (hi | mid | lo),
// W = the bf16 weight [N][K]; the planes are plane pairs of ONE K loop, smallest first — a tile's accumulators see 3 K / 64
// K-steps between two epilogues.  nplanes == kWeightPlanes: W = the three plane matrices [3][N][K] of an fp32 weight, and the six
// plane products a_i w_j, i + j <= 2, are the pairs (6 K / 64 K-steps per tile; what is dropped is below 2^-24 of the leading
// term).  kind: S4_PARTIAL (C = fp32 slabs [n_slices][M][N], bias must be null), S4_QKV (C = qkv [M][N] fp32 + cache append),
// S4_SPLIT3 (C = bf16 planes [M][3 N] of gelu(...)).
int launch_gemm_s4_prefill(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, int nplanes, int kind, int n_slices,
                           const PrefillQkv* qkv, hipStream_t s) {
    ZG_REQUIRE(M > 0 && N % 64 == 0 && K % 64 == 0 && K >= 128 && (nplanes == 2 || nplanes == 3 || nplanes == kWeightPlanes), ZG_ERR_UNSUPPORTED,
               "s4 prefill gemm: M=%d N=%d K=%d planes=%d", M, N, K, nplanes);
    GemmPlanes pl{};
    pl.lda = kSplit * K;  // the plane buffer always holds three planes per row; nplanes = 2 multiplies hi + mid only
    pl.ldb = K;
    pl.kpp = K / 64;
    if (nplanes == kWeightPlanes) {  // (a, w) = (lo, hi) (mid, mid) (hi, lo) (mid, hi) (hi, mid) (hi, hi): smallest terms first
        pl.npairs = 6;
        pl.pa_bits = 0x001012u;
        pl.pb_bits = 0x010210u;
        pl.b_plane_major = true;
        ZG_REQUIRE((size_t)N * K * 2 * 3 < ((size_t)1 << 31), ZG_ERR_SHAPE, "s4 prefill gemm: weight planes of %d x %d beyond a 32-bit buffer descriptor", N, K);
    } else {
        pl.npairs = nplanes;
        pl.pa_bits = nplanes == 3 ? 0x012u : 0x01u;  // pair 0 = the smallest plane
        pl.pb_bits = 0;
    }
    const int ldc = kind == S4_SPLIT3 ? kSplit * N : N;
    const size_t out_bytes = (size_t)(kind == S4_PARTIAL ? n_slices : 1) * M * ldc * (kind == S4_SPLIT3 ? 2 : 4);
    ZG_REQUIRE(out_bytes < ((size_t)1 << 32) && (size_t)M * pl.lda * 2 < ((size_t)1 << 31) && (size_t)N * K * 2 < ((size_t)1 << 31), ZG_ERR_SHAPE,
               "s4 prefill gemm: operands of %d x %d x %d beyond the 32-bit buffer descriptors", M, N, K);
    ZG_REQUIRE(n_slices >= 1 && pl.kpp % n_slices == 0 && pl.kpp / n_slices >= 2, ZG_ERR_ARG, "s4 prefill gemm: %d K slices of %d K-steps", n_slices, pl.kpp);
    const PrefillQkv none{};
    gemm_note_launch();
    switch (kind) {
        case S4_PARTIAL:
            ZG_REQUIRE(bias == nullptr, ZG_ERR_ARG, "s4 prefill gemm: partial slabs carry no bias");
            return launch_s4_kind<3, S4_PARTIAL, false, false>(A, W, nullptr, C, M, N, pl, ldc, n_slices, none, s);
        case S4_QKV: {
            ZG_REQUIRE(qkv && N == 3 * qkv->E && n_slices == 1, ZG_ERR_ARG, "s4 prefill gemm: S4_QKV needs the cache description");
            // 1.5 rounds of tiles (c_attn of eight 1023-token prompts: 384 tiles, 256 CUs): the last half round is split in K halves
            // over ALL workgroups with a partial hand-over instead of running on half of the chip (gemm_s4_kernel, SK)
            const int cus_env = getenv("ZGPT2_GEMM_WGS") ? atoi(getenv("ZGPT2_GEMM_WGS")) : 0;
            const int G = cus_env > 0 ? cus_env : 256, tiles = ((M + 255) / 256) * ((N + 191) / 192);
            if (qkv->sk_ws != nullptr && qkv->sk_flags != nullptr && G % 16 == 0 && tiles > G && (tiles % G) * 2 == G && pl.kpp % 2 == 0 &&
                pl.kpp >= 4 && (size_t)(G / 2) * 196608 <= qkv->sk_ws_bytes && (size_t)(G / 2) * 4 <= qkv->sk_flags_words)
                return launch_s4_kind<3, S4_QKV, false, false, true>(A, W, bias, C, M, N, pl, ldc, 1, *qkv, s);
            return launch_s4_kind<3, S4_QKV, false, false>(A, W, bias, C, M, N, pl, ldc, 1, *qkv, s);
        }
        case S4_SPLIT3:
            ZG_REQUIRE(n_slices == 1, ZG_ERR_ARG, "s4 prefill gemm: S4_SPLIT3 is not sliced");
            return launch_s4_kind<3, S4_SPLIT3, true, true>(A, W, bias, C, M, N, pl, ldc, 1, none, s);
    }
    ZG_REQUIRE(false, ZG_ERR_ARG, "s4 prefill gemm: kind %d", kind);
}

}  // namespace zg
