// gemm_mfma.hip — the batched-regime Linear on the CDNA4 matrix cores:
//   C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (optionally GELU), bf16 operands, fp32 accumulate.
//
// This is Linear.forward (reference src/ops.zig:21-46: cblas_sgemm RowMajor/NoTrans/Trans) for
// M >> 1 — zg_linear_forward at M >= 16, the prompt prefill GEMMs and the BASELINE "768x3072 GEMM" point
// (c_fc: K = 768, N = 3072).  Both operands are K-contiguous ("NT"), exactly the layouts ops.Linear
// already uses (x is [M, in], weight is [out, in]), so neither needs a transpose.
//
// One kernel family, `gemm_p8_kernel<BN, ...>`:
//   * PERSISTENT: at most 256 workgroups (one per CU) walk the output tiles; tile shape 256 x BN with
//     BN in {256, 192} picked per problem so that the tile count divides over the CUs (M = 8192, N = 3072:
//     512 tiles of 256 x 192 = exactly two per CU instead of 1.5 tiles of 256 x 256).
//   * 8 waves, 512 threads, 16x16x32 bf16 MFMAs with the operands SWAPPED (A operand = weight rows, B
//     operand = activation rows), so a lane's four accumulator values are four consecutive output COLUMNS
//     of one row: the epilogue packs them into one 8-byte LDS write and the tile leaves as full-line rows.
//   * K-step of 64, two LDS buffers of four "half-tile units" (A rows of quadrant-half 0 / 1 of every
//     wave, B rows likewise), each unit filled by LDS-DMA (buffer_load ... lds, 16 B per lane, out-of-range
//     rows read as zero so M and N need not be tile multiples).  The image is XOR-swizzled on the SOURCE
//     side (the DMA writes lane-linearly) and on the fragment read: ds_read_b128 is conflict free.
//   * 4 phases per K-step (one accumulator quadrant each), every phase = {fragment reads + one unit of DMA,
//     s_barrier, MFMAs, s_barrier}; the two waves of a SIMD belong to two groups staggered by one barrier,
//     so one of them multiplies while the other loads.  DMA runs four units (a whole K-step) ahead under
//     COUNTED vmcnt waits; the stream of units continues across tile boundaries, i.e. the next tile's
//     first K-steps land while the current tile's epilogue runs.
//   * XCD-aware tile order: each XCD (private 4-MiB L2) owns a contiguous range of a column-banded order.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "zg_kernels.h"

namespace zg {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4v;

// gelu(x) = x / (1 + exp(-2u)), u = x * 0.7978845608 * (1 + 0.044715 x^2)  (src/ops.zig:225), with the
// -2 log2(e) factor folded into the polynomial so that the exponential is a bare v_exp_f32.
__device__ __forceinline__ float gelu_fast(float x) {
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
    const float arg = x * fmaf(x * x, k2, k1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
}

// Two elements at a time: the multiplies / FMAs become v_pk_*_f32 (the epilogue is VALU bound: ~1000 issue slots
// per wave per tile against ~25 free slots per phase in the main loop).
typedef __attribute__((ext_vector_type(2))) float f32x2v;
__device__ __forceinline__ f32x2v gelu_fast2(f32x2v x) {
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
    const f32x2v t = x * x;
    const f32x2v p = __builtin_elementwise_fma(t, f32x2v{k2, k2}, f32x2v{k1, k1});
    const f32x2v arg = x * p;
    f32x2v e;
    e.x = __builtin_amdgcn_exp2f(arg.x);
    e.y = __builtin_amdgcn_exp2f(arg.y);
    const f32x2v d = e + f32x2v{1.0f, 1.0f};
    f32x2v r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return x * r;
}

template <int BN_>
struct P8 {
    static constexpr int BM = 256, BN = BN_, BK = 64;
    static constexpr int WM = (BN == 256) ? 2 : 4, WN = 8 / WM;  // waves along M / N
    static constexpr int TM = BM / WM, TN = BN / WN;             // wave tile
    static constexpr int MT = TM / 16, NT = TN / 16;             // 16x16 MFMA tiles per wave
    static constexpr int QM = MT / 2, QN = NT / 2;               // per accumulator quadrant
    static constexpr int A_UNIT = 128 * 128;                     // bytes: 128 rows x 64 k bf16
    static constexpr int B_UNIT = (BN / 2) * 128;
    static constexpr int BUF = 2 * A_UNIT + 2 * B_UNIT;
    static constexpr int B_PIECES = B_UNIT / 1024;               // 1-KiB DMA pieces per B unit: 16 or 12
    static constexpr int ST_ROW = TN * 2 + 16;                   // padded staging row (bytes)
    static constexpr int ST_WAVE = 16 * ST_ROW;
    static constexpr int ST_OFF = 2 * BUF;
    static constexpr int LDS = ST_OFF + 8 * ST_WAVE;
    static constexpr int W = 4 + 2 * B_PIECES / 8;               // DMA instructions per wave per K-step: 8 or 7
    static __device__ __host__ constexpr int off_a(int h) { return h * A_UNIT; }
    static __device__ __host__ constexpr int off_b(int h) { return 2 * A_UNIT + h * B_UNIT; }
};

// NTILE fragments pairs (kk = 0 / 1) of one unit: tile i sits 16 rows = 2048 B further; hand-issued so that the
// reads stay where they are written (the compiler's own LDS loads may move across s_barrier)
template <int NTILE, int OFF>
__device__ __forceinline__ void read_frags(bf16x8 (&f)[NTILE * 2], unsigned addr0, unsigned addr1) {
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i * 2 + 0]) : "v"(addr0), "i"(OFF + i * 2048));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i * 2 + 1]) : "v"(addr1), "i"(OFF + i * 2048));
    }
}

// hand-issued 16-B load (scalar base + per-lane byte offset) and the matching counted wait that hands the
// destination registers back to the compiler
__device__ __forceinline__ void load_x4(f32x4v& dst, unsigned byte_off, const float* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base));
}
template <int CNT, int N>
__device__ __forceinline__ void wait_loads(f32x4v (&v)[N]) {
    static_assert(N == 4 || N == 6, "wait_loads: 4 or 6 destination vectors");
    static_assert(CNT == 7 || CNT == 8, "wait_loads: count");
    if constexpr (N == 4) {
        if constexpr (CNT == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        else asm volatile("s_waitcnt vmcnt(7)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    } else {
        if constexpr (CNT == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
        else asm volatile("s_waitcnt vmcnt(7)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    }
}

#define ZG_SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ lds_ptr_t to_lds(unsigned byte_addr) { return (lds_ptr_t)(size_t)byte_addr; }

// tile index in the column-banded order -> (tm, tn): bands of `gw` N-tiles, M-major inside a band, so an
// XCD's contiguous range is a block of (few M-tiles) x (gw N-tiles) whose panels stay in its L2.
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

template <int BN, bool GELU, bool OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         const float* __restrict__ bias, void* __restrict__ C, int M,
                                                         int N, GemmPlanes pl, int ldc, int tiles_m, int tiles_n, int gw, int dbg) {
    using P = P8<BN>;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // waves w and w + 4 share a SIMD: one of each group per SIMD
    const int wr = wave / P::WN, wc = wave % P::WN;
    const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)lds;

    // ---- this workgroup's tiles: XCD x = bid % 8 owns a contiguous range of the banded order
    const int n_tiles = tiles_m * tiles_n, G = gridDim.x, bid = blockIdx.x;
    const int nx = G < 8 ? G : 8;  // XCD groups that have workgroups
    const int xcd = bid % nx, loc = bid / nx;
    const int gx = G / nx + (xcd < G % nx ? 1 : 0);
    const int q8 = n_tiles / nx, r8 = n_tiles % nx;
    const int t_begin = xcd * q8 + min(xcd, r8), t_end = t_begin + q8 + (xcd < r8 ? 1 : 0);
    int idx = t_begin + loc;
    if (idx >= t_end) return;

    // ---- DMA source descriptors and per-lane offsets (constant for the whole kernel, relative to a tile)
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * pl.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)N * pl.ldb * 2), 0x00020000);
    // piece = 8 unit rows x 128 B; lane -> unit row r = piece * 8 + lane / 8, LDS position p = lane % 8 holds
    // source chunk p ^ ((r >> 1) & 7)
    auto rel_off = [&](int piece, int rows_per_wave_half, int tile_dim, int h, int ld) -> unsigned {
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int trow = (r / rows_per_wave_half) * tile_dim + h * rows_per_wave_half + r % rows_per_wave_half;
        return (unsigned)(trow * ld + c * 8) * 2u;
    };
    // Per operand ONE per-lane offset (piece `wave` of half 0); the wave's second piece and half 1 are whole rows
    // further (the swizzle term repeats every 16 rows), i.e. wave-uniform byte deltas that go into the scalar
    // tile base.  B-unit pieces per wave: BN = 256 -> {w, w + 8} for both halves; BN = 192 (12 pieces) ->
    // half 0: {w, w + 8 if w < 4}, half 1: {w, w + 4 if w >= 4}: every wave issues W = 7 pieces per K-step.
    const int pa1 = wave + 8;
    const int pb1[2] = {wave + 8, BN == 256 ? wave + 8 : wave + 4};
    const bool vb1[2] = {BN == 256 || wave < 4, BN == 256 || wave >= 4};
    auto tile_row = [&](int r, int half_rows, int tile_dim, int h) {  // unit row -> row of the tile
        return (r / half_rows) * tile_dim + h * half_rows + r % half_rows;
    };
    const unsigned relA = rel_off(wave, P::TM / 2, P::TM, 0, pl.lda), relB = rel_off(wave, P::TN / 2, P::TN, 0, pl.ldb);
    unsigned dA[2][2], dB[2][2];  // [half][piece slot] byte deltas (scalars)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ra0 = tile_row(wave * 8, P::TM / 2, P::TM, 0), rb0 = tile_row(wave * 8, P::TN / 2, P::TN, 0);
        dA[h][0] = (unsigned)(tile_row(wave * 8, P::TM / 2, P::TM, h) - ra0) * (unsigned)pl.lda * 2u;
        dA[h][1] = (unsigned)(tile_row(pa1 * 8, P::TM / 2, P::TM, h) - ra0) * (unsigned)pl.lda * 2u;
        dB[h][0] = (unsigned)(tile_row(wave * 8, P::TN / 2, P::TN, h) - rb0) * (unsigned)pl.ldb * 2u;
        dB[h][1] = (unsigned)(tile_row(pb1[h] * 8, P::TN / 2, P::TN, h) - rb0) * (unsigned)pl.ldb * 2u;
    }

    // ---- fragment read addresses: unit row = wr * TM/2 + i * 16 + (lane & 15), 16-B chunk kk * 4 + lane / 16
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned sw0 = (unsigned)((lq ^ ((lane >> 1) & 7)) << 4);
    const unsigned a_addr0 = lds_base + (wr * (P::TM / 2) + l15) * 128 + sw0, a_addr1 = a_addr0 ^ 64u;
    const unsigned b_addr0 = lds_base + (wc * (P::TN / 2) + l15) * 128 + sw0, b_addr1 = b_addr0 ^ 64u;

    // K-steps walk the plane pairs: step kt = pair * kpp + kk multiplies A plane pa[pair] with B plane pb[pair]
    const int kpp = pl.kpp, nt = pl.kpp * pl.npairs;
    int pi_cur = 0, kk_cur = 0;  // pair / step inside the pair of the current K-step t
    int tm, tn;
    tile_of(idx, tiles_m, tiles_n, gw, tm, tn);
    const unsigned strideA = 256u * (unsigned)pl.lda * 2u, strideB = (unsigned)BN * (unsigned)pl.ldb * 2u;
    unsigned curA = (unsigned)tm * strideA, curB = (unsigned)tn * strideB;
    int m0 = tm * 256, n0 = tn * BN;
    constexpr unsigned kOob = 0x80000000u;  // tile base of "no next tile": every lane out of range -> zero fill
    unsigned nxtA = kOob, nxtB = kOob;
    int nidx = idx + gx;
    int ntm = 0, ntn = 0;
    if (nidx < t_end) {
        tile_of(nidx, tiles_m, tiles_n, gw, ntm, ntn);
        nxtA = (unsigned)ntm * strideA;
        nxtB = (unsigned)ntn * strideB;
    }

    // unit of K-step (t + d) of the virtual stream (continues into the next tile) -> buffer X
    enum { U_A0 = 0, U_B0 = 1, U_B1 = 2, U_A1 = 3 };
    auto issue = [&](auto UT, auto XT, int t, int d) {
        constexpr int U = decltype(UT)::value, X = decltype(XT)::value;
        constexpr bool isA = (U == U_A0 || U == U_A1);
        constexpr int h = (U == U_A1 || U == U_B1) ? 1 : 0;
        (void)t;
        int kk = kk_cur + d, pi = pi_cur;  // d <= 2 <= kpp
        if (kk >= kpp) {
            kk -= kpp;
            ++pi;
        }
        const bool in_cur = pi < pl.npairs;
        if (!in_cur) pi = 0;
        const unsigned plane = ((isA ? pl.pa_bits : pl.pb_bits) >> (4 * pi)) & 15u;
        const unsigned kb = (plane * (unsigned)kpp + (unsigned)kk) * 128u;
        const unsigned base = isA ? (in_cur ? curA : nxtA) : (in_cur ? curB : nxtB);
        const unsigned dst = lds_base + X * P::BUF + (isA ? P::off_a(h) : P::off_b(h));
        if constexpr (isA) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, to_lds(dst + wave * 1024), 16, relA + (base + dA[h][0]), kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, to_lds(dst + pa1 * 1024), 16, relA + (base + dA[h][1]), kb, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, to_lds(dst + wave * 1024), 16, relB + (base + dB[h][0]), kb, 0, 0);
            if (vb1[h])
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, to_lds(dst + pb1[h] * 1024), 16, relB + (base + dB[h][1]), kb, 0, 0);
        }
    };
    auto wait_dma = [&]() {  // everything but the newest K-step's worth of this wave's DMA has landed
        if constexpr (P::W == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    };

    // The accumulators START at the bias, as the reference pre-fills the output rows with the bias and accumulates
    // with beta = 1 (src/ops.zig:24-29, :42): no bias add (and no zeroing) in the epilogue.
    f32x4v acc[P::MT][P::NT];
    auto load_bias = [&](f32x4v (&bv)[P::NT], int n0_) {
#pragma unroll
        for (int j = 0; j < P::NT; ++j) {
            const int col = n0_ + wc * P::TN + j * 16 + 4 * (lane >> 4);
            bv[j] = (bias != nullptr && col < N) ? *reinterpret_cast<const f32x4v*>(bias + col) : f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto init_acc = [&](const f32x4v (&bv)[P::NT]) {
#pragma unroll
        for (int i = 0; i < P::MT; ++i)
#pragma unroll
            for (int j = 0; j < P::NT; ++j) acc[i][j] = bv[j];
    };
    {
        f32x4v bv0[P::NT];
        load_bias(bv0, n0);
        init_acc(bv0);
    }
    bf16x8 fa[P::QM * 2], fb[2][P::QN * 2];  // [tile * 2 + kk]; the A set of half 1 replaces half 0 in phase 2

    auto read_a = [&](auto XT, auto QT) {
        constexpr int X = decltype(XT)::value, qa = decltype(QT)::value;
        read_frags<P::QM, P::off_a(qa)>(fa, a_addr0 + X * P::BUF, a_addr1 + X * P::BUF);
    };
    auto read_b = [&](auto XT, auto QT) {
        constexpr int X = decltype(XT)::value, qb = decltype(QT)::value;
        read_frags<P::QN, P::off_b(qb)>(fb[qb], b_addr0 + X * P::BUF, b_addr1 + X * P::BUF);
    };
    // accumulator quadrant (qa, qb) += A(qa) x B(qb) over the K-step; operands swapped: D[n][m]
    auto mma = [&](auto QA, auto QB) {
        constexpr int qa = decltype(QA)::value, qb = decltype(QB)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ZG_SB();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < P::QM; ++i)
#pragma unroll
                for (int j = 0; j < P::QN; ++j)
                    acc[qa * P::QM + i][qb * P::QN + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        fb[qb][j * 2 + kk], fa[i * 2 + kk], acc[qa * P::QM + i][qb * P::QN + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        ZG_SB();
    };
    auto bar = [&]() {
        ZG_SB();
        __builtin_amdgcn_s_barrier();
        ZG_SB();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // one K-step (index t of the current tile) out of buffer X; units of steps t + 1 / t + 2 are issued on the way
    // The next tile's bias (its accumulators start there) is fetched during the current tile's last K-step by
    // hand-issued loads: a compiler-tracked load consumed after the epilogue's stores would be waited for with
    // vmcnt(0), i.e. would drain the stores.  Loads retire in order among themselves, and exactly W DMA pieces
    // follow these, so the epilogue's leading vmcnt(W) covers them.
    f32x4v bvn[P::NT];
    auto prefetch_bias = [&]() {
        if (bias == nullptr) return;
        const int nn0 = (idx + gx < t_end) ? ntn * BN : n0;
#pragma unroll
        for (int j = 0; j < P::NT; ++j) {
            int col = nn0 + wc * P::TN + j * 16 + 4 * (lane >> 4);
            col = min(col, N - 4);  // columns past N are masked at the store; keep the address valid
            load_x4(bvn[j], (unsigned)col * 4u, bias);
        }
    };
    bool owe = false;  // group 1 owes the barrier that puts it one interval behind group 0 again (after a tile end)
    auto kstep = [&](auto XT, int t) {
        constexpr int X = decltype(XT)::value;
        using XO = std::integral_constant<int, X ^ 1>;
        // phase 0: quadrant (0, 0)
        read_b(XT, I0{});
        read_a(XT, I0{});
        if (t == nt - 1) prefetch_bias();  // exactly W DMA pieces are issued between here and the epilogue
        issue(I2{}, XO{}, t, 1);  // B half 1 of step t + 1
        wait_dma();               // B half 1 of step t landed (read in phase 1)
        bar();
        if (owe) {
            bar();
            owe = false;
        }
        mma(I0{}, I0{});
        bar();
        // phase 1: quadrant (0, 1)
        read_b(XT, I1{});
        issue(I3{}, XO{}, t, 1);  // A half 1 of step t + 1
        wait_dma();               // A half 1 of step t landed (read in phase 2)
        bar();
        mma(I0{}, I1{});
        bar();
        // phase 2: quadrant (1, 1)
        read_a(XT, I1{});
        issue(I0{}, XT, t, 2);    // A half 0 of step t + 2 (this buffer: last read two phases ago)
        bar();
        mma(I1{}, I1{});
        bar();
        // phase 3: quadrant (1, 0), no fragment reads
        issue(I1{}, XT, t, 2);    // B half 0 of step t + 2
        wait_dma();               // A half 0 / B half 0 of step t + 1 landed (read in the next phase 0)
        bar();
        mma(I1{}, I0{});
        // Tile end: group 1 goes straight from its last MFMAs into its epilogue (no barrier), so that the two
        // groups' epilogues — VALU bound, with a store tail — run side by side instead of one after the other.
        if (grp == 1 && t == nt - 1) owe = true;
        else bar();
    };

    // ---- epilogue of one tile: bias (+ GELU), convert, wave-private LDS staging, full-row 16-B stores
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    constexpr int NPASS = OUT_BF16 ? 1 : 2;       // fp32 output goes in two column halves through the same staging rows
    constexpr int NTP = P::NT / NPASS;            // n-tiles per pass
    constexpr int CPR = P::TN * 2 / 16;           // 16-B chunks per staged row
    auto epilogue = [&]() {
        if (dbg & 4) {  // diagnostic: no epilogue at all (accumulators kept alive through an impossible store)
            float tsum = 0.0f;
#pragma unroll
            for (int i = 0; i < P::MT; ++i)
#pragma unroll
                for (int j = 0; j < P::NT; ++j) {
                    tsum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
                    acc[i][j] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};  // (diagnostic path: bias dropped)
                }
            if (tsum == 1.2345e33f) reinterpret_cast<float*>(C)[0] = tsum;
            return;
        }
        int lane_e = lane;  // opaque copy: nothing derived from it can be hoisted into (and pinned across) the main loop
        asm volatile("" : "+v"(lane_e));
        const int l15 = lane_e & 15, lq = lane_e >> 4;
        char* st = lds + P::ST_OFF + wave * P::ST_WAVE;
        if (bias != nullptr) wait_loads<P::W>(bvn);
        else {
#pragma unroll
            for (int j = 0; j < P::NT; ++j) bvn[j] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int i = 0; i < P::MT; ++i) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
                for (int jj = 0; jj < NTP; ++jj) {
                    const int j = ps * NTP + jj;
                    f32x4v v = acc[i][j];
                    if (GELU) {
                        const f32x2v g0 = gelu_fast2(f32x2v{v[0], v[1]}), g1 = gelu_fast2(f32x2v{v[2], v[3]});
                        v = f32x4v{g0.x, g0.y, g1.x, g1.y};
                    }
                    if (OUT_BF16) {
                        u32x2 pk;
                        pk.x = cvt_pk_bf16(v[0], v[1]);
                        pk.y = cvt_pk_bf16(v[2], v[3]);
                        *reinterpret_cast<u32x2*>(st + l15 * P::ST_ROW + (jj * 16 + 4 * lq) * 2) = pk;
                    } else {
                        *reinterpret_cast<f32x4v*>(st + l15 * P::ST_ROW + (jj * 16 + 4 * lq) * 4) = v;
                    }
                }
                // 16 rows x (TN * 2) bytes back out as whole rows
#pragma unroll
                for (int it = 0; it < (16 * CPR) / 64; ++it) {
                    const int c = it * 64 + lane_e, row = c / CPR, ch = c % CPR;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * P::ST_ROW + ch * 16);
                    const int grow = m0 + wr * P::TM + i * 16 + row;
                    const int gcol = n0 + wc * P::TN + ps * (P::TN / NPASS) + ch * (16 / ESZ);
                    if (grow < M && gcol < N && (!(dbg & 1) || v.x == 0x12345678u))
                    {
                        u32x4* dstp = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(C) + ((size_t)grow * ldc + gcol) * ESZ);
                        if (dbg & 16) __builtin_nontemporal_store(v, dstp);  // A/B: measured ~2 us slower at M = 8192
                        else *dstp = v;
                    }
                }
            }
        }
        init_acc(bvn);
        // The stores share the vmcnt counter with the DMA and may retire out of order with respect to it.  The
        // counted waits stay SAFE without a drain: the DMA pieces retire in order among themselves, so "at most W
        // operations outstanding" still implies "at most the W newest DMA pieces outstanding" — stores still in
        // flight only make a wait conservative.  (dbg bit 8: drain anyway, for A/B timing.)
        if (dbg & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ZG_SB();
    };
    auto next_tile = [&]() {
        idx = nidx;
        m0 = ntm * 256;
        n0 = ntn * BN;
        curA = nxtA;
        curB = nxtB;
        nidx = idx + gx;
        nxtA = kOob;
        nxtB = kOob;
        if (nidx < t_end) {
            tile_of(nidx, tiles_m, tiles_n, gw, ntm, ntn);
            nxtA = (unsigned)ntm * strideA;
            nxtB = (unsigned)ntn * strideB;
        }
    };
    auto advance = [&]() {  // K-step t -> t + 1; true at the end of the tile
        if (++kk_cur == kpp) {
            kk_cur = 0;
            ++pi_cur;
        }
        if (pi_cur < pl.npairs) return false;
        pi_cur = 0;
        return true;
    };

    // ---- prologue: K-step 0 complete + the first two units of K-step 1, as the steady state expects
    issue(I0{}, I0{}, 0, 0);
    issue(I1{}, I0{}, 0, 0);
    issue(I2{}, I0{}, 0, 0);
    issue(I3{}, I0{}, 0, 0);
    issue(I0{}, I1{}, 0, 1);
    issue(I1{}, I1{}, 0, 1);
    wait_dma();  // A half 0 / B half 0 of step 0
    bar();
    if (grp == 1) bar();  // stagger: group 1 runs one barrier behind group 0

    // two K-steps per trip so that the buffer is a compile-time constant; a tile may end after either
    int t = 0;
    for (;;) {
        kstep(I0{}, t);
        ++t;
        if (advance()) {
            epilogue();
            if (idx + gx >= t_end) break;
            next_tile();
            t = 0;
        }
        kstep(I1{}, t);
        ++t;
        if (advance()) {
            epilogue();
            if (idx + gx >= t_end) break;
            next_tile();
            t = 0;
        }
    }
    bar();  // group 0: the barrier group 1 ran ahead by at start-up; group 1: the one it still owes
}

template <int BN, bool GELU, bool OUT_BF16>
int launch_p8(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc, hipStream_t s) {
    using P = P8<BN>;
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_kernel<BN, GELU, OUT_BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS));
        raised = true;
    }
    const int tiles_m = (M + 255) / 256, tiles_n = (N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
    int gw = 8;  // tile-order band width (sweep 1 .. 16: profiles/NOTEBOOK.md)
    if (gw > tiles_n) gw = tiles_n;
    const int cus_env = getenv("ZGPT2_GEMM_WGS") ? atoi(getenv("ZGPT2_GEMM_WGS")) : 0;  // tests: few workgroups, many tiles each
    const int cus = cus_env > 0 ? cus_env : 256;
    const int grid = n_tiles < cus ? n_tiles : cus;
    hipLaunchKernelGGL((gemm_p8_kernel<BN, GELU, OUT_BF16>), dim3(grid), dim3(512), P::LDS, s, A, B, bias, C, M, N, pl,
                       ldc, tiles_m, tiles_n, gw, getenv("ZGPT2_GEMM_DBG") ? atoi(getenv("ZGPT2_GEMM_DBG")) : 0);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <int BN>
int launch_p8_bn(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc,
                 bool gelu, bool out_bf16, hipStream_t s) {
    if (gelu) return out_bf16 ? launch_p8<BN, true, true>(A, B, bias, C, M, N, pl, ldc, s)
                              : launch_p8<BN, true, false>(A, B, bias, C, M, N, pl, ldc, s);
    return out_bf16 ? launch_p8<BN, false, true>(A, B, bias, C, M, N, pl, ldc, s)
                    : launch_p8<BN, false, false>(A, B, bias, C, M, N, pl, ldc, s);
}

}  // namespace

static unsigned long long g_gemm_launches = 0;
unsigned long long gemm_mfma_launch_count() { return g_gemm_launches; }
void gemm_note_launch() { ++g_gemm_launches; }
int gemm_debug_stamps(unsigned long long* out, size_t n_words) { return gemm_s4_stamps(out, n_words); }

int launch_gemm_planes(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl,
                       int ldc, bool gelu, bool out_bf16, hipStream_t s) {
    const int K = pl.kpp * 64;
    ZG_REQUIRE(M > 0 && N > 0 && pl.kpp >= 2 && pl.npairs >= 1 && pl.npairs <= 8, ZG_ERR_UNSUPPORTED,
               "gemm: M=%d N=%d K=%d pairs=%d: K must be a multiple of 64 and at least 128", M, N, K, pl.npairs);
    ZG_REQUIRE(pl.lda % 8 == 0 && pl.ldb % 8 == 0 && pl.lda >= K && pl.ldb >= K, ZG_ERR_UNSUPPORTED, "gemm: lda=%d ldb=%d", pl.lda, pl.ldb);
    // bf16 output leaves in 16-byte pieces of 8 columns; fp32 output may have any width and row length (the four-wave kernel
    // stores a ragged row end dword by dword: Linear 1600 -> 50257 at batch >= 16)
    const bool ragged = !out_bf16 && (N % 4 != 0 || ldc % 4 != 0);
    ZG_REQUIRE(ldc >= N && (!out_bf16 || (N % 8 == 0 && ldc % 8 == 0)), ZG_ERR_UNSUPPORTED, "gemm: N=%d ldc=%d (bf16 output: multiples of 8)", N, ldc);
    ZG_REQUIRE((size_t)M * ldc * (out_bf16 ? 2 : 4) < ((size_t)1 << 32), ZG_ERR_SHAPE, "gemm: output over 4 GiB");
    ZG_REQUIRE((size_t)N * pl.ldb < (1u << 30), ZG_ERR_SHAPE, "gemm: B operand over 2 GiB");
    if ((size_t)M * pl.lda >= (1u << 30)) {
        // The A operand is addressed through a 32-bit buffer descriptor: a taller A goes in row chunks (whole tiles,
        // under 2 GiB each) — e.g. the mlp c_proj of a long fp32-weight prefill, whose plane rows are 12 E wide.
        const int rows = (int)((((size_t)1 << 30) - 1) / (size_t)pl.lda) / 256 * 256;
        ZG_REQUIRE(rows >= 256, ZG_ERR_SHAPE, "gemm: lda=%d too wide", pl.lda);
        const size_t esz = out_bf16 ? 2 : 4;
        for (int r0 = 0; r0 < M; r0 += rows) {
            const int mc = M - r0 < rows ? M - r0 : rows;
            ZG_TRY(launch_gemm_planes(A + (size_t)r0 * pl.lda, B, bias, reinterpret_cast<char*>(C) + (size_t)r0 * ldc * esz, mc, N, pl, ldc,
                                      gelu, out_bf16, s));
        }
        return ZG_OK;
    }
    // tile width: the one that wastes fewer CU-rounds (M = 8192, N = 3072: 512 tiles of 256 x 192 = 2.0 per CU
    // against 384 tiles of 256 x 256 = two rounds with half the chip idle in the second)
    const char* kk = getenv("ZGPT2_GEMM_KERNEL");  // test / measurement hook: s4 | p8 | p8:192 | p8:256
    const int bn_env = (kk && !strcmp(kk, "p8:192")) ? 192 : (kk && !strcmp(kk, "p8:256")) ? 256 : 0;
    auto cost = [&](int bn) {
        const long tiles = (long)((M + 255) / 256) * ((N + bn - 1) / bn);
        return ((tiles + 255) / 256) * bn;
    };
    // (ties and near-ties go to the 192-wide tiles: they run on the four-wave kernel, which is the faster generation —
    // M = 16384: 76 us against 88.7 us on the eight-wave kernel's 256-wide tiles at equal cost)
    int bn = cost(192) * 8 <= cost(256) * 9 ? 192 : 256;
    if (bn_env == 192 || bn_env == 256) bn = bn_env;
    ++g_gemm_launches;
    // Two generations of the kernel: the four-wave software-pipelined one (gemm_s4.hip) wherever 192-wide tiles are the
    // choice (its 256-wide instantiation does not fit the register file without spills yet), the eight-wave one below
    // for 256-wide tiles.  ZGPT2_GEMM_KERNEL=p8 / s4 forces one (s4 then always with 192-wide tiles).  (The third generation —
    // a tile's epilogue under the next tile's main loop, bitwise equal and slower — lives in tools/experiments/gemm_ov.hip.)
    const bool force_s4 = kk && !strcmp(kk, "s4"), force_p8 = kk && !strncmp(kk, "p8", 2);
    // The four-wave kernel packs its arguments (gemm_s4_args_ok: lda / ldb < 65536, K < 16384 per plane, <= 6 plane pairs);
    // a shape beyond that runs on the eight-wave kernel — unless it is ragged, which only the four-wave kernel stores.
    const bool s4_ok = gemm_s4_args_ok(pl, ldc);
    ZG_REQUIRE(s4_ok || !ragged, ZG_ERR_UNSUPPORTED, "gemm: a ragged output (N=%d, ldc=%d) with lda %d / ldb %d / K %d beyond the four-wave kernel", N,
               ldc, pl.lda, pl.ldb, K);
    if (s4_ok && (force_s4 || ragged || (!force_p8 && bn == 192))) return launch_gemm_s4(A, B, bias, C, M, N, pl, ldc, gelu, out_bf16, 192, s);
    return bn == 192 ? launch_p8_bn<192>(A, B, bias, C, M, N, pl, ldc, gelu, out_bf16, s)
                     : launch_p8_bn<256>(A, B, bias, C, M, N, pl, ldc, gelu, out_bf16, s);
}

int launch_gemm_bf16_nt(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                        bool gelu, bool out_bf16, hipStream_t s) {
    ZG_REQUIRE(K >= 128 && K % 64 == 0, ZG_ERR_UNSUPPORTED, "gemm_bf16_nt: K=%d must be a multiple of 64 and at least 128", K);
    GemmPlanes pl{};
    pl.lda = K;
    pl.ldb = K;
    pl.kpp = K / 64;
    pl.npairs = 1;
    return launch_gemm_planes(A, B, bias, C, M, N, pl, ldc, gelu, out_bf16, s);
}

}  // namespace zg
