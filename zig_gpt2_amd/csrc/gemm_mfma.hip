// gemm_mfma.hip — the batched-regime Linear on the CDNA4 matrix cores:
//   C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (optionally GELU), bf16 operands, fp32 accumulate.
//
// This is Linear.forward (reference src/ops.zig:21-46: cblas_sgemm RowMajor/NoTrans/Trans) for
// M >> 1 — prompt prefill and the BASELINE "768x3072 GEMM" point (c_fc: K = 768, N = 3072).  Both
// operands are K-contiguous ("NT"), exactly the layouts ops.Linear already uses (x is [M, in], weight
// is [out, in]), so neither needs a transpose: an MFMA A fragment is 8 consecutive k of one row of x
// and a B fragment is 8 consecutive k of one row of W.
//
// Structure: 128 x 128 x 64 tile per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2
// v_mfma_f32_32x32x16_bf16 tiles, 64 accumulator VGPRs), two LDS stages of 32 KiB filled by
// global_load_lds_dwordx4 (16 B per lane straight into LDS, no VGPR round trip), XOR-swizzled so
// that the ds_read_b128 fragment reads are bank-conflict free: a tile row is 128 B = 8 chunks of
// 16 B and chunk c of row r is stored at position c ^ ((r >> 1) & 7).  Because the LDS-DMA writes
// lane-linearly, the swizzle is applied to the per-lane global SOURCE address and again on the
// fragment read (guide rule: both sides or neither).  Workgroups are renumbered so that the 8 XCDs
// (private L2s) each own a contiguous range of output tiles.  The epilogue stages the wave's
// 64 x 64 tile through LDS and writes full 128-B row segments with 16-B stores.
#include <stdlib.h>

#include "zg_kernels.h"

namespace zg {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;          // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;      // A + B
constexpr int kLdsBytes = 2 * kStageBytes;       // double buffered: 64 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Fill one 128 x 64 bf16 operand tile: 16 wave-instructions of 1 KiB, 4 per wave.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int ld, int row0, int k0, char* lds_tile,
                                           int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;               // 1-KiB piece = 8 tile rows
        const int row = piece * 8 + (lane >> 3);      // tile row this lane fills
        const int pos = lane & 7;                     // chunk position inside the LDS row
        const int chunk = pos ^ ((row >> 1) & 7);     // which source chunk belongs there
        const bf16_t* src = G + (size_t)(row0 + row) * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + off);
}

template <bool GELU, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_kernel(const bf16_t* __restrict__ A,
                                                              const bf16_t* __restrict__ B,
                                                              const float* __restrict__ bias, void* __restrict__ C,
                                                              int M, int N, int K, int ldc, int tiles_n,
                                                              int n_tiles) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware renumbering (bijective for any tile count): XCD x = bid % 8 owns a contiguous range.
    const int bid = blockIdx.x;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nt = K / BK;
    stage_tile(A, K, m0, 0, lds, wave, lane);
    stage_tile(B, K, n0, 0, lds + kTileBytes, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();

    const int frow = lane & 31, fk = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        char* cur = lds + (t & 1) * kStageBytes;
        if (t + 1 < nt) {
            char* nxt = lds + ((t + 1) & 1) * kStageBytes;
            stage_tile(A, K, m0, (t + 1) * BK, nxt, wave, lane);
            stage_tile(B, K, n0, (t + 1) * BK, nxt + kTileBytes, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = read_frag(cur, wm * 64 + i * 32 + frow, kk * 2 + fk);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = read_frag(cur + kTileBytes, wn * 64 + j * 32 + frow, kk * 2 + fk);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): next stage landed
        __syncthreads();
    }

    // ---- epilogue: bias (+ GELU), convert, stage the wave's 64 x 64 tile in LDS, 16-B row stores
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    char* wtile = lds + wave * (64 * 64 * ESZ);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = j * 32 + frow;
        const float bv = bias ? bias[n0 + wn * 64 + col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                float v = acc[i][j][r] + bv;
                if (GELU) v = gelu_ref(v);
                if (OUT_BF16) reinterpret_cast<bf16_t*>(wtile)[row * 64 + col] = f32_to_bf16_rne(v);
                else reinterpret_cast<float*>(wtile)[row * 64 + col] = v;
            }
    }
    // each wave re-reads only its own strip: wave-local ordering suffices
    constexpr int ROW_BYTES = 64 * ESZ, CHUNKS_PER_ROW = ROW_BYTES / 16, CHUNKS = 64 * CHUNKS_PER_ROW;
#pragma unroll
    for (int it = 0; it < CHUNKS / 64; ++it) {
        const int c = it * 64 + lane;
        const int row = c / CHUNKS_PER_ROW, cc = c % CHUNKS_PER_ROW;
        const u32x4 v = *reinterpret_cast<const u32x4*>(wtile + row * ROW_BYTES + cc * 16);
        char* dst = reinterpret_cast<char*>(C) + ((size_t)(m0 + wm * 64 + row) * ldc + n0 + wn * 64) * ESZ + cc * 16;
        *reinterpret_cast<u32x4*>(dst) = v;
    }
}

template <bool GELU, bool OUT_BF16>
int launch_gemm_t(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                  hipStream_t s) {
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_kernel<GELU, OUT_BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        raised = true;
    }
    const int tiles_m = M / BM, tiles_n = N / BN;
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<GELU, OUT_BF16>), dim3(tiles_m * tiles_n), dim3(256), kLdsBytes, s, A, B,
                       bias, C, M, N, K, ldc, tiles_n, tiles_m * tiles_n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}


// ------------------------------------------------------------------------------------------------
// Deep-pipelined variant: 256 x 128 x 64 tile, 8 waves (4 x 2, each 64 x 64), one workgroup per CU,
// a 3-slot LDS ring (3 x 48 KiB) filled two K-steps ahead by LDS-DMA.  The loop has ONE raw
// s_barrier per K-step and counted vmcnt (never 0 in steady state), so the DMA of stages t+1 / t+2
// stays in flight across the barrier while stage t is multiplied:
//     wait vmcnt(6)  -> this wave's pieces of stage t landed (stage t+1's 6 may still fly)
//     s_barrier      -> everybody's pieces landed AND everybody finished reading slot (t-1) % 3
//     issue stage t+2 into slot (t+2) % 3 == (t-1) % 3
//     ds_read + MFMA on slot t % 3
// All LDS is one array and no ordinary global load lives in the loop (hipcc otherwise drains the DMA
// queue with vmcnt(0)); bias is fetched in the epilogue.
constexpr int DM = 256, DN = 128;
constexpr int kDeepStage = (DM + DN) * BK * 2;  // 48 KiB
constexpr int kDeepLds = 3 * kDeepStage;        // 144 KiB

// rows: DM (A, 32 pieces) then DN (B, 16 pieces): 48 one-KiB pieces per stage, 6 per wave.
__device__ __forceinline__ void stage_deep(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int K, int m0,
                                           int n0, int k0, char* slot, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int piece = wave * 6 + q;                 // 0..47
        const bool isB = piece >= 32;
        const int prow = (isB ? piece - 32 : piece) * 8 + (lane >> 3);  // row inside its tile
        const int pos = lane & 7;
        const int chunk = pos ^ ((prow >> 1) & 7);
        const bf16_t* src = (isB ? B + (size_t)(n0 + prow) * K : A + (size_t)(m0 + prow) * K) + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(slot + piece * 1024), 16, 0, 0);
    }
}

// gelu(x) = x / (1 + exp(-2u)), u = x * 0.7978845608 * (1 + 0.044715 x^2)  (src/ops.zig:225), with the
// -2 log2(e) factor folded into the polynomial so that the exponential is a bare v_exp_f32.
__device__ __forceinline__ float gelu_fast(float x) {
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
    const float arg = x * fmaf(x * x, k2, k1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
}

// fp32 -> bf16 (round to nearest even) in one instruction (gfx950 v_cvt_pk_bf16_f32).
__device__ __forceinline__ bf16_t cvt_bf16(float x) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    return (bf16_t)r;
}

template <bool GELU, bool OUT_BF16>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_deep_kernel(const bf16_t* __restrict__ A,
                                                                   const bf16_t* __restrict__ B,
                                                                   const float* __restrict__ bias,
                                                                   void* __restrict__ C, int M, int N, int K, int ldc,
                                                                   int tiles_n, int n_tiles) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int bid = blockIdx.x;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * DM, n0 = tn * DN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nt = K / BK;
    stage_deep(A, B, K, m0, n0, 0, lds, wave, lane);
    if (nt > 1) stage_deep(A, B, K, m0, n0, BK, lds + kDeepStage, wave, lane);

    const int frow = lane & 31, fk = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) __builtin_amdgcn_s_waitcnt(0x0f76);  // vmcnt(6): stage t landed, t+1 may fly
        else __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) stage_deep(A, B, K, m0, n0, (t + 2) * BK, lds + ((t + 2) % 3) * kDeepStage, wave, lane);
        const char* cur = lds + (t % 3) * kDeepStage;
        const char* curB = cur + DM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = read_frag(cur, wm * 64 + i * 32 + frow, kk * 2 + fk);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = read_frag(curB, wn * 64 + j * 32 + frow, kk * 2 + fk);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring before it becomes the store staging area

    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    char* wtile = lds + wave * (64 * 64 * ESZ);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = j * 32 + frow;
        const float bv = bias ? bias[n0 + wn * 64 + col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                float v = acc[i][j][r] + bv;
                if (GELU) v = gelu_fast(v);
                if (OUT_BF16) reinterpret_cast<bf16_t*>(wtile)[row * 64 + col] = f32_to_bf16_rne(v);
                else reinterpret_cast<float*>(wtile)[row * 64 + col] = v;
            }
    }
    constexpr int ROW_BYTES = 64 * ESZ, CHUNKS_PER_ROW = ROW_BYTES / 16, CHUNKS = 64 * CHUNKS_PER_ROW;
#pragma unroll
    for (int it = 0; it < CHUNKS / 64; ++it) {
        const int c = it * 64 + lane;
        const int row = c / CHUNKS_PER_ROW, cc = c % CHUNKS_PER_ROW;
        const u32x4 v = *reinterpret_cast<const u32x4*>(wtile + row * ROW_BYTES + cc * 16);
        char* dst = reinterpret_cast<char*>(C) + ((size_t)(m0 + wm * 64 + row) * ldc + n0 + wn * 64) * ESZ + cc * 16;
        *reinterpret_cast<u32x4*>(dst) = v;
    }
}

template <bool GELU, bool OUT_BF16>
int launch_gemm_deep_t(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                       hipStream_t s) {
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_deep_kernel<GELU, OUT_BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kDeepLds));
        raised = true;
    }
    const int tiles_m = M / DM, tiles_n = N / DN;
    hipLaunchKernelGGL((gemm_nt_bf16_deep_kernel<GELU, OUT_BF16>), dim3(tiles_m * tiles_n), dim3(512), kDeepLds, s, A,
                       B, bias, C, M, N, K, ldc, tiles_n, tiles_m * tiles_n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}


// ------------------------------------------------------------------------------------------------
// 256 x 256 x 32 variant.  Measured on this chip the LDS-staged kernels are bound by the global->LDS
// fill rate per CU, so TFLOP/s scale with the tile's FLOP per staged byte: 128x128 = 65, 256x128 = 87,
// 256x256 = 131 FLOP/B.  8 waves as 2 (M) x 4 (N), each 128 x 64 = 4 x 2 MFMA tiles (128 accumulator
// VGPRs); a 4-slot ring of 32-KiB stages (BK = 32) filled three K-steps ahead; one raw s_barrier per
// K-step, counted vmcnt(8).  A tile row is 64 B = 4 chunks of 16 B; chunk c of row r sits at position
// c ^ ((r >> 2) & 3), which makes the 16-lane groups of ds_read_b128 conflict free.
constexpr int QM = 256, QN = 256, QK = 32;
constexpr int kQStage = (QM + QN) * QK * 2;  // 32 KiB
constexpr int kQSlots = 5;                   // ring slots: 4 stages in flight + the one being multiplied
constexpr int kQLds = kQSlots * kQStage;     // 160 KiB: the whole LDS of a CU

__device__ __forceinline__ void stage_q(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int K, int m0,
                                        int n0, int k0, char* slot, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;                  // 0..31: 16 A pieces (16 rows each) then 16 B pieces
        const bool isB = piece >= 16;
        const int prow = (isB ? piece - 16 : piece) * 16 + (lane >> 2);
        const int pos = lane & 3;
        const int chunk = pos ^ ((prow >> 2) & 3);
        const bf16_t* src = (isB ? B + (size_t)(n0 + prow) * K : A + (size_t)(m0 + prow) * K) + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(slot + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag_q(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}

template <bool GELU, bool OUT_BF16>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_q_kernel(const bf16_t* __restrict__ A,
                                                                const bf16_t* __restrict__ B,
                                                                const float* __restrict__ bias, void* __restrict__ C,
                                                                int M, int N, int K, int ldc, int tiles_n, int n_tiles,
                                                                int gw, int prio, int ablate) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int bid = blockIdx.x;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    // Column-grouped order: the XCD's contiguous tile range walks down M inside a band of `gw` N-tiles,
    // so the band's B panels stay resident in the XCD's 4-MiB L2 while A panels stream through once.
    int tm, tn;
    {
        const int tiles_m = n_tiles / tiles_n;
        const int band = tile / (tiles_m * gw);               // full bands first
        const int full = tiles_n / gw;
        if (band < full) {
            const int r = tile - band * tiles_m * gw;
            tm = r / gw;
            tn = band * gw + r % gw;
        } else {                                              // last, narrower band
            const int w = tiles_n - full * gw;
            const int r = tile - full * tiles_m * gw;
            tm = r / w;
            tn = full * gw + r % w;
        }
    }
    const int m0 = tm * QM, n0 = tn * QN;
    // diagnostic (ZGPT2_ABLATE=8): stage every tile from the same panels (all-L2-hit upper bound)
    const int ms = (ablate & 8) ? 0 : m0, ns = (ablate & 8) ? 0 : n0;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nt = K / QK;
#pragma unroll
    for (int st = 0; st < kQSlots - 1; ++st)
        if (st < nt) stage_q(A, B, K, ms, ns, st * QK, lds + st * kQStage, wave, lane);

    const int frow = lane & 31, fk = lane >> 5;
    // Role-split schedule.  The two waves that share a SIMD (w and w + 4) run half a stage out of phase:
    // in every barrier interval one of them only reads its fragments of a stage from LDS into registers
    // (12 ds_read_b128) while the other only issues that stage's 16 MFMAs, so the matrix pipe of each SIMD
    // is fed continuously by alternating waves instead of both waves loading and then both multiplying
    // (ablation: DMA, LDS reads and MFMA each take ~170-185 us of a 304 us lock-step run at 8192x4096x4096).
    //   interval i (between barriers i and i+1):  group g runs phase p = i - g;
    //   p even -> LOAD(stage p/2), p odd -> COMPUTE(stage (p-1)/2).
    // Stage s is waited for (counted vmcnt) before barrier 2s; the DMA of stage s+3 is issued after
    // barrier 2s into the slot whose last reader (group 1, interval 2s-1) has passed that barrier.
    const int grp = wave >> 2;
    bf16x8 fa[2][4], fb[2][2];
    auto load_stage = [&](int stage) {
        const char* cur = lds + (stage % kQSlots) * kQStage;
        const char* curB = cur + QM * QK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][i] = read_frag_q(cur, wm * 128 + i * 32 + frow, kk * 2 + fk);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[kk][j] = read_frag_q(curB, wn * 64 + j * 32 + frow, kk * 2 + fk);
        }
    };
    auto compute_stage = [&]() {
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        if (prio) __builtin_amdgcn_s_setprio(0);
    };
    // even barrier of stage s: wait for the stage, synchronise, refill the slot freed one stage ago
    auto even_barrier = [&](int st) {
        if (st < nt) {
            const int ahead = min(nt - 1 - st, kQSlots - 2);  // stages issued beyond st (4 loads each)
            if (ahead >= 3) __builtin_amdgcn_s_waitcnt(0x0f7c);       // vmcnt(12)
            else if (ahead == 2) __builtin_amdgcn_s_waitcnt(0x0f78);  // vmcnt(8)
            else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0f74);  // vmcnt(4)
            else __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0)
        }
        __builtin_amdgcn_s_barrier();
        const int nx = st + kQSlots - 1;  // its slot was last read one stage ago (group 1, interval 2 st - 1)
        if (nx < nt) stage_q(A, B, K, ms, ns, nx * QK, lds + (nx % kQSlots) * kQStage, wave, lane);
    };
    // Both groups execute exactly 2 * nt + 1 barriers.
    if (grp == 0) {
        for (int st = 0; st < nt; ++st) {
            even_barrier(st);                  // barrier 2*st
            load_stage(st);                    // interval 2*st
            __builtin_amdgcn_s_barrier();      // barrier 2*st + 1
            compute_stage();                   // interval 2*st + 1
        }
        even_barrier(nt);                      // barrier 2*nt (group 1 is still computing after it)
    } else {
        even_barrier(0);                       // barrier 0, idle interval 0
        for (int st = 0; st < nt; ++st) {
            __builtin_amdgcn_s_barrier();      // barrier 2*st + 1
            load_stage(st);                    // interval 2*st + 1
            even_barrier(st + 1);              // barrier 2*st + 2
            compute_stage();                   // interval 2*st + 2
        }
    }
    __builtin_amdgcn_s_barrier();  // ring no longer read: it becomes the store staging area

    // epilogue in two halves of 64 rows per wave (16 KiB of staging per wave even for fp32 output)
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    constexpr int ROW_BYTES = 64 * ESZ, CHUNKS_PER_ROW = ROW_BYTES / 16, CHUNKS = 64 * CHUNKS_PER_ROW;
    char* wtile = lds + wave * (64 * 64 * 4);
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = bias ? bias[n0 + wn * 64 + j * 32 + frow] : 0.0f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + frow;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    float v = acc[half * 2 + ii][j][r] + bv[j];
                    if (GELU) v = gelu_fast(v);
                    if (OUT_BF16) reinterpret_cast<bf16_t*>(wtile)[row * 64 + col] = f32_to_bf16_rne(v);
                    else reinterpret_cast<float*>(wtile)[row * 64 + col] = v;
                }
        }
#pragma unroll
        for (int it = 0; it < CHUNKS / 64; ++it) {
            const int c = it * 64 + lane;
            const int row = c / CHUNKS_PER_ROW, cc = c % CHUNKS_PER_ROW;
            const u32x4 v = *reinterpret_cast<const u32x4*>(wtile + row * ROW_BYTES + cc * 16);
            char* dst = reinterpret_cast<char*>(C) +
                        ((size_t)(m0 + wm * 128 + half * 64 + row) * ldc + n0 + wn * 64) * ESZ + cc * 16;
            *reinterpret_cast<u32x4*>(dst) = v;
        }
    }
}

template <bool GELU, bool OUT_BF16>
int launch_gemm_q_t(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                    hipStream_t s) {
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_q_kernel<GELU, OUT_BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kQLds));
        raised = true;
    }
    const int tiles_m = M / QM, tiles_n = N / QN;
    static const int gw_env = getenv("ZGPT2_GW") ? atoi(getenv("ZGPT2_GW")) : 0;
    int gw = gw_env > 0 ? gw_env : 6;
    if (gw > tiles_n) gw = tiles_n;
    hipLaunchKernelGGL((gemm_nt_bf16_q_kernel<GELU, OUT_BF16>), dim3(tiles_m * tiles_n), dim3(512), kQLds, s, A, B,
                       bias, C, M, N, K, ldc, tiles_n, tiles_m * tiles_n, gw, getenv("ZGPT2_PRIO") ? atoi(getenv("ZGPT2_PRIO")) : 1,
                       getenv("ZGPT2_ABLATE") ? atoi(getenv("ZGPT2_ABLATE")) : 0);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}


// ------------------------------------------------------------------------------------------------
// 256 x 256 tile, FULL-LINE staging.  The BK = 32 ring above asks L2 for every 128-B line twice (64 B per
// row per stage) and tops out at ~14 B/clk/CU of LDS-DMA; the BK = 64 kernels move ~19 B/clk/CU.  Here a
// ring unit is HALF the rows (128 of A + 128 of B) for 64 k: 32 pieces of 8 rows x 128 B, so each line is
// requested once.  K-step T (64 k) consumes units 2T (row half 0) and 2T + 1 (row half 1); 5 slots of
// 32 KiB; after barrier T the units 2T+3 and 2T+4 are issued into the slots step T-1 just released.
template <bool GELU, bool OUT_BF16, int MF>  // MF = 32: v_mfma_f32_32x32x16_bf16, 16: v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_f_kernel(const bf16_t* __restrict__ A,
                                                                const bf16_t* __restrict__ B,
                                                                const float* __restrict__ bias, void* __restrict__ C,
                                                                int M, int N, int K, int ldc, int tiles_n, int n_tiles,
                                                                int gw, int ablate) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int bid = blockIdx.x;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    int tm, tn;
    {
        const int tiles_m = n_tiles / tiles_n;
        const int band = tile / (tiles_m * gw), full = tiles_n / gw;
        if (band < full) {
            const int r = tile - band * tiles_m * gw;
            tm = r / gw;
            tn = band * gw + r % gw;
        } else {
            const int w = tiles_n - full * gw, r = tile - full * tiles_m * gw;
            tm = r / w;
            tn = full * gw + r % w;
        }
    }
    const int m0 = tm * QM, n0 = tn * QN;

    typedef __attribute__((ext_vector_type(4))) float f32x4v;
    f32x16 acc[4][2];     // MF == 32: 4 x 2 tiles of 32 x 32
    f32x4v acc16[8][4];   // MF == 16: 8 x 4 tiles of 16 x 16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};

    const int nt = K / 64, nu = 2 * nt;
    // unit u -> K-step u >> 1, row half u & 1; each wave issues 4 pieces (2 of A, 2 of B)
    auto issue_unit = [&](int u) {
        char* slot = lds + (u % 5) * 32768;
        const int k0 = (u >> 1) * 64, half = u & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;                 // 0..31: 16 A pieces then 16 B pieces, 8 rows each
            const bool isB = piece >= 16;
            const int prow = (isB ? piece - 16 : piece) * 8 + (lane >> 3);  // row inside the 128-row half
            const int pos = lane & 7;
            const int chunk = pos ^ ((prow >> 1) & 7);
            const bf16_t* src = (isB ? B + (size_t)(n0 + half * 128 + prow) * K : A + (size_t)(m0 + half * 128 + prow) * K) +
                                k0 + chunk * 8;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(slot + piece * 1024), 16, 0, 0);
        }
    };
    issue_unit(0);
    if (nu > 1) issue_unit(1);
    if (nu > 2) issue_unit(2);

    const int frow = lane & 31, fk = lane >> 5;
    for (int T = 0; T < nt; ++T) {
        if (T + 1 < nt) __builtin_amdgcn_s_waitcnt(0x0f74);  // vmcnt(4): units 2T, 2T+1 landed; 2T+2 may fly
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        if (2 * T + 3 < nu) issue_unit(2 * T + 3);
        if (2 * T + 4 < nu) issue_unit(2 * T + 4);
        const char* ua = lds + ((2 * T + wm) % 5) * 32768;                 // A half wm
        const char* ub = lds + ((2 * T + (wn >> 1)) % 5) * 32768 + 16384;  // B half wn >> 1
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MF == 16) {
            const int r16 = lane & 15, q16 = lane >> 4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {  // two 32-k substeps per unit pair
                bf16x8 a[8], b[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = read_frag(ua, i * 16 + r16, kk * 4 + q16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = read_frag(ub, (wn & 1) * 64 + j * 16 + r16, kk * 4 + q16);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc16[i][j], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = read_frag(ua, i * 32 + frow, kk * 2 + fk);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = read_frag(ub, (wn & 1) * 64 + j * 32 + frow, kk * 2 + fk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_s_barrier();  // ring no longer read: it becomes the store staging area
    if (ablate & 32) {  // diagnostic: no epilogue at all (keep the accumulators alive)
        float t = 0.0f;
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) t += acc16[i][j][0];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) t += acc[i][j][0];
        if (t == 1.2345f) reinterpret_cast<float*>(C)[0] = t;
        return;
    }

    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    constexpr int ROW_BYTES = 64 * ESZ, CHUNKS_PER_ROW = ROW_BYTES / 16, CHUNKS = 64 * CHUNKS_PER_ROW;
    char* wtile = lds + wave * (64 * 64 * 4);
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = bias ? bias[n0 + wn * 64 + j * 32 + frow] : 0.0f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + frow;
            if constexpr (MF == 32) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                        float v = acc[half * 2 + ii][j][r] + bv[j];
                        if (GELU) v = gelu_fast(v);
                        if (OUT_BF16) reinterpret_cast<bf16_t*>(wtile)[row * 64 + col] = cvt_bf16(v);
                        else reinterpret_cast<float*>(wtile)[row * 64 + col] = v;
                    }
            }
        }
        if constexpr (MF == 16) {  // D: col = lane & 15, row = 4 (lane >> 4) + r
            const int c16 = lane & 15, q16 = lane >> 4;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float b16 = bias ? bias[n0 + wn * 64 + jj * 16 + c16] : 0.0f;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = ii * 16 + q16 * 4 + r;
                        float v = acc16[half * 4 + ii][jj][r] + b16;
                        if (GELU) v = gelu_fast(v);
                        if (OUT_BF16) reinterpret_cast<bf16_t*>(wtile)[row * 64 + jj * 16 + c16] = cvt_bf16(v);
                        else reinterpret_cast<float*>(wtile)[row * 64 + jj * 16 + c16] = v;
                    }
            }
        }
#pragma unroll
        for (int it = 0; it < CHUNKS / 64; ++it) {
            const int c = it * 64 + lane;
            const int row = c / CHUNKS_PER_ROW, cc = c % CHUNKS_PER_ROW;
            const u32x4 v = *reinterpret_cast<const u32x4*>(wtile + row * ROW_BYTES + cc * 16);
            char* dst = reinterpret_cast<char*>(C) +
                        ((size_t)(m0 + wm * 128 + half * 64 + row) * ldc + n0 + wn * 64) * ESZ + cc * 16;
            if (!(ablate & 16)) *reinterpret_cast<u32x4*>(dst) = v;
            else if (v.x == 0x12345678u) *reinterpret_cast<u32x4*>(dst) = v;
        }
    }
}

template <bool GELU, bool OUT_BF16, int MF>
int launch_gemm_f_t(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                    hipStream_t s) {
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_f_kernel<GELU, OUT_BF16, MF>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768));
        raised = true;
    }
    const int tiles_m = M / QM, tiles_n = N / QN;
    static const int gw_env = getenv("ZGPT2_GW") ? atoi(getenv("ZGPT2_GW")) : 0;
    int gw = gw_env > 0 ? gw_env : 6;
    if (gw > tiles_n) gw = tiles_n;
    hipLaunchKernelGGL((gemm_nt_bf16_f_kernel<GELU, OUT_BF16, MF>), dim3(tiles_m * tiles_n), dim3(512), 5 * 32768, s, A, B,
                       bias, C, M, N, K, ldc, tiles_n, tiles_m * tiles_n, gw,
                       getenv("ZGPT2_ABLATE") ? atoi(getenv("ZGPT2_ABLATE")) : 0);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace

int launch_gemm_bf16_nt(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                        bool gelu, bool out_bf16, hipStream_t s) {
    ZG_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0, ZG_ERR_UNSUPPORTED,
               "gemm_bf16_nt: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, BM, BN, BK);
    ZG_REQUIRE(ldc >= N && ldc % 8 == 0, ZG_ERR_ARG, "gemm_bf16_nt: ldc %d", ldc);
    static const int variant = getenv("ZGPT2_GEMM") ? atoi(getenv("ZGPT2_GEMM")) : 0;  // 1: force 128x128
    if (M % QM == 0 && N % QN == 0 && K % 64 == 0 && (variant == 0 || variant == 5)) {
        if (gelu) return out_bf16 ? launch_gemm_f_t<true, true, 16>(A, B, bias, C, M, N, K, ldc, s)
                                  : launch_gemm_f_t<true, false, 16>(A, B, bias, C, M, N, K, ldc, s);
        return out_bf16 ? launch_gemm_f_t<false, true, 16>(A, B, bias, C, M, N, K, ldc, s)
                        : launch_gemm_f_t<false, false, 16>(A, B, bias, C, M, N, K, ldc, s);
    }
    if (M % QM == 0 && N % QN == 0 && K % 64 == 0 && variant == 4) {
        if (gelu) return out_bf16 ? launch_gemm_f_t<true, true, 32>(A, B, bias, C, M, N, K, ldc, s)
                                  : launch_gemm_f_t<true, false, 32>(A, B, bias, C, M, N, K, ldc, s);
        return out_bf16 ? launch_gemm_f_t<false, true, 32>(A, B, bias, C, M, N, K, ldc, s)
                        : launch_gemm_f_t<false, false, 32>(A, B, bias, C, M, N, K, ldc, s);
    }
    if (M % QM == 0 && N % QN == 0 && variant == 3) {
        if (gelu) return out_bf16 ? launch_gemm_q_t<true, true>(A, B, bias, C, M, N, K, ldc, s)
                                  : launch_gemm_q_t<true, false>(A, B, bias, C, M, N, K, ldc, s);
        return out_bf16 ? launch_gemm_q_t<false, true>(A, B, bias, C, M, N, K, ldc, s)
                        : launch_gemm_q_t<false, false>(A, B, bias, C, M, N, K, ldc, s);
    }
    if (M % DM == 0 && variant != 1) {
        if (gelu) return out_bf16 ? launch_gemm_deep_t<true, true>(A, B, bias, C, M, N, K, ldc, s)
                                  : launch_gemm_deep_t<true, false>(A, B, bias, C, M, N, K, ldc, s);
        return out_bf16 ? launch_gemm_deep_t<false, true>(A, B, bias, C, M, N, K, ldc, s)
                        : launch_gemm_deep_t<false, false>(A, B, bias, C, M, N, K, ldc, s);
    }
    if (gelu) return out_bf16 ? launch_gemm_t<true, true>(A, B, bias, C, M, N, K, ldc, s)
                              : launch_gemm_t<true, false>(A, B, bias, C, M, N, K, ldc, s);
    return out_bf16 ? launch_gemm_t<false, true>(A, B, bias, C, M, N, K, ldc, s)
                    : launch_gemm_t<false, false>(A, B, bias, C, M, N, K, ldc, s);
}

}  // namespace zg
