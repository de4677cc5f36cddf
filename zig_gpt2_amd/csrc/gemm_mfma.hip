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

}  // namespace

int launch_gemm_bf16_nt(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                        bool gelu, bool out_bf16, hipStream_t s) {
    ZG_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0, ZG_ERR_UNSUPPORTED,
               "gemm_bf16_nt: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, BM, BN, BK);
    ZG_REQUIRE(ldc >= N && ldc % 8 == 0, ZG_ERR_ARG, "gemm_bf16_nt: ldc %d", ldc);
    if (gelu) return out_bf16 ? launch_gemm_t<true, true>(A, B, bias, C, M, N, K, ldc, s)
                              : launch_gemm_t<true, false>(A, B, bias, C, M, N, K, ldc, s);
    return out_bf16 ? launch_gemm_t<false, true>(A, B, bias, C, M, N, K, ldc, s)
                    : launch_gemm_t<false, false>(A, B, bias, C, M, N, K, ldc, s);
}

}  // namespace zg
