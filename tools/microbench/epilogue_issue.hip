// epilogue_issue.hip — what does the bias + GELU + bf16 epilogue of the 256 x 192 GEMM tile (gemm_s4.hip, the T2 benchmark's
// instantiation) cost when ONE wave per SIMD issues it — the persistent four-wave kernel's situation — against TWO waves per
// SIMD sharing the same tile (what an eight-wave kernel with 64 x 96 wave tiles would run)?  Same instruction mix as the
// kernel's epilogue block: per 16 accumulators of a lane 16 v_accvgpr_read, the packed GELU (3 v_pk_mul, v_pk_fma, v_pk_add
// per pair + 2 v_exp + 2 v_rcp), 8 v_cvt_pk_bf16_f32, 4 ds_write_b64; per 32-row m-tile 6 ds_read_b128 + 6 16-byte stores of
// whole 128-byte row pieces.  One workgroup per CU on all 256 CUs, stores to distinct tiles (HBM write traffic as in the GEMM).
//   W = 1: 4 waves, 192 accumulators each (4 m-tiles x 3 column tiles);  W = 2: 8 waves, 96 each (2 m-tiles x 3).
// Variants: G = GELU on / off, S = 0 arithmetic only (no LDS staging, no stores) / 1 everything.
// Prints cycles per tile epilogue (s_memtime, max over the workgroup's waves, averaged over iterations of workgroup 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2v;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned cvt_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v)); }

constexpr int kStRow = 3 * 64 + 16;  // staging image of one wave: 32 rows x 96 bf16 columns, rows padded by 16 B (as S4<3>::ST_ROW)

template <int W, bool G, bool S>
__global__ __launch_bounds__(256 * W, 1) void k(const float* __restrict__ src, unsigned short* __restrict__ out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int MT = 4 / W;  // m-tiles of 32 rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // the wave's accumulators, parked in AGPRs as the GEMM's are
    float acc[MT * 3 * 16];
#pragma unroll
    for (int i = 0; i < MT * 3 * 16; ++i) {
        const float v = src[(size_t)((blockIdx.x * 8 + wave) * 64 + lane) * 192 + i];
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[i]) : "v"(v));
    }
    const unsigned st_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)wave * 32u * kStRow;
    const unsigned st_w = st_base + (unsigned)l31 * kStRow + (unsigned)hh * 8u;
    unsigned st_r[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {  // chunk c of the read-back: row (lane >> 3) + 8 (c % 4) ... 12 x 16-byte pieces per row, 32 rows: 384 pieces / 64 lanes
        const int piece = c * 64 + lane, row = piece / 12, col16 = piece % 12;
        st_r[c] = st_base + (unsigned)row * kStRow + (unsigned)col16 * 16u;
    }
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7FFFFFFF, 0x00020000);
    // tile of this workgroup: rows 256 b .. +255 of a [65536][192] bf16 matrix; waves 2 x 2 over 128 x 96 quarters (W = 1) or
    // 4 x 2 over 64 x 96 eighths (W = 2), m-tiles of 32 rows each
    const unsigned tile_off = (unsigned)blockIdx.x * 256u * 384u;
    unsigned long long total = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float av[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(av[r]) : "a"(acc[(i * 3 + j) * 16 + r]));
                f32x2v x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = f32x2v{av[2 * q], av[2 * q + 1]};
                if (G) {
                    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
                    f32x2v t[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = x[q] * x[q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = __builtin_elementwise_fma(t[q], f32x2v{k2, k2}, f32x2v{k1, k1});
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = x[q] * t[q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = f32x2v{__builtin_amdgcn_exp2f(t[q].x), __builtin_amdgcn_exp2f(t[q].y)};
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = t[q] + f32x2v{1.0f, 1.0f};
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = f32x2v{__builtin_amdgcn_rcpf(t[q].x), __builtin_amdgcn_rcpf(t[q].y)};
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[q] = x[q] * t[q];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk = {cvt_pk(x[2 * g].x, x[2 * g].y), cvt_pk(x[2 * g + 1].x, x[2 * g + 1].y)};
                    if (S) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(st_w), "v"(pk), "i"(j * 64 + g * 16) : "memory");
                    else asm volatile("" ::"v"(pk));
                }
            }
            if (S) {
                u32x4 o[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) asm volatile("ds_read_b128 %0, %1" : "=v"(o[c]) : "v"(st_r[c]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]));
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int piece = c * 64 + lane, row = piece / 12, col16 = piece % 12;
                    const unsigned off = tile_off + (unsigned)(((wave >> 1) * MT + i) * 32 + row) * 384u + (unsigned)(wave & 1) * 192u + (unsigned)col16 * 16u;
                    __builtin_amdgcn_raw_buffer_store_b128(o[c], rc, off, 0, 0);
                }
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        __shared__ unsigned long long wmax;
        if (threadIdx.x == 0) wmax = 0;
        __syncthreads();
        if (lane == 0) atomicMax(&wmax, t1 - t0);
        __syncthreads();
        total += wmax;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = total;
}

template <int W, bool G, bool S>
double run(const float* src, unsigned short* out, unsigned long long* cyc, int grid) {
    const int iters = 200;
    const int lds = 4 * W * 32 * kStRow;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<W, G, S>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((k<W, G, S>), dim3(grid), dim3(256 * W), lds, 0, src, out, cyc, 20);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<W, G, S>), dim3(grid), dim3(256 * W), lds, 0, src, out, cyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256];
    CK(hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < grid; ++i) sum += (double)h[i];
    const double cycles = sum / grid / iters;
    printf("waves/SIMD %d  gelu %d  staging+stores %d  grid %3d : %8.0f cycles per tile epilogue   (%.2f us per epilogue by wall clock)\n", W, (int)G, (int)S, grid,
           cycles, ms * 1000.0 / iters);
    return cycles;
}

int main() {
    float* src;
    unsigned short* out;
    unsigned long long* cyc;
    const size_t n_src = (size_t)256 * 8 * 64 * 192;
    CK(hipMalloc(&src, n_src * 4));
    CK(hipMalloc(&out, (size_t)256 * 256 * 384));
    CK(hipMalloc(&cyc, 256 * 8));
    float* h = (float*)malloc(n_src * 4);
    unsigned s = 12345;
    for (size_t i = 0; i < n_src; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((int)(s >> 8) % 2001 - 1000) * 0.004f;
    }
    CK(hipMemcpy(src, h, n_src * 4, hipMemcpyHostToDevice));
    for (int grid : {256, 1}) {
        const double a = run<1, true, true>(src, out, cyc, grid);
        const double b = run<2, true, true>(src, out, cyc, grid);
        run<1, false, true>(src, out, cyc, grid);
        run<2, false, true>(src, out, cyc, grid);
        run<1, true, false>(src, out, cyc, grid);
        run<2, true, false>(src, out, cyc, grid);
        printf("  -> two waves per SIMD issue the same epilogue in %.2f of the time of one (grid %d)\n", b / a, grid);
    }
    return 0;
}
