// dma_intake.hip — how many bytes per clock one CU takes in through LDS-DMA (buffer_load ... lds) as a function of
// the bytes it keeps in flight, on the address pattern of the 768x3072 GEMM (M = 8192: 256 x 192 tiles, K-steps of
// 64 = 128-B row pieces, 2 tiles per workgroup, XCD-contiguous tile ranges), with and without MFMA / ds_read traffic
// beside it.  One workgroup per CU; every wave keeps D 1-KiB pieces outstanding (counted vmcnt).
//   NW   waves per workgroup (4 = one per SIMD, 8 = two)
//   D    pieces in flight per wave  -> NW * D KiB per CU
//   MF   32x32x16 bf16 MFMAs issued per piece (the real kernel: 3.43 per piece at 100 % of the matrix pipe)
//   RD   ds_read_b128 per piece
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int M = 8192, N = 3072, K = 768, BM = 256, BN = 192, KT = K / 64;
constexpr int TILES_M = M / BM, TILES_N = N / BN, PIECES = (BM + BN) / 8;  // 56 pieces of 8 rows x 128 B per K-step

__device__ __forceinline__ void tile_of(int idx, int gw, int& tm, int& tn) {
    const int band = idx / (TILES_M * gw);
    const int r = idx - band * TILES_M * gw;
    tm = r / gw;
    tn = band * gw + r % gw;
}

template <int NW, int D, int MF, int RD, bool VGPR>
__global__ __launch_bounds__(NW * 64, 1) void k(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                float* out, unsigned long long* cyc, int reps) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)lds;
    const int n_tiles = TILES_M * TILES_N, G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid % 8, loc = bid / 8, gx = G / 8;
    const int q8 = n_tiles / 8;
    const int t_begin = xcd * q8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)N * K * 2), 0x00020000);
    const int r8 = lane >> 3, c8 = (lane & 7) ^ ((r8 >> 1) & 7);
    const unsigned rel = (unsigned)(r8 * K + c8 * 8) * 2u;

    bf16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(float)((lane * 7 + j) % 13 - 6); fb[j] = (__bf16)(float)((lane * 5 + j) % 11 - 5); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    f32x4 vg[VGPR ? D : 1];

    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = __builtin_readcyclecounter();
    int cnt = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int ti = 0; ti < 2; ++ti) {
            int tm, tn;
            tile_of(t_begin + loc + ti * gx, 8, tm, tn);
            const unsigned baseA = (unsigned)tm * BM * K * 2u, baseB = (unsigned)tn * BN * K * 2u;
            for (int kt = 0; kt < KT; ++kt) {
                const unsigned kb = (unsigned)kt * 128u;
                const unsigned slot = lds_base + (unsigned)(cnt & 1) * (PIECES * 1024u);
#pragma unroll
                for (int pp = 0; pp < PIECES / NW; ++pp) {
                    const int p = pp * NW + wave;  // wave-uniform piece index
                    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(D - 1) : "memory");
                    const bool isA = p < BM / 8;
                    const unsigned off = rel + (isA ? baseA + (unsigned)p * 8u * K * 2u : baseB + (unsigned)(p - BM / 8) * 8u * K * 2u);
                    if constexpr (VGPR) {
                        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isA ? ra : rb, off, kb, 0));
                        asm volatile("" ::"v"(v));
                        (void)vg;
                    } else {
                        if (isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(size_t)(slot + p * 1024u), 16, off, kb, 0, 0);
                        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(size_t)(slot + p * 1024u), 16, off, kb, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < RD; ++r) {
                        f32x4 t;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(lds_base + (unsigned)((lane * 144 + pp * 4096) & 0xFFF0)), "i"(r * 16384));
                        asm volatile("" ::"v"(t));
                    }
#pragma unroll
                    for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();
                ++cnt;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    float t = sink[0];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int NW, int D, int MF, int RD, bool VGPR = false>
void run(const unsigned short* A, const unsigned short* B, float* out, unsigned long long* cyc) {
    auto fn = k<NW, D, MF, RD, VGPR>;
    const int lds = 2 * PIECES * 1024 + 16384;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int reps = 20;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fn, dim3(256), dim3(NW * 64), lds, 0, A, B, out, cyc, reps);
    CK(hipEventRecord(e0));
    const int L = 3;
    for (int r = 0; r < L; ++r) hipLaunchKernelGGL(fn, dim3(256), dim3(NW * 64), lds, 0, A, B, out, cyc, reps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double cs = 0; for (int i = 0; i < 256; ++i) cs += (double)h[i]; cs /= 256.0;
    const double ksteps = (double)reps * 2 * KT;
    const double us = ms * 1e3 / L;
    // s_memtime / readcyclecounter ticks at 100 MHz on gfx950: derive shader cycles from wall time at the nominal clock
    const double bytes_cu = ksteps * PIECES * 1024.0;
    printf("NW=%d D=%2d inflight=%3d KiB MF=%d RD=%d %s | %8.1f us | per K-step %6.3f us = %5.0f clk@2.4GHz | %5.1f B/clk/CU@2.4 | chip %5.2f TB/s | mfma-only %4.0f clk | ticks %.0f\n",
           NW, D, NW * D, MF, RD, VGPR ? "vgpr" : "lds ", us, us / ksteps, us / ksteps * 2400.0, bytes_cu / (us * 2400.0),
           bytes_cu * 256 / us / 1e6, (double)(PIECES / NW) * MF * 32.0, cs);
}

int main(int argc, char** argv) {
    unsigned short *A, *B; float* out; unsigned long long* cyc;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 256 * 8));
    // random bf16 bit patterns of moderate magnitude
    {
        size_t na = (size_t)M * K, nb = (size_t)N * K;
        unsigned short* h = (unsigned short*)malloc((na > nb ? na : nb) * 2);
        unsigned s = 12345u;
        for (size_t i = 0; i < na; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3c00u + ((s >> 16) & 0x3ffu) + ((s >> 3) & 0x8000u)); }
        CK(hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice));
        for (size_t i = 0; i < nb; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3c00u + ((s >> 16) & 0x3ffu) + ((s >> 3) & 0x8000u)); }
        CK(hipMemcpy(B, h, nb * 2, hipMemcpyHostToDevice));
        free(h);
    }
    printf("# LDS-DMA intake vs bytes in flight, GEMM 8192x3072x768 address pattern (256x192 tiles, 56 KiB per K-step)\n");
    run<4, 4, 0, 0>(A, B, out, cyc);
    run<4, 7, 0, 0>(A, B, out, cyc);
    run<4, 14, 0, 0>(A, B, out, cyc);
    run<4, 21, 0, 0>(A, B, out, cyc);
    run<4, 28, 0, 0>(A, B, out, cyc);
    run<4, 42, 0, 0>(A, B, out, cyc);
    run<8, 2, 0, 0>(A, B, out, cyc);
    run<8, 4, 0, 0>(A, B, out, cyc);
    run<8, 7, 0, 0>(A, B, out, cyc);
    run<8, 14, 0, 0>(A, B, out, cyc);
    run<8, 21, 0, 0>(A, B, out, cyc);
    printf("# with MFMAs (3 or 4 per piece; 3.43 = the GEMM at 100 %% of the pipe) and 2 ds_read_b128 per piece\n");
    run<4, 7, 3, 2>(A, B, out, cyc);
    run<4, 14, 3, 2>(A, B, out, cyc);
    run<4, 21, 3, 2>(A, B, out, cyc);
    run<4, 28, 3, 2>(A, B, out, cyc);
    run<4, 42, 3, 2>(A, B, out, cyc);
    run<4, 14, 4, 2>(A, B, out, cyc);
    run<4, 28, 4, 2>(A, B, out, cyc);
    run<4, 42, 4, 2>(A, B, out, cyc);
    run<8, 7, 3, 2>(A, B, out, cyc);
    run<8, 14, 3, 2>(A, B, out, cyc);
    run<8, 21, 3, 2>(A, B, out, cyc);
    run<8, 14, 4, 2>(A, B, out, cyc);
    run<4, 28, 3, 0>(A, B, out, cyc);
    run<4, 28, 0, 2>(A, B, out, cyc);
    printf("# MFMA only (no DMA): D irrelevant\n");
    printf("# plain 16-B loads to VGPRs instead of LDS-DMA\n");
    run<4, 14, 0, 0, true>(A, B, out, cyc);
    run<4, 28, 0, 0, true>(A, B, out, cyc);
    run<4, 28, 3, 0, true>(A, B, out, cyc);
    return 0;
}
