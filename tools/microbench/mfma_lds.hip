// mfma_lds.hip — what a ds_read_b128 costs beside v_mfma_f32_32x32x16_bf16 when ONE wave per SIMD runs both
// (4 waves per CU, 512 registers each), by what the MFMAs read and where the reads are placed.
//   V = 0  MFMA sources are constant registers, reads go to scratch registers
//   V = 1  MFMA sources are the registers the reads of the previous step wrote (two buffers), lgkmcnt(0) per step
//   V = 2  as 1 without any wait (wrong data, same instruction stream)
//   P = 0  reads clumped at the head of the step;  P = 1  one read behind each of the first R MFMAs
//   R      reads per step (6 MFMAs per step), W waves per workgroup (4 or 8)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int V, int P, int R, int W>
__global__ __launch_bounds__(W * 64, 1) void k(float* out, unsigned long long* cyc, int steps, unsigned seed) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned base = (unsigned)(unsigned long)(lds_ptr_t)lds;
    // conflict-free 16-B slots: row = lane & 31 (128-B rows), chunk swizzled as in the GEMM
    const unsigned addr = base + (unsigned)(wave * 8192 + (lane & 31) * 128 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4));
    for (int i = threadIdx.x; i < 16384; i += W * 64) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i * 7u + seed) % 5u;
    __syncthreads();
    bf16x8 f[2][6];
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) f[b][i][j] = (__bf16)(float)((lane * 7 + i * 3 + j + seed) % 13 - 6);
    bf16x8 scratch[6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) scratch[i][j] = (__bf16)0.0f;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = __builtin_readcyclecounter();
    auto rd = [&](bf16x8& dst, int i) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(0)); (void)i; };
    auto step = [&](auto CB) {
        constexpr int cb = decltype(CB)::value, nb = cb ^ 1;
        if constexpr (V == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); }
        if constexpr (P == 0) {
#pragma unroll
            for (int i = 0; i < R; ++i) { if constexpr (V == 0) rd(scratch[i % 6], i); else rd(f[nb][i % 6], i); }
            SB();
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            // sources: A = f[cb][m % 2 ? 4 : 5] .., B = f[cb][m % 3]: five distinct registers of the buffer, as the GEMM step
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[V == 0 ? 0 : cb][m % 3], f[V == 0 ? 0 : cb][3 + m / 3], acc[m], 0, 0, 0);
            SB();
            if constexpr (P == 1) {
                if (m < R) { if constexpr (V == 0) rd(scratch[m % 6], m); else rd(f[nb][m % 6], m); }
                if (m + 6 < R) { if constexpr (V == 0) rd(scratch[(m + 6) % 6], m); else rd(f[nb][(m + 6) % 6], m); }
                SB();
            }
        }
    };
    for (int s = 0; s < steps; s += 2) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    float t = 0.f;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    for (int i = 0; i < 6; ++i) t += (float)scratch[i][0];
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int V, int P, int R, int W>
void run(float* out, unsigned long long* cyc) {
    auto fn = k<V, P, R, W>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int steps = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(256), dim3(W * 64), 65536, 0, out, cyc, steps, 1u);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fn, dim3(256), dim3(W * 64), 65536, 0, out, cyc, steps, (unsigned)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double cs = 0; for (int i = 0; i < 256; ++i) cs += (double)h[i]; cs /= 256.0;
    const double mf = (double)steps * 6;
    printf("V=%d P=%d R=%d W=%d | %7.1f us per launch | %6.1f cycles per MFMA per wave (%5.1f per SIMD-MFMA) | clock %.2f GHz | %6.0f TF\n", V, P, R, W,
           ms * 1e3 / 3, cs / mf, cs / mf / (W / 4), cs / (ms * 1e3 / 3) / 1e3, 256.0 * W * mf * 32768.0 / (ms / 3 * 1e-3) / 1e12);
}

int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 2048));
    run<0, 0, 0, 4>(out, cyc);
    run<0, 0, 5, 4>(out, cyc);
    run<0, 1, 5, 4>(out, cyc);
    run<1, 0, 5, 4>(out, cyc);
    run<1, 1, 5, 4>(out, cyc);
    run<2, 0, 5, 4>(out, cyc);
    run<2, 1, 5, 4>(out, cyc);
    run<1, 1, 3, 4>(out, cyc);
    run<1, 1, 1, 4>(out, cyc);
    run<1, 1, 7, 4>(out, cyc);
    run<0, 0, 0, 8>(out, cyc);
    run<0, 1, 5, 8>(out, cyc);
    run<1, 0, 5, 8>(out, cyc);
    run<1, 1, 5, 8>(out, cyc);
    return 0;
}
