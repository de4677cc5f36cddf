// valu_beside_mfma.hip — what ONE wave per SIMD pays for N vector-ALU instructions of a kind placed behind each
// v_mfma_f32_32x32x16_bf16 (the drain ops of gemm_ov_kernel), and what the same instructions cost alone.
//   cycles per MFMA gap (32 = the matrix pipe's pace) by kind and count; MF = 0: the fillers alone, cycles per filler
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define SB() __builtin_amdgcn_sched_barrier(0)
static const char* kKind[] = {"none", "v_mul_f32", "v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32",
                              "v_mov_b32", "s_nop 0", "v_add_f32 1.0", "v_accvgpr_read", "v_mul_f32 x3 + v_exp"};

template <int KIND>
__device__ __forceinline__ void filler(float (&t)[8], f32x2 (&p)[4], int k) {
    float& a = t[k & 7];
    f32x2& q = p[k & 3];
    if constexpr (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(t[(k + 3) & 7]));
    if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(t[(k + 3) & 7]), "v"(t[(k + 5) & 7]));
    if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    if constexpr (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    if constexpr (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q) : "v"(p[(k + 1) & 3]));
    if constexpr (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q) : "v"(p[(k + 1) & 3]), "v"(p[(k + 2) & 3]));
    if constexpr (KIND == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(t[(k + 3) & 7]), "v"(t[(k + 5) & 7]));
    if constexpr (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(t[(k + 3) & 7]));
    if constexpr (KIND == 9) asm volatile("s_nop 0");
    if constexpr (KIND == 10) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a));
    if constexpr (KIND == 11) asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(a)::"a200");
    if constexpr (KIND == 12) {
        if ((k & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
        else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(t[(k + 3) & 7]));
    }
}

template <int KIND, int N, int MF>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int steps, unsigned seed) {
    const int lane = threadIdx.x & 63;
    bf16x8 f[5];
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 8; ++j) f[i][j] = (__bf16)(float)((lane * 7 + i * 3 + j + seed) % 13 - 6);
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float t[8];
    f32x2 p[4];
    for (int i = 0; i < 8; ++i) t[i] = 1.0f + 1e-3f * (float)((lane + i + seed) % 7);
    for (int i = 0; i < 4; ++i) p[i] = f32x2{t[i], t[i + 4]};
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            if constexpr (MF) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m % 3], f[3 + m / 3], acc[m], 0, 0, 0);
            SB();
#pragma unroll
            for (int n = 0; n < N; ++n) filler<KIND>(t, p, m * N + n);
            SB();
        }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    float r = 0.f;
    for (int i = 0; i < 6; ++i) for (int q = 0; q < 16; ++q) r += acc[i][q];
    for (int i = 0; i < 8; ++i) r += t[i];
    for (int i = 0; i < 4; ++i) r += p[i].x + p[i].y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND, int N, int MF>
void run(float* out, unsigned long long* cyc) {
    auto fn = k<KIND, N, MF>;
    const int steps = 2048;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(256), dim3(256), 0, 0, out, cyc, steps, 1u);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fn, dim3(256), dim3(256), 0, 0, out, cyc, steps, (unsigned)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double cs = 0; for (int i = 0; i < 256; ++i) cs += (double)h[i]; cs /= 256.0;
    const double gaps = (double)steps * 6;
    if (MF) printf("%-22s x %d per MFMA | %6.1f cycles per MFMA gap (+%5.1f, %4.1f per filler) | clock %.2f GHz\n", kKind[KIND], N, cs / gaps,
                   cs / gaps - 32.4, N ? (cs / gaps - 32.4) / N : 0.0, cs / (ms * 1e3 / 3) / 1e3);
    else printf("%-22s alone        | %6.1f cycles per instruction | clock %.2f GHz\n", kKind[KIND], cs / gaps / N, cs / (ms * 1e3 / 3) / 1e3);
}

template <int KIND>
void sweep(float* out, unsigned long long* cyc) {
    run<KIND, 4, 0>(out, cyc);
    run<KIND, 1, 1>(out, cyc);
    run<KIND, 2, 1>(out, cyc);
    run<KIND, 3, 1>(out, cyc);
    run<KIND, 4, 1>(out, cyc);
    run<KIND, 6, 1>(out, cyc);
}

int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 2048));
    run<0, 0, 1>(out, cyc);
    sweep<1>(out, cyc); sweep<2>(out, cyc); sweep<3>(out, cyc); sweep<4>(out, cyc); sweep<5>(out, cyc); sweep<6>(out, cyc);
    sweep<7>(out, cyc); sweep<8>(out, cyc); sweep<9>(out, cyc); sweep<10>(out, cyc); sweep<11>(out, cyc); sweep<12>(out, cyc);
    return 0;
}
