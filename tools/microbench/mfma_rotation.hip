// mfma_rotation.hip — how many independent accumulators does a lone wave per SIMD need to issue v_mfma_f32_32x32x16_bf16 back
// to back?  R accumulators used in rotation (the same one again every R MFMAs), one wave per SIMD (256-thread workgroups, one
// per CU), cycles per MFMA from s_memtime inside the kernel.  (The prompt GEMM of prefill.hip rotates 4 per wave, gemm_s4 12.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int R>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)(float)((lane * 7 + i * 3 + j + seed) % 13 - 6);
            b[i][j] = (__bf16)(float)((lane * 5 + i + j + seed) % 11 - 5);
        }
    f32x16 acc[R];
    for (int i = 0; i < R; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < iters; ++s) {
#pragma unroll
        for (int m = 0; m < 48; ++m) acc[m % R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 1], b[(m >> 1) & 1], acc[m % R], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < R; ++i)
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int R>
void run(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<R>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1u);
    hipLaunchKernelGGL(k<R>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 2u);
    CK(hipDeviceSynchronize());
    unsigned long long h;
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%2d accumulators in rotation: %.1f cycles per MFMA\n", R, (double)h / (48.0 * iters));
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 64));
    run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<6>(out, cyc); run<8>(out, cyc); run<12>(out, cyc);
    return 0;
}
