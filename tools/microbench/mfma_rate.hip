// mfma_rate.hip — matrix-pipe ceilings of the GEMM main loop shape on MI355X: 8 waves per CU (512-thread
// workgroups, one per CU), N MFMAs per "stage", optional workgroup barrier per stage, optional LDS reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int PER_STAGE, bool BARRIER, bool LDSREAD>
__global__ __launch_bounds__(512, 1) void k(float* out, int stages, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(float)((lane * 7 + i * 3 + j + seed) % 13 - 6);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)((lane * 5 + i + j + seed) % 11 - 5);
    f32x16 acc[8];
    f32x4 acc4[8];
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    for (int s = 0; s < stages; ++s) {
        if (LDSREAD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + ((threadIdx.x * 16 + i * 8192 + s * 64) & 0xFFF0));
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const bf16x8*>(lds + ((threadIdx.x * 16 + 32768 + i * 8192 + s * 64) & 0xFFF0));
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < PER_STAGE; ++m) {
            if (SHAPE == 32) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 3], b[(m >> 2) & 1], acc[m & 7], 0, 0, 0);
            else acc4[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 3], b[(m >> 2) & 1], acc4[m & 7], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) t += acc[i][r]; for (int r = 0; r < 4; ++r) t += acc4[i][r]; }
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int SHAPE, int PER_STAGE, bool BARRIER, bool LDSREAD>
void run(const char* name, float* out) {
    const int stages = 4096 * 16 / PER_STAGE;
    auto fn = k<SHAPE, PER_STAGE, BARRIER, LDSREAD>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(256), dim3(512), 65536, 0, out, stages, 1u);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fn, dim3(256), dim3(512), 65536, 0, out, stages, (unsigned)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop_per_mfma = SHAPE == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const double flops = 5.0 * 256 * 8 * (double)stages * PER_STAGE * flop_per_mfma;
    printf("%-44s %8.1f TFLOP/s\n", name, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; CK(hipMalloc(&out, 4096));
    run<32, 16, false, false>("32x32x16, 16/stage, no barrier", out);
    run<32, 16, true, false>("32x32x16, 16/stage, barrier", out);
    run<32, 32, true, false>("32x32x16, 32/stage, barrier", out);
    run<32, 64, true, false>("32x32x16, 64/stage, barrier", out);
    run<32, 16, true, true>("32x32x16, 16/stage, barrier + 6 ds_read_b128", out);
    run<32, 32, true, true>("32x32x16, 32/stage, barrier + 6 ds_read_b128", out);
    run<32, 16, false, true>("32x32x16, 16/stage, no barrier + 6 ds_read", out);
    run<16, 16, false, false>("16x16x32, 16/stage, no barrier", out);
    run<16, 64, true, false>("16x16x32, 64/stage, barrier", out);
    return 0;
}
