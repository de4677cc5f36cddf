// mfma_dma_gap.hip — what does one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB) cost a lone wave per SIMD that is issuing
// v_mfma_f32_32x32x16_bf16 back to back?  48 MFMAs per round with P pieces dealt evenly between them (P = 0, 4, 8, 16, 24),
// four waves per workgroup (one per SIMD, as the GEMMs run), one workgroup per CU, source rows L2-resident; cycles per round
// from s_memtime.  Also with two ds_read_b128 behind every MFMA that carries no piece (R = 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int P, int R>
__global__ __launch_bounds__(256, 1) void k(const unsigned short* __restrict__ src, float* out, unsigned long long* cyc, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)(float)((lane * 7 + i * 3 + j + seed) % 13 - 6);
            b[i][j] = (__bf16)(float)((lane * 5 + i + j + seed) % 11 - 5);
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 24, 0x00020000);
    const unsigned rel = (unsigned)(lane >> 3) * 1536u + (unsigned)(lane & 7) * 16u;  // 8 rows of 1536 B, 128 B of each
    const unsigned lds_a = (unsigned)(unsigned long)(lds_ptr_t)lds + (unsigned)lane * 16u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < iters; ++s) {
        const unsigned base = (unsigned)((blockIdx.x * 64 + (s & 7) * 8) * 1536 * 8 + wave * 1536 * 8 * 16);
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 1], b[(m >> 1) & 1], acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int f = P ? (m * P) / 48 : 0;
            if (P && (f * 48 + P - 1) / P == m) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + wave * 32768 + f * 1024), 16, rel, base + (unsigned)f * 128u, 0, 0);
            } else if (R) {
                bf16x8 t0_, t1_;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t0_) : "v"(lds_a + (unsigned)(m * 1024)));
                asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(t1_) : "v"(lds_a + (unsigned)(m * 1024)));
                if (m % 12 == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (P) __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (t == 12345.678f) out[threadIdx.x] = t + lds[lane];
}

template <int P, int R>
void run(const unsigned short* src, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<P, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<P, R>), dim3(256), dim3(256), 131072, 0, src, out, cyc, iters, 1u + rep);
    CK(hipDeviceSynchronize());
    unsigned long long h;
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%2d pieces per 48 MFMAs%s: %7.1f cycles per round (1536 = MFMAs alone)%s\n", P, R ? ", two ds_read_b128 behind the other MFMAs" : "",
           (double)h / iters, P ? "" : "");
}

int main() {
    unsigned short* src; float* out; unsigned long long* cyc;
    CK(hipMalloc(&src, 1u << 24)); CK(hipMemset(src, 0, 1u << 24)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 64));
    run<0, 0>(src, out, cyc); run<4, 0>(src, out, cyc); run<8, 0>(src, out, cyc); run<16, 0>(src, out, cyc); run<24, 0>(src, out, cyc);
    run<0, 1>(src, out, cyc); run<16, 1>(src, out, cyc);
    return 0;
}
