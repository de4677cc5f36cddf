// pingpong_probe.hip — two waves on one SIMD, one issuing v_mfma_f32_32x32x16_bf16 back to back (48: the matrix work of one
// attention key tile), the other the vector work of a tile (270 instructions in the attention kernel's mix: subtract, v_exp,
// v_cvt_pk_bf16_f32, shifts / masks of the plane split, adds): does each run as fast as alone?  That is what a "ping-pong"
// arrangement of the prompt attention (the two waves of a SIMD held in opposite phases by workgroup barriers) would rely on.
// Eight waves per workgroup (wave w and w + 4 share SIMD w), one workgroup per CU, all 256 CUs; cycles per round by s_memtime.
//   mode 0  waves 0-3: MFMA rounds, waves 4-7 idle          mode 1  waves 4-7: vector rounds, waves 0-3 idle
//   mode 2  both, each its own kind (ping-pong)             mode 3  all eight waves: MFMA round then vector round, unsynchronised
//   mode 4  all eight waves: the same work interleaved as the compiler does today (10 vector instructions behind 27 of 48 MFMAs)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void valu_unit(float (&t)[16], unsigned (&u)[8], int k) {  // 10 instructions
    float& a = t[k & 15];
    float& b = t[(k + 5) & 15];
    unsigned& c = u[k & 7];
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(t[(k + 3) & 15]));
    asm volatile("v_exp_f32 %0, %0" : "+v"(b));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c) : "v"(t[(k + 7) & 15]), "v"(t[(k + 9) & 15]));
    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[(k + 1) & 7]) : "v"(c));
    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[(k + 2) & 7]) : "v"(c));
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(t[(k + 11) & 15]) : "v"(u[(k + 1) & 7]));
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(t[(k + 12) & 15]) : "v"(u[(k + 2) & 7]));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[(k + 3) & 7]) : "v"(t[(k + 11) & 15]), "v"(t[(k + 12) & 15]));
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[(k + 13) & 15]) : "v"(t[(k + 1) & 15]));
    asm volatile("v_max_f32 %0, %0, %1" : "+v"(t[(k + 14) & 15]) : "v"(t[(k + 2) & 15]));
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int rounds, unsigned seed) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)((float)((lane * 7 + i * 3 + j + seed) % 13 - 6) * 0.01f);
            b[i][j] = (__bf16)((float)((lane * 5 + i + j + seed) % 11 - 5) * 0.01f);
        }
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float t[16];
    unsigned u[8];
    for (int i = 0; i < 16; ++i) t[i] = (float)((lane + i + seed) % 7) * 0.1f;
    for (int i = 0; i < 8; ++i) u[i] = lane + i;
    const bool mf = MODE == 0 ? wave < 4 : MODE == 1 ? false : MODE == 2 ? wave < 4 : true;
    const bool va = MODE == 0 ? false : MODE == 1 ? wave >= 4 : MODE == 2 ? wave >= 4 : true;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < rounds; ++s) {
        if (MODE == 4) {
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 1], b[(m >> 1) & 1], acc[m & 1], 0, 0, 0);
                SB();
                if (m % 16 < 9) valu_unit(t, u, m);  // 27 units of 10 = 270 vector instructions
                SB();
            }
        } else {
            if (mf) {
#pragma unroll
                for (int m = 0; m < 48; ++m) {
                    acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 1], b[(m >> 1) & 1], acc[m & 1], 0, 0, 0);
                    SB();
                }
            }
            if (va) {
#pragma unroll
                for (int m = 0; m < 27; ++m) valu_unit(t, u, m);
            }
        }
        if (MODE == 2) __syncthreads();  // the ping-pong barrier: both kinds of wave meet after every round
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 16; ++q) r += acc[i][q];
    for (int i = 0; i < 16; ++i) r += t[i];
    for (int i = 0; i < 8; ++i) r += (float)u[i];
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE>
void run(float* out, unsigned long long* cyc, const char* what) {
    const int rounds = 400;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, cyc, 10, 1u);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, cyc, rounds, 2u);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    printf("mode %d  %-62s | wave 0 (SIMD 0): %7.0f cycles per round | wave 4 (SIMD 0): %7.0f\n", MODE, what, (double)h[0] / rounds, (double)h[4] / rounds);
}

int main() {
    float* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&cyc, 64));
    run<0>(out, cyc, "48 MFMAs alone (waves 0-3)");
    run<1>(out, cyc, "270 vector instructions alone (waves 4-7)");
    run<2>(out, cyc, "ping-pong: MFMAs on waves 0-3 beside vector on 4-7, barrier");
    run<3>(out, cyc, "all 8 waves: 48 MFMAs then 270 vector, unsynchronised");
    run<4>(out, cyc, "all 8 waves: interleaved (10 vector behind 27 of 48 MFMAs)");
    printf("(a round = the work of ONE wave-tile per SIMD in modes 0-2, of TWO in modes 3-4)\n");
    return 0;
}
