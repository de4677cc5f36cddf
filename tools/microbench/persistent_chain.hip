// persistent_chain.hip — what does one all-to-all hand-over of an activation vector cost INSIDE a persistent kernel, with
// tagged data instead of launch boundaries?  G workgroups (one per CU) run P dependent phases; in every phase each
// workgroup polls the whole N-element vector of the previous phase ((value, tag) 8-byte words, agent-scope loads),
// reduces it (a stand-in for the dot products), and publishes its own N / G elements of the next vector (agent-scope
// stores).  Prints microseconds per phase; compare with the ~2.5-3 us a dependent kernel launch costs in the decode
// graph (1.5 us boundary + the first load round trip).
//   persistent_chain [G=256] [N=768] [P=2000] [threads=256]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;

__global__ __launch_bounds__(1024) void chain(u64* buf, int N, int P, int per, float* out, unsigned* fail) {
    __shared__ float s_red[32];
    __shared__ int s_ok;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int w = blockIdx.x;
    float carry = 1.0f;
    for (int p = 1; p <= P; ++p) {
        const u64* src = buf + (size_t)((p - 1) & 1) * N;
        u64* dst = buf + (size_t)(p & 1) * N;
        const unsigned tag = (unsigned)(p - 1);
        // poll the previous phase's vector (phase 0 = the host-initialised buffer, tag 0)
        float part = 0.0f;
        for (int spins = 0;; ++spins) {
            bool ok = true;
            part = 0.0f;
            for (int i = tid; i < N; i += nt) {
                const u64 v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(v >> 32) == tag;
                part += __uint_as_float((unsigned)v);
            }
            if (__syncthreads_and(ok)) break;
            if (spins > (1 << 22)) { if (tid == 0) atomicAdd(fail, 1u); break; }
        }
        // block reduce (stand-in for the dot products of the phase)
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) s_red[wave] = part;
        __syncthreads();
        float tot = 0.0f;
        for (int i = 0; i < nw; ++i) tot += s_red[i];
        carry = tot * (1.0f / (float)N);
        // publish this workgroup's elements of the next vector
        if (tid < per) {
            const int e = w * per + tid;
            if (e < N) __hip_atomic_store(dst + e, ((u64)(unsigned)p << 32) | (u64)__float_as_uint(carry), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (tid == 0) out[w] = carry;
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 768, P = argc > 3 ? atoi(argv[3]) : 2000;
    const int threads = argc > 4 ? atoi(argv[4]) : 256;
    const int per = (N + G - 1) / G;
    u64* buf; float* out; unsigned* fail;
    CK(hipMalloc(&buf, (size_t)2 * N * 8)); CK(hipMalloc(&out, G * 4)); CK(hipMalloc(&fail, 4));
    u64* h = (u64*)malloc((size_t)2 * N * 8);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 2 * N; ++i) h[i] = (i < N) ? (u64)0x3f800000u : ((u64)0xffffffffu << 32);  // buffer 0: ones with tag 0; buffer 1: stale tag
        CK(hipMemcpy(buf, h, (size_t)2 * N * 8, hipMemcpyHostToDevice));
        CK(hipMemset(fail, 0, 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain, dim3(G), dim3(threads), 0, 0, buf, N, P, per, out, fail);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        float o0; unsigned f; CK(hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("G=%d N=%d threads=%d P=%d: %.3f us per phase (result %.3f, spin-limit hits %u)\n", G, N, threads, P, ms * 1e3 / P, o0, f);
    }
    return 0;
}
