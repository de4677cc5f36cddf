// xcd_chain.hip — persistent_chain.hip with the participants confined to ONE XCD: what does an all-to-all hand-over of an
// activation vector cost between workgroups that share an L2?  256 workgroups are launched (the dispatcher deals them round robin
// over the 8 XCDs), the ones that do not sit on XCD `x` leave at once, the others (32: one per CU of that XCD when the chip is
// empty) take dense ranks from a counter and run P dependent phases: poll the N tagged words of the previous phase, reduce, publish
// N / ranks words of the next.  Loads / stores of the hand-over at the scope given (agent: as persistent_chain; workgroup-scope
// atomics still go through the L2 on this chip when glc is forced — variant 1 uses __builtin_nontemporal / sc0 loads).
//   xcd_chain [N=768] [P=2000] [threads=256] [xcd=0] [scope: 0 agent, 1 workgroup-scope atomics]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;

template <int SCOPE>
__device__ inline u64 ld(const u64* p) {
    if (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 v;  // sc0: past the CU's L1, served by this XCD's L2
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE>
__device__ inline void st(u64* p, u64 v) {
    if (SCOPE == 0) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

template <int SCOPE>
__global__ __launch_bounds__(1024) void chain(u64* buf, int N, int P, int xcd, unsigned* rank_ctr, unsigned* nranks, float* out, unsigned* fail) {
    __shared__ float s_red[32];
    __shared__ unsigned s_rank, s_n;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    // every workgroup of the launch, participant or not, is counted in nranks[1] (participants AFTER they took their rank): once
    // that count is the grid size, rank_ctr is final and every participant reads the same number of ranks
    if ((int)xcc != xcd) {
        if (tid == 0) atomicAdd(nranks + 1, 1u);
        return;
    }
    if (tid == 0) {
        s_rank = atomicAdd(rank_ctr, 1u);
        __threadfence();
        atomicAdd(nranks + 1, 1u);
        for (int spins = 0; __hip_atomic_load(nranks + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && spins < (1 << 22); ++spins) __builtin_amdgcn_s_sleep(4);
        s_n = __hip_atomic_load(rank_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned w = s_rank;
    const int G = (int)s_n, per = (N + G - 1) / G;
    float carry = 1.0f;
    for (int p = 1; p <= P; ++p) {
        const u64* src = buf + (size_t)((p - 1) & 1) * N;
        u64* dst = buf + (size_t)(p & 1) * N;
        const unsigned tag = (unsigned)(p - 1);
        float part = 0.0f;
        for (int spins = 0;; ++spins) {
            bool ok = true;
            part = 0.0f;
            for (int i = tid; i < N; i += nt) {
                const u64 v = ld<SCOPE>(src + i);
                ok = ok && (unsigned)(v >> 32) == tag;
                part += __uint_as_float((unsigned)v);
            }
            if (__syncthreads_and(ok)) break;
            if (spins > (1 << 20)) { if (tid == 0) atomicAdd(fail, 1u); break; }
        }
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) s_red[wave] = part;
        __syncthreads();
        float tot = 0.0f;
        for (int i = 0; i < nw; ++i) tot += s_red[i];
        carry = tot * (1.0f / (float)N);
        if (tid < per) {
            const int e = (int)w * per + tid;
            if (e < N) st<SCOPE>(dst + e, ((u64)(unsigned)p << 32) | (u64)__float_as_uint(carry));
        }
        __syncthreads();
    }
    if (tid == 0) out[w] = carry;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 768, P = argc > 2 ? atoi(argv[2]) : 2000, threads = argc > 3 ? atoi(argv[3]) : 256;
    const int xcd = argc > 4 ? atoi(argv[4]) : 0, scope = argc > 5 ? atoi(argv[5]) : 0;
    u64* buf; float* out; unsigned *fail, *ctr, *nr;
    CK(hipMalloc(&buf, (size_t)2 * N * 8)); CK(hipMalloc(&out, 256 * 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&nr, 8));
    u64* h = (u64*)malloc((size_t)2 * N * 8);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 2 * N; ++i) h[i] = (i < N) ? (u64)0x3f800000u : ((u64)0xffffffffu << 32);
        CK(hipMemcpy(buf, h, (size_t)2 * N * 8, hipMemcpyHostToDevice));
        CK(hipMemset(fail, 0, 4)); CK(hipMemset(ctr, 0, 4));
        CK(hipMemset(nr, 0, 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (scope == 0) hipLaunchKernelGGL(chain<0>, dim3(256), dim3(threads), 0, 0, buf, N, P, xcd, ctr, nr, out, fail);
        else hipLaunchKernelGGL(chain<1>, dim3(256), dim3(threads), 0, 0, buf, N, P, xcd, ctr, nr, out, fail);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        float o0; unsigned f, g; CK(hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&g, ctr, 4, hipMemcpyDeviceToHost));
        printf("xcd=%d participants=%u N=%d threads=%d scope=%s P=%d: %.3f us per phase (result %.3f, spin-limit hits %u)\n", xcd, g, N, threads, scope ? "sc0 (L2)" : "agent", P,
               ms * 1e3 / P, o0, f);
    }
    return 0;
}
