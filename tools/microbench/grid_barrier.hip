// grid_barrier.hip — what does a device-wide barrier inside ONE persistent kernel cost on MI355X, compared
// with the 1.6-1.9 us kernel boundary of a hipGraph chain (launch_floor.hip)?  Decides whether a
// persistent decode "megakernel" (one launch per token, ~60 phase barriers) can beat one kernel per op.
//   flat : every workgroup does one agent-scope atomicAdd on a single counter, then spins on a generation word
//   tree : workgroups first meet on a per-XCD counter (XCC_ID from the hardware register), the last arriver
//          of each XCD goes to the global counter
// A phase is: barrier, then every workgroup reads what its left neighbour wrote before the barrier (checks
// that the barrier really orders memory) and writes its own slot.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Bar {
    unsigned count;      // arrivals of the current generation
    unsigned pad0[31];
    unsigned gen;        // completed generations
    unsigned pad1[31];
    unsigned xcount[8][32];  // per-XCD arrival counters (one 128-B line each)
};

__device__ __forceinline__ void barrier_flat(Bar* b, unsigned n_wg, unsigned& my_gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++my_gen;
        __threadfence();
        const unsigned prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == n_wg - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, my_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != my_gen) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__device__ __forceinline__ void barrier_tree(Bar* b, unsigned n_wg, unsigned wg_per_xcd, unsigned& my_gen, unsigned xcd) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++my_gen;
        __threadfence();
        const unsigned prev = __hip_atomic_fetch_add(&b->xcount[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool released = false;
        if (prev == wg_per_xcd - 1) {
            __hip_atomic_store(&b->xcount[xcd][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned p2 = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p2 == 7) {
                __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b->gen, my_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                released = true;
            }
        }
        if (!released)
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != my_gen) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void k_phases(Bar* b, float* slots, int n_phases, unsigned gen0, int* errors, int wg_per_xcd) {
    unsigned my_gen = gen0;
    const unsigned n = gridDim.x, me = blockIdx.x, left = (me + n - 1) % n;
    const unsigned xcd = MODE == 1 ? xcc_id() : 0;
    for (int p = 0; p < n_phases; ++p) {
        if (threadIdx.x == 0) slots[(size_t)(p & 1) * n + me] = (float)(p * 7 + (int)me);
        if (MODE == 0) barrier_flat(b, n, my_gen);
        else barrier_tree(b, n, wg_per_xcd, my_gen, xcd);
        if (threadIdx.x == 0) {
            const float v = __builtin_nontemporal_load(&slots[(size_t)(p & 1) * n + left]);
            if (v != (float)(p * 7 + (int)left)) atomicAdd(errors, 1);
        }
    }
}

__global__ void k_xcd_census(int* per_xcd) { if (threadIdx.x == 0) atomicAdd(&per_xcd[xcc_id() & 7], 1); }

int main() {
    Bar* bar; float* slots; int* errors; int* census;
    CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMemset(bar, 0, sizeof(Bar)));
    CK(hipMalloc(&slots, 2 * 4096 * sizeof(float)));
    CK(hipMalloc(&errors, 4)); CK(hipMemset(errors, 0, 4));
    CK(hipMalloc(&census, 32)); CK(hipMemset(census, 0, 32));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int n_wg : {256, 512}) {
        CK(hipMemset(census, 0, 32));
        hipLaunchKernelGGL(k_xcd_census, dim3(n_wg), dim3(256), 0, s, census);
        int h[8]; CK(hipMemcpy(h, census, 32, hipMemcpyDeviceToHost));
        printf("%d workgroups, per-XCD census:", n_wg);
        for (int i = 0; i < 8; ++i) printf(" %d", h[i]);
        printf("\n");
        for (int mode = 0; mode < 2; ++mode) {
            const int phases = 2000;
            unsigned gen0 = 0;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, s));
                if (mode == 0) hipLaunchKernelGGL(k_phases<0>, dim3(n_wg), dim3(256), 0, s, bar, slots, phases, gen0, errors, n_wg / 8);
                else hipLaunchKernelGGL(k_phases<1>, dim3(n_wg), dim3(256), 0, s, bar, slots, phases, gen0, errors, n_wg / 8);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                gen0 += phases;
            }
            int herr; CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
            printf("  %s barrier: %.3f us per phase (ordering errors: %d)\n", mode ? "tree" : "flat", best * 1000.f / phases, herr);
        }
    }
    return 0;
}
