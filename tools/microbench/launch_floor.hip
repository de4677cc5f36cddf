// launch_floor.hip — what does a dependent chain of tiny kernels cost on MI355X?  Measures, for a few
// kernel bodies, the average time per kernel of a 256-kernel chain replayed from a hipGraph
// (HIP events), so the decode step's kernels can be compared against the platform floor.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Big { const float* a; float* out; const int* idx; int n; int pad[50]; };

__global__ __launch_bounds__(256) void k_empty(Big b) { if (b.n == -1) b.out[0] = 1.0f; }
__global__ __launch_bounds__(256) void k_store(Big b) { b.out[blockIdx.x * 256 + threadIdx.x] = 1.0f; }
__global__ __launch_bounds__(256) void k_load1(Big b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float4 v = reinterpret_cast<const float4*>(b.a)[i];
    b.out[i] = v.x + v.y + v.z + v.w;
}
__global__ __launch_bounds__(256) void k_load2dep(Big b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int j = b.idx[i & 1023];
    const float4 v = reinterpret_cast<const float4*>(b.a)[i + j];
    b.out[i] = v.x + v.y + v.z + v.w;
}
__global__ __launch_bounds__(256) void k_load3dep(Big b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int j = b.idx[i & 1023];
    const int k = b.idx[(i + j + 1) & 1023];
    const float4 v = reinterpret_cast<const float4*>(b.a)[i + j + k];
    b.out[i] = v.x + v.y + v.z + v.w;
}
__global__ __launch_bounds__(256) void k_load6(Big b) {  // 6 independent 16-B loads per lane (one gemv pass)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) { const float4 v = reinterpret_cast<const float4*>(b.a)[i + (size_t)q * 65536]; s += v.x + v.y + v.z + v.w; }
    b.out[i] = s;
}
__global__ __launch_bounds__(256) void k_hot(Big b) {  // every wave reads the same 3 KB then its own 16 B
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) { const float4 v = reinterpret_cast<const float4*>(b.a)[(threadIdx.x & 63) + q * 64]; s += v.x + v.y + v.z + v.w; }
    const float4 w = reinterpret_cast<const float4*>(b.a)[i + 4096];
    b.out[i] = s + w.x;
}

template <typename F> float chain(F launch, hipStream_t s, int n = 256, int reps = 20) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1000.f / (n * reps);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *a, *out; int* idx;
    const size_t N = 64u << 20;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&out, N * 4)); CK(hipMalloc(&idx, 4096));
    CK(hipMemset(a, 0, N * 4)); CK(hipMemset(idx, 0, 4096));
    Big b{a, out, idx, 0, {0}};
    const int grids[] = {1, 48, 96, 192, 288, 576, 1571};
    printf("%-12s", "grid");
    const char* names[] = {"empty", "store", "load1", "load2dep", "load3dep", "load6", "hot3KB"};
    for (auto n : names) printf("%10s", n);
    printf("   (us per kernel in a 256-kernel graph chain)\n");
    for (int G : grids) {
        printf("%-12d", G);
        printf("%10.2f", chain([&] { hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, s, b); }, s));
        printf("%10.2f", chain([&] { hipLaunchKernelGGL(k_store, dim3(G), dim3(256), 0, s, b); }, s));
        printf("%10.2f", chain([&] { hipLaunchKernelGGL(k_load1, dim3(G), dim3(256), 0, s, b); }, s));
        printf("%10.2f", chain([&] { hipLaunchKernelGGL(k_load2dep, dim3(G), dim3(256), 0, s, b); }, s));
        printf("%10.2f", chain([&] { hipLaunchKernelGGL(k_load3dep, dim3(G), dim3(256), 0, s, b); }, s));
        printf("%10.2f", chain([&] { hipLaunchKernelGGL(k_load6, dim3(G), dim3(256), 0, s, b); }, s));
        printf("%10.2f", chain([&] { hipLaunchKernelGGL(k_hot, dim3(G), dim3(256), 0, s, b); }, s));
        printf("\n");
    }
    // eager chain for comparison
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_load1, dim3(96), dim3(256), 0, s, b);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager load1 grid 96: %.2f us per kernel\n", ms * 1000.f / 2000);
    return 0;
}
