// What one synchronous op-tier call pays around its kernel: a tiny kernel working on pinned host memory in place, then the
// host learns that it finished — by hipStreamSynchronize (default scheduling / spin scheduling), by a stream write of a
// sequence number into pinned memory the host polls (hipStreamWriteValue32), or by the kernel's own last store.  Also: a small
// host-to-device hand-over as hipMemcpyAsync from pinned memory against a copy kernel reading the pinned buffer.
// usage: sync_cost [spin]      (spin: hipSetDeviceFlags(hipDeviceScheduleSpin) before anything else)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void scale(float* x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0001f + 1.0f; }
__global__ void scale_flag(float* x, int n, volatile unsigned* flag, unsigned seq) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0001f + 1.0f;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
__global__ void copyk(const float* a, float* b, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "spin")) printf("hipSetDeviceFlags(spin): %s\n", hipGetErrorString(hipSetDeviceFlags(hipDeviceScheduleSpin)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 768, iters = 2000;
    float *pin, *dev; unsigned* flag;
    CK(hipHostMalloc((void**)&pin, 1 << 20, hipHostMallocDefault)); CK(hipMalloc((void**)&dev, 1 << 20));
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault)); *flag = 0;
    for (int i = 0; i < n; ++i) pin[i] = 1.0f;
    auto run = [&](const char* name, auto body) {
        for (int i = 0; i < 200; ++i) body(i + 1);
        (void)hipStreamSynchronize(s);
        const double t0 = now();
        for (int i = 0; i < iters; ++i) body(1000 + i);
        (void)hipStreamSynchronize(s);
        printf("%-62s %7.2f us per call\n", name, (now() - t0) / iters * 1e6);
    };
    run("zero-copy kernel + hipStreamSynchronize", [&](unsigned) { hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, pin, n); (void)hipStreamSynchronize(s); });
    run("device kernel + hipStreamSynchronize", [&](unsigned) { hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, dev, n); (void)hipStreamSynchronize(s); });
    run("zero-copy kernel + hipStreamWriteValue32 + host poll", [&](unsigned q) {
        hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, pin, n);
        if (hipStreamWriteValue32(s, flag, q, 0) != hipSuccess) { printf("hipStreamWriteValue32 unsupported\n"); exit(1); }
        while (*(volatile unsigned*)flag != q) {}
    });
    run("zero-copy kernel storing its own flag + host poll", [&](unsigned q) {
        hipLaunchKernelGGL(scale_flag, dim3(1), dim3(1024), 0, s, pin, n, flag, q + 100000u);
        while (*(volatile unsigned*)flag != q + 100000u) {}
    });
    run("zero-copy kernel + 1-thread flag kernel + host poll", [&](unsigned q) {
        hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, pin, n);
        hipLaunchKernelGGL(scale_flag, dim3(1), dim3(64), 0, s, dev, 0, flag, q + 200000u);
        while (*(volatile unsigned*)flag != q + 200000u) {}
    });
    run("hipMemcpyAsync pinned->device 3 KB + kernel + sync", [&](unsigned) {
        (void)hipMemcpyAsync(dev, pin, n * 4, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, dev, n); (void)hipStreamSynchronize(s); });
    run("copy kernel pinned->device 3 KB + kernel + sync", [&](unsigned) {
        hipLaunchKernelGGL(copyk, dim3(3), dim3(256), 0, s, pin, dev, n); hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, dev, n); (void)hipStreamSynchronize(s); });
    run("kernel + hipMemcpyAsync device->pinned 3 KB + sync", [&](unsigned) {
        hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, dev, n); (void)hipMemcpyAsync(pin, dev, n * 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
    run("kernel + copy kernel device->pinned 3 KB + sync", [&](unsigned) {
        hipLaunchKernelGGL(scale, dim3(3), dim3(256), 0, s, dev, n); hipLaunchKernelGGL(copyk, dim3(3), dim3(256), 0, s, dev, pin, n); (void)hipStreamSynchronize(s); });
    run("hipPointerGetAttributes(host pointer)", [&](unsigned) { static float hostbuf[16]; hipPointerAttribute_t a; if (hipPointerGetAttributes(&a, hostbuf) != hipSuccess) (void)hipGetLastError(); });
    run("hipPointerGetAttributes(device pointer)", [&](unsigned) { hipPointerAttribute_t a; (void)hipPointerGetAttributes(&a, dev); });
    return 0;
}
