// Development probe: which XCD (HW_REG_XCC_ID) do the blocks of consecutive launches of a replayed hipGraph land on?
// A chain of kernels with the decode step's grid sizes; every block records its XCC_ID.  Prints, per replay and per
// kernel, the XCD of block 0 and whether block b sits on (xcd(block 0) + b) % 8 throughout.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_probe tools/microbench/xcd_probe.hip && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void rec_kernel(unsigned char* out, int slot_stride, const int* replay, int k) {
    if (threadIdx.x == 0) {
        const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
        out[(size_t)(*replay * 64 + k) * slot_stride + blockIdx.x] = (unsigned char)x;
    }
    // a little work so that consecutive kernels do not overlap trivially
    __builtin_amdgcn_s_sleep(32);
}
__global__ void bump_kernel(int* replay) { *replay += 1; }

int main() {
    const int grids[] = {1, 144, 12, 48, 192, 48, 144, 24, 48, 192, 48, 524};
    const int nk = sizeof(grids) / sizeof(int), stride = 1024, replays = 6;
    unsigned char* out; int* rep;
    CK(hipMalloc(&out, (size_t)replays * 64 * stride)); CK(hipMemset(out, 0xff, (size_t)replays * 64 * stride));
    CK(hipMalloc(&rep, 4)); CK(hipMemset(rep, 0, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(rec_kernel, dim3(grids[k]), dim3(256), 0, s, out, stride, rep, k);
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, rep);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<unsigned char> h((size_t)replays * 64 * stride);
    CK(hipMemcpy(h.data(), out, h.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < replays; ++r) {
        printf("replay %d:", r);
        for (int k = 0; k < nk; ++k) {
            const unsigned char* p = &h[(size_t)(r * 64 + k) * stride];
            bool rr = true;
            for (int b = 0; b < grids[k]; ++b) rr &= p[b] == (p[0] + b) % 8;
            printf("  g%d:x%d%s", grids[k], p[0], rr ? "" : "!");
        }
        printf("\n");
    }
    return 0;
}
