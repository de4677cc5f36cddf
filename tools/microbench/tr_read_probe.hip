// tr_read_probe.hip — what ds_read_b64_tr_b16 delivers (gfx950).  The causal prompt attention reads V fragments (MFMA A operand of
// O^T = V^T P^T: lane = head dimension, 8 consecutive KEYS per lane) out of a [key][d] LDS image with it; this program pins the
// lane map that kernel relies on: in a 16-lane group, lane i supplies the 8-byte address of row i / 4, columns 4 (i % 4) .. + 3 of a
// 4 x 16 block of 16-bit elements, and receives column i of the block (rows 0..3, low half first).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__global__ void probe2(uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t img[32 * 16 * 4];
    const int lane = threadIdx.x;
    for (int e = lane; e < 32 * 64; e += 64) {
        const int key = e / 64, d = e % 64;
        img[((d >> 4) * 32 + key) * 16 + (d & 15)] = (uint16_t)(key * 64 + d);
    }
    __syncthreads();
    const int i = lane & 15, G = lane >> 4;
    const int key = 4 * (lane >> 5) + (i >> 2), db = G & 1;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)img + (unsigned)(((db * 32 + key) * 16 + 4 * (i & 3)) * 2);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[lane * 2] = v.x;
    out[lane * 2 + 1] = v.y;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(probe2, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[128];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int dcol = 16 * ((lane >> 4) & 1) + (lane & 15), k0 = 4 * (lane >> 5);
        const uint16_t got[4] = {(uint16_t)h[2 * lane], (uint16_t)(h[2 * lane] >> 16), (uint16_t)h[2 * lane + 1], (uint16_t)(h[2 * lane + 1] >> 16)};
        for (int j = 0; j < 4; ++j) {
            const int want = (k0 + j) * 64 + dcol;
            if (got[j] != want) ++bad;
        }
        if (lane < 4 || lane == 17 || lane == 40)
            printf("lane %2d: got (key,d) = (%d,%d) (%d,%d) (%d,%d) (%d,%d); want keys %d..%d of d %d\n", lane, got[0] / 64, got[0] % 64, got[1] / 64,
                   got[1] % 64, got[2] / 64, got[2] % 64, got[3] / 64, got[3] % 64, k0, k0 + 3, dcol);
    }
    printf("ds_read_b64_tr_b16 lane map as assumed: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
