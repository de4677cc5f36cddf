// soffset_bounds_probe.hip — is the SGPR offset of a raw buffer access part of the range check on gfx950?
// LLVM documents only the VGPR offset (voffset) of llvm.amdgcn.raw.buffer.* as bounds-checked.  Two kernels of this library
// rely on voffset + soffset being checked against num_records: the 128-row prompt GEMM switches a DMA piece off by adding
// 2 GiB to its scalar offset (prefill.hip), the prompt attention reads K / V tiles at a scalar tile offset through a
// descriptor that ends behind the prompt's last key (attn_prefill.hip: later keys must read as zero, not as whatever the
// cache holds there).  This program pins the rule: a dword load and a buffer_load ... lds whose voffset is in range and
// whose voffset + soffset is not must return zero / write zero; in-range accesses must return the data.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void probe(const uint32_t* buf, unsigned records, uint32_t* out) {
    __shared__ uint32_t tile[256];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(buf), 0, records, 0x00020000);
    const unsigned lane = threadIdx.x;
    tile[lane] = 0xdeadbeefu;
    tile[64 + lane] = 0xdeadbeefu;
    __syncthreads();
    // [0] in range; [1] voffset in range, soffset pushes it past the end; [2] soffset = 2 GiB (the "kill" offset)
    out[lane] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4u, 0, 0);
    out[64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4u, (int)records, 0);
    out[128 + lane] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4u, (int)0x80000000u, 0);
    // the same through the LDS-DMA form
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)&tile[0], 4, lane * 4u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)&tile[64], 4, lane * 4u, (int)records, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[192 + lane] = tile[lane];
    out[256 + lane] = tile[64 + lane];
}

int main() {
    const unsigned records = 256;  // bytes the descriptor covers; the allocation behind it holds a sentinel
    uint32_t h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
    uint32_t *d, *o;
    hipMalloc(&d, sizeof h);
    hipMalloc(&o, 320 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, records, o);
    uint32_t r[320];
    if (hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        bad += r[l] != 0x1000u + l;          // in range: data
        bad += r[64 + l] != 0;               // voffset + soffset >= num_records: zero (NOT the sentinel at buf[64 + l])
        bad += r[128 + l] != 0;              // + 2 GiB: zero
        bad += r[192 + l] != 0x1000u + l;    // LDS-DMA in range
        bad += r[256 + l] != 0;              // LDS-DMA out of range: zero written
    }
    printf("load past the end via soffset: %#x (sentinel there: %#x); via 2 GiB: %#x; lds-dma: %#x\n", r[64], 0x1000u + 64, r[128], r[256]);
    printf("raw buffer range check covers voffset + soffset: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
