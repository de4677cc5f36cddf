// prefill_epi.h — device helpers shared by the epilogues of the whole-prompt Linears (prefill.hip, gemm_s4.hip).
#pragma once
#include "zg_kernels.h"

namespace zg {

// Cache append (src/ops.zig:152-157) of four consecutive values of one K or V row into the head-major caches [b][h][ctx][64]:
// element offset `off` = ((b H + head) ctx + t) 64 + d, in the cache's storage form (fp32, fp16 saturating, or the 24-bit pair of
// planes — zg_common.h b24_round).
__device__ __forceinline__ void kv_cache_store4(const PrefillQkv& q, void* cache, size_t off, f32x4 v) {
    if (q.kv_mode == 2) {  // four elements: 8 bytes of the bf16 plane, 4 of the byte plane
        const uint32_t r0 = b24_round(v.x), r1 = b24_round(v.y), r2 = b24_round(v.z), r3 = b24_round(v.w);
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(cache) + off) = u32x2{(r0 >> 8) | ((r1 >> 8) << 16), (r2 >> 8) | ((r3 >> 8) << 16)};
        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(cache) + q.kv_lo + off) =
            (r0 & 0xffu) | ((r1 & 0xffu) << 8) | ((r2 & 0xffu) << 16) | (r3 << 24);
    } else if (q.kv_mode) {
        _Float16* d = reinterpret_cast<_Float16*>(cache) + off;
        const float lim = 65504.0f;  // saturate: an inf in the cache would poison masked positions (0 * inf)
        d[0] = (_Float16)fminf(fmaxf(v.x, -lim), lim); d[1] = (_Float16)fminf(fmaxf(v.y, -lim), lim);
        d[2] = (_Float16)fminf(fmaxf(v.z, -lim), lim); d[3] = (_Float16)fminf(fmaxf(v.w, -lim), lim);
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(cache) + off) = v;
    }
}

// ... for four consecutive columns [n, n + 4) of qkv row m (PF_QKV epilogues; columns below E are q: nothing to append)
__device__ __forceinline__ void qkv_cache_store(const PrefillQkv& q, int m, int n, f32x4 v) {
    if (n < q.E) return;
    const int which = n >= 2 * q.E;
    const int e = n - (which ? 2 * q.E : q.E);
    const int b = m / q.P, t = m - b * q.P;
    kv_cache_store4(q, which ? q.v_cache : q.k_cache, (((size_t)b * q.H + (e >> 6)) * q.ctx + t) * 64 + (e & 63), v);
}

}  // namespace zg
