// zg_common.h — shared host/device helpers for libzgpt2_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zgpt2.h"

namespace zg {

// ------------------------------------------------------------------------------ error plumbing
void set_error(const char* fmt, ...);
// Diagnostic: the launchers of the decode kernels note the instantiation they picked (zg_debug_last_kernel; bench.py quotes
// it as the roofline's kernel symbol instead of a hand-kept table).
void note_kernel(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
// Test hook ZGPT2_DECODE_PATHS_OFF (bit mask, read per call): the lock-step decode's newer paths switched off so that the paths
// they replaced — still the route of shapes the newer kernels do not take — can be held to the same tokens (tests/test_planes_gpu.py):
// 1 activation planes between kernels, 2 the four-wave plane-fed Linear, 4 tagged hand-overs (tickets instead), 8 LayerNorm
// statistics by tile, 16 line-shaped weight loads, 32 the wave-per-tile lm_head.
int decode_paths_off();

#define ZG_HIP(expr)                                                         \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) return zg::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define ZG_REQUIRE(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            zg::set_error(__VA_ARGS__);    \
            return (code);                 \
        }                                  \
    } while (0)

#define ZG_TRY(expr)             \
    do {                         \
        int _s = (expr);         \
        if (_s != ZG_OK) return _s; \
    } while (0)

// ------------------------------------------------------------------------------ device helpers
typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int kWave = 64;  // CDNA wavefront

__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) {
    return __uint_as_float(packed & 0xFFFF0000u);
}
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
    uint32_t b = __float_as_uint(x);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}

// B24, the 24-bit KV cache element: the fp32 value rounded (nearest even) to 16 mantissa bits — sign, 8 exponent bits, the 15
// upper mantissa bits — kept as a bf16-shaped upper half in one plane and the next 8 mantissa bits in a byte plane with the
// same element index.  Relative error 2^-17 per element.  b24_round returns the 24 bits right-aligned.
__device__ __forceinline__ uint32_t b24_round(float x) {
    const uint32_t b = __float_as_uint(x);
    uint32_t r = (b + 0x7Fu + ((b >> 8) & 1u)) >> 8;
    // a finite value within 2^-17 of FLT_MAX must not round up to inf (0 x inf would poison masked positions, as the fp16 cache's
    // clamp prevents): one ulp back.  inf / NaN inputs keep their class.
    if ((r & 0x7F8000u) == 0x7F8000u && (b & 0x7F800000u) != 0x7F800000u) r -= 1u;
    return r;
}
__device__ __forceinline__ void b24_store(void* cache, size_t lo_off, size_t elem, float x) {
    const uint32_t r = b24_round(x);
    reinterpret_cast<uint16_t*>(cache)[elem] = (uint16_t)(r >> 8);
    reinterpret_cast<uint8_t*>(cache)[lo_off + elem] = (uint8_t)r;
}

// Two fp32 values -> packed bf16 pair (round to nearest even), one gfx950 instruction (v_cvt_pk_bf16_f32); lo half = a.
// Through the compiler's own conversion, NOT an asm statement: the hazard recognizer pads nothing around inline asm, and a
// packed pair that an MFMA reads as an operand right behind the conversion was read half-written — alternating groups of four
// B-operand columns kept the old register contents (the pipelined prompt attention, round 5: tools/dbg_attn_all.py).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// Exact three-term bf16 split of two fp32 values: x = hi + mid + lo (24 = 8 + 8 + 8 mantissa bits; every
// residual is exactly representable).  Outputs are packed pairs {x1 : x0}.  11 VALU instructions per pair —
// the integer-emulated rounding it replaces made the split the longest phase of the batched GEMV prologue.
__device__ __forceinline__ void split3_pk(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - bf16_lo(hi), r1 = x1 - bf16_hi(hi);
    mid = cvt_pk_bf16(r0, r1);
    lo = cvt_pk_bf16(r0 - bf16_lo(mid), r1 - bf16_hi(mid));
}

// Activation planes of the lock-step batch in global memory: the epilogue of the producing kernel writes the exact
// three-term bf16 split of its output row, the consuming Linear loads them as MFMA A fragments — no LDS staging and no
// split in front of its MFMAs.  Element (plane p, batch row m < 8, column k) of a [3][8][K] triple sits at
//   (((k >> 5) * 3 + p) * 8 + m) * 32 + (k & 31)    bf16 elements,
// i.e. the three planes of the 8 rows of one 32-k MFMA step are 1536 contiguous bytes; a triple takes 48 K bytes.
constexpr int kPlaneStep = 3 * 8 * 32;  // elements per 32-k step
__host__ __device__ inline size_t plane_elem(int p, int m, int k) { return ((size_t)((k >> 5) * 3 + p) * 8 + m) * 32 + (k & 31); }

// Kernel-argument hygiene of the decode kernels.  Only the first 14 dwords of the arguments are preloaded into SGPRs at
// wave launch (-amdgpu-kernarg-preload-count=16, two of them hold the kernarg pointer); any other field costs a scalar
// load from the kernarg segment, which the compiler issues where the field is first used and waits for on the spot — a
// cold round trip to memory in a real decode step (every launch has its own argument block).  So (1) whatever an
// address of the up-front vector loads depends on travels in the leading arguments, and (2) fields that the tail of a
// kernel needs are touched with ZG_PIN right behind the vector loads: their scalar loads then complete under the vector
// memory latency instead of in the epilogue.
#define ZG_PIN(v) asm volatile("" ::"s"(v))

// DPP row rotate inside each 16-lane row: every lane reads the lane `n` to its right (cyclic).
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x120 | N, 0xF, 0xF, true));
}

// Sum over each 16-lane row; every lane of the row receives the total.
__device__ __forceinline__ float row16_allsum(float v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    return v;
}

// Read one lane's value into a scalar register (v_readlane_b32): a few cycles, no LDS crossbar.
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}

// Sum over groups of LPR consecutive lanes (LPR in {16, 32, 64}); all lanes get the total.
// Rows are combined through v_readlane (scalar broadcast) rather than ds_bpermute: the LDS crossbar
// costs ~100+ cycles of dependent latency per hop, which dominated the small latency-bound kernels.
template <int LPR>
__device__ __forceinline__ float group_allsum(float v) {
    v = row16_allsum(v);
    if (LPR == 32) {
        const float lo = lane_value(v, 0) + lane_value(v, 16), hi = lane_value(v, 32) + lane_value(v, 48);
        v = (threadIdx.x & 32) ? hi : lo;
    }
    if (LPR == 64) v = (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
    return v;
}

__device__ __forceinline__ float wave_allsum(float v) { return group_allsum<64>(v); }

__device__ __forceinline__ float wave_allmax(float v) {
    v = fmaxf(v, dpp_row_ror<8>(v));
    v = fmaxf(v, dpp_row_ror<4>(v));
    v = fmaxf(v, dpp_row_ror<2>(v));
    v = fmaxf(v, dpp_row_ror<1>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// gelu of src/ops.zig:221-228: 0.5 x (1 + tanh(u)), u = x * 0.7978845608 * (1 + 0.044715 x^2).
// 0.5 (1 + tanh u) == 1 / (1 + exp(-2u)) exactly; this form keeps relative accuracy in both tails.
__device__ __forceinline__ float gelu_ref(float x) {
    const float u = x * 0.7978845608f * (1.0f + 0.044715f * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}

}  // namespace zg
