// attn_prefill.hip — causal scaled_dot_product_attention for all n prompt positions at once (src/ops.zig:249-307: per head
// two sgemm calls and a softmax per query row; the reference runs it once per token, src/main.zig:331-334), on the bf16
// matrix cores at fp32 accuracy.
//
// Numerics.  q, k, v arrive as fp32 (the c_attn rows).  Every operand of a matrix product is the EXACT three-term bf16 split
// x = hi + mid + lo (zg_common.h split3_pk), and a product a . b is the six plane products a_i b_j with i + j <= 2 — what is
// dropped is below 2^-24 of the leading term, every bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32: the
// result is fp32-sgemm grade, as the fp32-weight Linears of prefill.hip.  The probabilities are split the same way after the
// softmax.  16 x the matrix rate of v_mfma_f32_32x32x2_f32 for 6 x the products.
//
// Structure.  A workgroup = four waves = four consecutive 32-query blocks of one head (128 queries) against a range of 32-key
// tiles; the K / V tiles are shared: the four waves fetch a tile as fp32 (a quarter each, coalesced 256-byte rows), split it
// once and write the planes into LDS — K as [key][64] rows with the 16-byte chunks XOR-swizzled for ds_read_b128 fragment
// reads, V as four [32 keys][16 d] sub-tiles read back TRANSPOSED by ds_read_b64_tr_b16 (the MFMA wants 8 consecutive keys
// per lane; tools/microbench/tr_read_probe.hip pins the lane map) — double buffered, one barrier per tile, the next tile's
// global loads in flight under the current tile's products.  Everything is computed transposed (S^T = K Q^T, O^T = V^T P^T):
// the query is the lane, so every per-query statistic is one register of one lane and P needs no transpose between the two
// products (the k index of an MFMA step only has to agree between its operands: the keys a lane holds after S^T are the keys
// its P fragment multiplies).  Long rows of few sequences are cut into key ranges over several workgroups whose (m, l, O)
// partials a small kernel merges (one prompt of 1023 tokens would otherwise be 96 workgroups of up to 32 tiles).
#include <type_traits>

#include "zg_kernels.h"

namespace zg {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int kKPlane = 32 * 128;        // one K plane of a tile: [32 keys][64 d] bf16
constexpr int kVBlock = 32 * 32 + 128;   // one 16-d block of a V plane: [32 keys][16 d] bf16, + 128 B: the two blocks a half-wave reads
                                         // in one ds_read_b64_tr_b16 then lie 288 dwords apart — the two halves of the 64 banks
constexpr int kVPlane = 4 * kVBlock;
constexpr int kLds = 2 * 3 * kKPlane + 2 * 3 * kVPlane;  // K ring + V ring, two slots each: 52224 B
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kDefer = 8.0f;  // log2 units

// A query's 32 scores of a tile sit in two lanes (l and l ^ 32, 16 keys each): v_permlane32_swap hands every lane both halves'
// values (vdst = {lower half's, lower half's}, src0 = {upper half's, upper half's}) — no LDS crossbar, no select
__device__ __forceinline__ float both_halves_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float both_halves_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ bf16x8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return __builtin_bit_cast(bf16x8, u32x4{a, b, c, d}); }

__device__ __forceinline__ void store_split4(bf16_t* dst, size_t plane, f32x4 v) {
    uint32_t a[3], b[3];
    split3_pk(v.x, v.y, a[0], a[1], a[2]);
    split3_pk(v.z, v.w, b[0], b[1], b[2]);
#pragma unroll
    for (int p = 0; p < kSplit; ++p) *reinterpret_cast<u32x2*>(dst + p * plane) = u32x2{a[p], b[p]};
}

// key tiles a group of four query blocks needs, and the key ranges (splits) it is cut into
__host__ __device__ inline int group_tiles(int g, int nqb) { return 4 * g + 4 < nqb ? 4 * g + 4 : nqb; }
__host__ __device__ inline int group_splits(int g, int nqb, int nts) { return (group_tiles(g, nqb) + nts - 1) / nts; }

// geo = key tiles per split | most splits of a group << 8 | query groups << 16.  part: [B][H][P][max splits][66] (O^T[64], m, l).
//
// One instruction stream per wave, software-pipelined over the key tiles so that the matrix pipe and the vector ALU of a SIMD
// work side by side WITHIN a wave (the first version ran S -> softmax -> split -> PV -> staging in sequence: 48 MFMAs = 1.5 k
// cycles of a 6.4 k-cycle tile, 45 % matrix-busy with two waves per SIMD):
//   phase A   S^T(t + 1) = K(t + 1) Q^T  (24 MFMAs)   beside   exp / sum / three-plane split of tile t's scores
//   phase B   O^T += V(t)^T P(t)^T       (24 MFMAs)   beside   split + LDS write of the tiles in flight: K(t + 2), V(t + 1)
// so K runs one tile ahead of V through its own two-slot ring; one barrier per tile.  The loop body has no branch (tiles above a
// wave's diagonal are computed fully masked: its siblings need the tile anyway, the barrier would make it wait for them), only
// the rare rescale of the running maximum sits in front of phase A.
// K / V rows of (sequence b, head h), key t: 64 floats at ksrc / vsrc + b kv.stride_b + h kv.stride_h + t kv.stride_t bytes — the k / v
// columns of the qkv rows, or the head-major fp32 caches the c_attn epilogue has just appended to (then the Linear need not
// store those columns at all).
struct AttnKv {
    const char* ksrc;
    const char* vsrc;
    size_t stride_b, stride_h;
    unsigned stride_t;
};
__global__ __launch_bounds__(256, 2) void attn_prefill_pl_kernel(const float* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ part,
                                                                 int P, int E, unsigned geo, const AttnKv kv) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hl = lane >> 5;
    const int nts = (int)(geo & 0xffu), max_s = (int)((geo >> 8) & 0xffu), ng = (int)(geo >> 16);
    // grid (H, B, groups x splits): the dispatcher hands out workgroups in linear order, x fastest — the longest groups of EVERY
    // (sequence, head) first, the short ones fill the tail (with the groups innermost the last sequence's 32-tile group started
    // when the rest of the chip was nearly done: 1.0 waves per SIMD on average instead of 2)
    const int xg = (int)blockIdx.z / max_s, s = (int)blockIdx.z - xg * max_s, g = ng - 1 - xg;
    const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x;
    const int nqb = (P + 31) >> 5;
    const int kt0 = s * nts, kt1 = min((s + 1) * nts, group_tiles(g, nqb));
    if (kt0 >= kt1) return;
#ifdef ZG_STAMPS  // diagnostic build: every workgroup's start / end (s_memrealtime, 100 MHz) and where it ran -> part[] (unsplit launches only)
    const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int n_t = kt1 - kt0;
    const int qb = 4 * g + wave, tq = qb * 32 + l31;
    const size_t row0 = (size_t)b * P;
    const int ld = 3 * E;

    // Q^T fragments (B operand of S^T = K Q^T): lane = query, 8 consecutive d per 16-d slice.  Scaled by log2(e) / sqrt(64) before the
    // split (one fp32 rounding, as the reference's own alpha of sgemm, ops.zig:275) so that the softmax runs on bare v_exp_f32.
    bf16x8 qf[3][4];
    {
        const float sc = 0.125f * kLog2e;
        const float* qp = qkv + (row0 + min(tq, P - 1)) * ld + h * 64 + hl * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * ks), c = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
            uint32_t w[4][3];
            split3_pk(a.x * sc, a.y * sc, w[0][0], w[0][1], w[0][2]);
            split3_pk(a.z * sc, a.w * sc, w[1][0], w[1][1], w[1][2]);
            split3_pk(c.x * sc, c.y * sc, w[2][0], w[2][1], w[2][2]);
            split3_pk(c.z * sc, c.w * sc, w[3][0], w[3][1], w[3][2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) qf[p][ks] = frag(w[0][p], w[1][p], w[2][p], w[3][p]);
        }
    }

    // staging: wave w fetches keys 8 w .. 8 w + 7 of a tile.  K: 16 lanes x 16 B = one 256-byte head row per key (its LDS row is 128
    // contiguous bytes: 32 banks).  V: a 16-lane group takes ONE 16-d block of four keys, the 128 contiguous bytes that block is in
    // LDS (with K's map the four blocks of a row, 288 dwords apart, met on the same banks: a third of all LDS cycles were conflicts).
    // The loads go through a buffer descriptor — the lane's part of the address is fixed, the tile a scalar offset — and rows past
    // the last sequence read as zero (rows past P of another sequence read its first rows: finite, and masked).
    // The descriptors end behind key P - 1 of this (sequence, head): later keys read as zero (a cache row past the prompt may hold
    // anything, and 0 x NaN would poison the second product; the scores of such keys are masked anyway).
    const size_t kv_base = (size_t)b * kv.stride_b + (size_t)h * kv.stride_h;
    const unsigned kv_bytes = (unsigned)(P - 1) * kv.stride_t + 256u;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kv.ksrc + kv_base), 0, kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kv.vsrc + kv_base), 0, kv_bytes, 0x00020000);
    const int skey = wave * 8 + (lane >> 4), sd0 = (lane & 15) * 4;
    const int vkey = wave * 8 + ((lane & 15) >> 2), vd0 = 16 * (lane >> 4) + 4 * (lane & 3);
    const unsigned kvo = (unsigned)skey * kv.stride_t + (unsigned)sd0 * 4u, vvo = (unsigned)vkey * kv.stride_t + (unsigned)vd0 * 4u;
    const unsigned row4 = 4u * kv.stride_t;  // four keys further on
    auto tile_off = [&](int kt) { return (unsigned)kt * 32u * kv.stride_t; };
    auto load_k = [&](int kt, u32x4(&kr)[2]) {
        const unsigned so = tile_off(kt);
        kr[0] = __builtin_amdgcn_raw_buffer_load_b128(rk, kvo, so, 0);
        kr[1] = __builtin_amdgcn_raw_buffer_load_b128(rk, kvo + row4, so, 0);
    };
    auto load_v = [&](int kt, u32x4(&vr)[2]) {
        const unsigned so = tile_off(kt);
        vr[0] = __builtin_amdgcn_raw_buffer_load_b128(rv, vvo, so, 0);
        vr[1] = __builtin_amdgcn_raw_buffer_load_b128(rv, vvo + row4, so, 0);
    };
    char* const kring = lds;                    // [2 slots][3 planes][32 keys][64 d]
    char* const vring = lds + 2 * 3 * kKPlane;  // [2 slots][3 planes][4 d blocks][32 keys][16 d]
    const int kaddr = skey * 128 + (((sd0 >> 3) ^ ((skey >> 1) & 7)) << 4) + ((sd0 >> 2) & 1) * 8;  // (key + 4: same swizzle term + 2)
    const int kaddr1 = (skey + 4) * 128 + (((sd0 >> 3) ^ (((skey + 4) >> 1) & 7)) << 4) + ((sd0 >> 2) & 1) * 8;
    const int vaddr = (vd0 >> 4) * kVBlock + vkey * 32 + (vd0 & 15) * 2;
    auto store_k = [&](int slot, const u32x4(&kr)[2]) {
        char* st = kring + slot * 3 * kKPlane;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 v = __builtin_bit_cast(f32x4, kr[j]);
            uint32_t a[3], c[3];
            split3_pk(v.x, v.y, a[0], a[1], a[2]);
            split3_pk(v.z, v.w, c[0], c[1], c[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(st + p * kKPlane + (j ? kaddr1 : kaddr)) = u32x2{a[p], c[p]};
        }
    };
    auto store_v = [&](int slot, const u32x4(&vr)[2]) {
        char* st = vring + slot * 3 * kVPlane;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 v = __builtin_bit_cast(f32x4, vr[j]);
            uint32_t a[3], c[3];
            split3_pk(v.x, v.y, a[0], a[1], a[2]);
            split3_pk(v.z, v.w, c[0], c[1], c[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(st + p * kVPlane + vaddr + j * 4 * 32) = u32x2{a[p], c[p]};
        }
    };

    // fragment addresses: K rows by key = l31; V blocks by the 16-lane group (d block) and the lane's place in the 4 x 16 read
    const int koff = l31 * 128, kswz = (l31 >> 1) & 7;
    const int voff = ((lane >> 4) & 1) * kVBlock + (4 * hl + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    constexpr int PK[6] = {2, 0, 1, 1, 0, 0}, PQ[6] = {0, 2, 1, 0, 1, 0};  // the six plane pairs (k or v plane, q or p plane), smallest first
    // S^T = K Q^T of the tile in K slot `slot`: ONE accumulator chain, the 24 MFMAs back to back (an MFMA that follows another one on
    // the same accumulator must follow it directly: anything issued between them costs the accumulate forwarding, ~43 cycles —
    // MI355X_MICROARCH.md, per-instruction constants).  Two chains taking the plane pairs in turn — which let this wave's own vector
    // work issue between MFMAs — cost 16 additions per tile to join and ran 1 % slower (89.6 against 90.6 us at 8 x 1023, same box):
    // the SIMD's other wave fills the matrix pipe's shadow anyway (tools/microbench/pingpong_probe.hip).
    auto s_tile = [&](int slot, f32x16& sa) __attribute__((always_inline)) {
        const char* kp = kring + slot * 3 * kKPlane;
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 kf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) kf[p] = *reinterpret_cast<const bf16x8*>(kp + p * kKPlane + koff + (((2 * ks + hl) ^ kswz) << 4));
#pragma unroll
            for (int t = 0; t < 6; ++t) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PK[t]], qf[PQ[t]][ks], sa, 0, 0, 0);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.0f;
    float mrun = -INFINITY, lrun = 0.0f;  // reference point of the exponentials (log2 units) and running sum of this lane's query

    // ---- prologue: K(0), V(0), K(1) staged; S(0); K(2), V(1) in flight
    u32x4 kr[2], vr[2];
    f32x16 sc;
    {
        u32x4 k1[2];
        load_k(kt0, kr);
        load_v(kt0, vr);
        load_k(kt0 + 1, k1);
        store_k(0, kr);
        store_v(0, vr);
        store_k(1, k1);
        load_k(kt0 + 2, kr);
        load_v(kt0 + 1, vr);
        __syncthreads();
        s_tile(0, sc);
        __syncthreads();  // (K slot 0 is rewritten in the first tile's phase B: every wave must have read K(0) out)
    }

    // one tile: i = its index in the range (slot parity SLOT = i & 1), WITH_S = a next tile exists
    auto body = [&](auto SLOT_T, auto WITH_S_T, int i) {
        constexpr int SLOT = decltype(SLOT_T)::value;
        constexpr bool WITH_S = decltype(WITH_S_T)::value;
        const int kt = kt0 + i;
        // sc[r] = log2(e) / 8 * q . k of key kt * 32 + (r & 3) + 8 (r >> 2) + 4 hl against query tq
        if (kt >= qb) {  // diagonal tile (and the tiles past it, which a sibling wave needs): position tq sees keys 0 .. tq
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                sc[r] = key <= tq ? sc[r] : -INFINITY;
            }
        }
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
        mx = both_halves_max(mx);
        // The reference point of the exponentials moves only when some query's maximum grew by more than 2^kDefer since it was
        // set (fp32 accumulators: probabilities up to 2^kDefer lose nothing, and the planes split them exactly) — the rescale
        // of the 32 output registers then runs on a few tiles of a row instead of all.  Everything at the old reference (O, l) is
        // rescaled at the decision, before any probability of this tile exists.
        if (__builtin_amdgcn_ballot_w64(mx > mrun + kDefer) != 0) {
            const float mnew = fmaxf(mrun, mx);
            const float corr = mnew == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(mrun - mnew);  // (a row that has seen no key yet)
            lrun *= corr;
            mrun = mnew;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= corr;
                o1[r] *= corr;
            }
        }
        // ---- phase A: the next tile's scores beside this tile's probabilities
        f32x16 sn;
        if constexpr (WITH_S) s_tile(SLOT ^ 1, sn);
        const float mref = fmaxf(mrun, -1e30f);  // (all keys masked so far: 2^(-inf - mref) = 0, not NaN)
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(sc[r] - mref);
            psum += sc[r];
        }
        lrun += both_halves_sum(psum);
        bf16x8 pf[3][2];  // P^T fragments: registers 8 s' .. 8 s' + 7 are the lane's 8 keys of key slice s'
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            uint32_t w[4][3];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3_pk(sc[8 * sp + 2 * e], sc[8 * sp + 2 * e + 1], w[e][0], w[e][1], w[e][2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) pf[p][sp] = frag(w[0][p], w[1][p], w[2][p], w[3][p]);
        }
        // ---- phase B: O^T += V^T P^T beside the staging of the tiles in flight.  The lane's V fragment = keys 16 s' + 4 hl + {0..3, 8..11}
        // of head dimension l31 (+ 32).
        const char* vp = vring + SLOT * 3 * kVPlane + voff;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            bf16x8 vf[2][3];
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const char* a = vp + p * kVPlane + 2 * dh * kVBlock + 16 * sp * 32;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 8 * 32));
                    vf[dh][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
            for (int t = 0; t < 6; ++t) {  // the two halves of the head dimension in turn: two accumulator chains
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][PK[t]], pf[PQ[t]][sp], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][PK[t]], pf[PQ[t]][sp], o1, 0, 0, 0);
            }
        }
        store_k(SLOT, kr);       // K(i + 2): the slot K(i) left in the previous tile's phase A
        store_v(SLOT ^ 1, vr);   // V(i + 1): the slot V(i - 1) left in the previous tile's phase B
        load_k(kt + 3, kr);
        load_v(kt + 2, vr);
        if constexpr (WITH_S) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = sn[r];
        }
        __syncthreads();
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        int i = 0;
        for (; i + 2 < n_t; i += 2) {
            body(std::integral_constant<int, 0>{}, T{}, i);
            body(std::integral_constant<int, 1>{}, T{}, i + 1);
        }
        if (n_t - i == 2) {
            body(std::integral_constant<int, 0>{}, T{}, i);
            body(std::integral_constant<int, 1>{}, F{}, i + 1);
        } else {
            body(std::integral_constant<int, 0>{}, F{}, i);
        }
    }
#ifdef ZG_STAMPS
    if (tid == 0 && max_s == 1) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(part) + 4 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        st[0] = stamp0;
        st[1] = __builtin_amdgcn_s_memrealtime();
        st[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |         // HW_REG_HW_ID
                ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  // HW_REG_XCC_ID
        st[3] = (unsigned long long)n_t;
    }
#endif
    if (qb >= nqb || tq >= P) return;

    // O^T: the lane holds d = (r & 3) + 8 (r >> 2) + 4 hl (+ 32 in o1) of its query
    if (group_splits(g, nqb, nts) == 1) {  // softmax divides by the sum (ops.zig:239); the c_proj GEMM takes the rows as planes
        const float inv = 1.0f / lrun;
        bf16_t* hi = out + (row0 + tq) * kSplit * E + h * 64;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int d = 8 * gq + 4 * hl;
            store_split4(hi + d, E, f32x4{o0[gq * 4] * inv, o0[gq * 4 + 1] * inv, o0[gq * 4 + 2] * inv, o0[gq * 4 + 3] * inv});
            store_split4(hi + 32 + d, E, f32x4{o1[gq * 4] * inv, o1[gq * 4 + 1] * inv, o1[gq * 4 + 2] * inv, o1[gq * 4 + 3] * inv});
        }
    } else {
        float* pp = part + ((((size_t)b * H + h) * P + tq) * max_s + s) * 66;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int d = 8 * gq + 4 * hl;  // (66-float records: 8-byte aligned)
            *reinterpret_cast<float2*>(pp + d) = float2{o0[gq * 4], o0[gq * 4 + 1]};
            *reinterpret_cast<float2*>(pp + d + 2) = float2{o0[gq * 4 + 2], o0[gq * 4 + 3]};
            *reinterpret_cast<float2*>(pp + 32 + d) = float2{o1[gq * 4], o1[gq * 4 + 1]};
            *reinterpret_cast<float2*>(pp + 32 + d + 2) = float2{o1[gq * 4 + 2], o1[gq * 4 + 3]};
        }
        if (hl == 0) *reinterpret_cast<float2*>(pp + 64) = float2{mrun, lrun};
    }
}

// The key ranges of a query meet: weights 2^(m_s - max m), sums in split order, one division; rows of groups that ran as one
// range were finished by the attention kernel itself.  One thread = four head dimensions of one (sequence, head, query).
__global__ __launch_bounds__(256) void attn_prefill_merge_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, int P, int E, int H, int B,
                                                                 unsigned geo) {
    const int nts = (int)(geo & 0xffu), max_s = (int)((geo >> 8) & 0xffu);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int d = (int)(i & 15) * 4;
    const size_t rec = i >> 4;  // (b, h, q)
    if (rec >= (size_t)B * H * P) return;
    const int q = (int)(rec % P), h = (int)((rec / P) % H), b = (int)(rec / ((size_t)P * H));
    const int n = group_splits(q >> 7, (P + 31) >> 5, nts);
    if (n == 1) return;
    const float* pp = part + rec * max_s * 66;
    float m = -INFINITY;
    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
    float l = 0.0f;
    if (n <= 8) {
        // every load of the (up to eight) ranges issued before the first use: a loop over n with the maximum threaded through it
        // waits for each range's record in turn — 2 n dependent round trips, 7.4 us per launch at one 1023-token prompt
        float2 ml[8], a[8], c[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float* r = pp + (s < n ? s : 0) * 66;
            ml[s] = *reinterpret_cast<const float2*>(r + 64);
            a[s] = *reinterpret_cast<const float2*>(r + d);
            c[s] = *reinterpret_cast<const float2*>(r + d + 2);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) m = s < n ? fmaxf(m, ml[s].x) : m;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float w = s < n ? __builtin_amdgcn_exp2f(ml[s].x - m) : 0.0f;
            l += w * ml[s].y;
            o.x += w * a[s].x; o.y += w * a[s].y; o.z += w * c[s].x; o.w += w * c[s].y;
        }
    } else {
        for (int s = 0; s < n; ++s) m = fmaxf(m, pp[s * 66 + 64]);
        for (int s = 0; s < n; ++s) {
            const float w = __builtin_amdgcn_exp2f(pp[s * 66 + 64] - m);
            l += w * pp[s * 66 + 65];
            const float2 a = *reinterpret_cast<const float2*>(pp + s * 66 + d), c = *reinterpret_cast<const float2*>(pp + s * 66 + d + 2);
            o.x += w * a.x; o.y += w * a.y; o.z += w * c.x; o.w += w * c.y;
        }
    }
    const float inv = 1.0f / l;
    store_split4(out + ((size_t)b * P + q) * kSplit * E + h * 64 + d, E, f32x4{o.x * inv, o.y * inv, o.z * inv, o.w * inv});
}

}  // namespace

// out[M][kSplit E] = split(causal attention of the q / k / v columns of qkv[M][3E]), M = B P rows ordered (b, t).  ws: fp32
// workspace for the partials of split key ranges (B H P max_splits 66 floats; none needed when every group runs as one range).
// k_cache / v_cache != null: K and V come from the head-major fp32 caches [b][h][ctx][64] (positions 0 .. P - 1 just appended
// by the c_attn epilogue) instead of the k / v columns of qkv.
int launch_attn_prefill(const float* qkv, bf16_t* out, int B, int P, int E, int H, float* ws, size_t ws_floats, const float* k_cache,
                        const float* v_cache, int ctx, hipStream_t s, int force_tiles) {
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_prefill_pl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        raised = true;
    }
    ZG_REQUIRE(E == H * 64, ZG_ERR_UNSUPPORTED, "attention prefill: head_dim must be 64 (n_embed %d, %d heads)", E, H);
    const int nqb = (P + 31) / 32, ng = (nqb + 3) / 4;
    // key tiles per workgroup: whole rows once there is a group per CU (measured at 12 heads x 1023 tokens, tools/bench_attn_prefill.py:
    // 4 / 5 sequences = 384 / 480 groups run 57 / 66 us whole against 62 / 75 us cut in two with the merge kernel behind; 3 sequences
    // are even, 2 are faster cut), else ranges of >= 4 tiles chosen so that the launch fills the chip about twice (2 workgroups fit a
    // CU) — as long as the partials fit the workspace
    int nts = 32 * ((nqb + 31) / 32);
    {
        const long groups = (long)B * H * ng;
        if (force_tiles > 0) nts = force_tiles;
        else
            for (int cand = nts; cand >= 4; cand /= 2) {
                long wgs = 0;
                for (int g = 0; g < ng; ++g) wgs += group_splits(g, nqb, cand);
                nts = cand;
                if (wgs * B * H >= 512 || groups >= 256) break;
            }
        if (nts > 255) nts = 255;
    }
    int max_s = group_splits(ng - 1, nqb, nts);
    if (max_s > 1 && (ws == nullptr || (size_t)B * H * P * max_s * 66 > ws_floats)) {  // no room for partials: whole rows
        nts = nqb < 255 ? nqb : 255;
        max_s = group_splits(ng - 1, nqb, nts);
        ZG_REQUIRE(max_s == 1, ZG_ERR_UNSUPPORTED, "attention prefill: %d positions without a workspace", P);
    }
    ZG_REQUIRE(ng < 65536 && max_s < 256, ZG_ERR_UNSUPPORTED, "attention prefill: %d positions", P);
    const unsigned geo = (unsigned)nts | ((unsigned)max_s << 8) | ((unsigned)ng << 16);
    AttnKv kv;
    if (k_cache != nullptr) {  // head-major fp32 caches [b][h][ctx][64]
        kv = AttnKv{reinterpret_cast<const char*>(k_cache), reinterpret_cast<const char*>(v_cache), (size_t)H * ctx * 256, (size_t)ctx * 256, 256u};
    } else {  // the k / v columns of the qkv rows
        kv = AttnKv{reinterpret_cast<const char*>(qkv + E), reinterpret_cast<const char*>(qkv + 2 * E), (size_t)P * 3 * E * 4, (size_t)256, (unsigned)(3 * E * 4)};
    }
    ZG_REQUIRE((size_t)P * kv.stride_t < ((size_t)1 << 31), ZG_ERR_SHAPE, "attention prefill: rows beyond a 32-bit buffer descriptor");
    hipLaunchKernelGGL(attn_prefill_pl_kernel, dim3(H, B, ng * max_s), dim3(256), kLds, s, qkv, out, ws, P, E, geo, kv);
    ZG_HIP(hipGetLastError());
    if (max_s > 1) {
        const size_t n = (size_t)B * H * P * 16;
        hipLaunchKernelGGL(attn_prefill_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, out, P, E, H, B, geo);
        ZG_HIP(hipGetLastError());
    }
    return ZG_OK;
}

}  // namespace zg
