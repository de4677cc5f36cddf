// attn.hip — KV-cache decode attention: softmax(alpha * q K^T) V for one query per sequence.
//
// Replaces scaled_dot_product_attention (reference src/ops.zig:249-307: per head two cblas_sgemm
// around ops.softmax) and makes the per-step whole-cache transposes of
// CausalSelfAttention.forward (src/ops.zig:153,158) unnecessary: the cache is addressed through
// (batch, head, position) strides, so both the model tier's head-major cache and the op tier's
// [T, H, hd] cache are read in place.
//
// HBM-bound and latency-bound: a wave owns 64 cache positions of one head and issues all of its
// K and V loads (16 B per lane, 16 lanes per 256-B row, fully coalesced) before any arithmetic.
// Scores are reduced across the 16 lanes of a row with a DPP butterfly reduce-scatter so that
// every lane ends up owning one position; the softmax statistics are wave reductions; the
// probabilities are handed back to the lanes that hold the matching V rows with DPP row broadcasts.
// Four waves (256 positions) are merged in LDS; partials (o[64], m, l) of the <= ctx/256 splits
// per head are combined by the consumer (the c_proj GEMV prologue, or attn_merge_kernel).
#include <stdlib.h>

#include "zg_kernels.h"
#include "zg_runtime.h"

namespace zg {

namespace {

constexpr float kNegBig = -1e30f;

template <typename KV>
__device__ __forceinline__ f32x4 load_kv4(const KV* p);
template <>
__device__ __forceinline__ f32x4 load_kv4<float>(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 load_kv4<_Float16>(const _Float16* p) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 v = *reinterpret_cast<const h4*>(p);
    f32x4 r;
    r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z; r.w = (float)v.w;
    return r;
}

// Butterfly reduce-scatter of 16 values over the 16 lanes of a DPP row: on return lane j of each
// row holds sum over the row's lanes of s[j].
__device__ __forceinline__ float row16_reduce_scatter(float (&s)[16], int lr) {
    // step 1: exchange across lane bit 3 (rotate by 8 == xor 8 inside a 16-lane row)
    float a8[8];
    {
        const bool hi = lr & 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float keep = hi ? s[j + 8] : s[j];
            const float send = hi ? s[j] : s[j + 8];
            a8[j] = keep + dpp_row_ror<8>(send);
        }
    }
    // step 2: across lane bit 2 — xor 4 is not one rotation: take both rotations and select
    float a4[4];
    {
        const bool hi = lr & 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float keep = hi ? a8[j + 4] : a8[j];
            const float send = hi ? a8[j] : a8[j + 4];
            // partner = lr ^ 4.  row_ror:n delivers lane (i - n) mod 16 to lane i, so lanes with
            // bit 2 set take ror 4 (from lr - 4) and the others ror 12 (from lr + 4).
            const float from_lo = dpp_row_ror<4>(send);
            const float from_hi = dpp_row_ror<12>(send);
            a4[j] = keep + (hi ? from_lo : from_hi);
        }
    }
    float a2[2];
    {
        const bool hi = lr & 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = hi ? a4[j + 2] : a4[j];
            const float send = hi ? a4[j] : a4[j + 2];
            const float from_lo = dpp_row_ror<2>(send);
            const float from_hi = dpp_row_ror<14>(send);
            a2[j] = keep + (hi ? from_lo : from_hi);
        }
    }
    {
        const bool hi = lr & 1;
        const float keep = hi ? a2[1] : a2[0];
        const float send = hi ? a2[0] : a2[1];
        const float from_lo = dpp_row_ror<1>(send);
        const float from_hi = dpp_row_ror<15>(send);
        return keep + (hi ? from_lo : from_hi);
    }
}

// Wave 0 of a split hands over its partial (o[lane], M, l).  Op tier / one sequence: plain stores, the consumer merges
// (attn_merge_kernel, or the c_proj prologue).  Lock-step batch with activation planes (AttnArgs.pl_out): the LAST split
// of (b, h) to arrive merges all of them — same arithmetic as merge_attn4 in gemv_internal.h: weights exp(m_s - max m), sums
// in split order, one reciprocal — and writes the head's 64 outputs as the three bf16 planes the c_proj Linear loads as
// MFMA A fragments (zg_common.h plane_elem).  Publish with write-through (agent-scope relaxed atomic) stores, drain
// them, take a ticket, read back with agent-scope loads: the fence-free pattern of the split-K Linears (gemv_mfma16.hip).
__device__ __forceinline__ void publish_partial(const AttnArgs& a, int n_heads, int nsplit, int b, int h, int split, int lane, float o,
                                                float M, float l, unsigned tag) {
    float* part = a.part + (((size_t)b * n_heads + h) * a.max_splits + split) * kPartStride;
    if (a.pl_out == nullptr) {
        part[lane] = o;
        if (lane == 0) {
            part[64] = M;
            part[65] = l;
        }
        return;
    }
    // nsplit = the launched splits: every one of them publishes
    float r = o, lsum = l;
    if (nsplit > 1 && a.part_tag) {
        // Tagged hand-over: every split but the LAST stores (value, tag) words and is done; the last split polls them — one
        // memory-side round trip behind the slowest split instead of the three of the ticket below (drain, ticket, read back).
        // The poller is the last split because a launch's workgroups are dispatched in block order (x, then y = split): the
        // workgroups it waits for are placed BEFORE it, so it can never hold a slot that one of them needs — whatever the
        // occupancy (CU masks, partitions, other resident kernels).  Arithmetic in split order, as the consumer-side merge.
        typedef unsigned long long u64;
        auto pack = [&](float v) { return ((u64)tag << 32) | (u64)__float_as_uint(v); };
        u64* pt0 = a.part_tag + ((size_t)b * n_heads + h) * a.max_splits * kPartStride;  // split 0 of this (sequence, head)
        const int last = nsplit - 1;
        if (split != last) {
            u64* pt = pt0 + split * kPartStride;
            __hip_atomic_store(pt + lane, pack(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) {
                __hip_atomic_store(pt + 64, pack(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pt + 65, pack(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        float mx = M;
        r = 0.0f;
        lsum = 0.0f;
        constexpr int MAXS = 4;
        if (nsplit <= MAXS) {
            u64 vo[MAXS - 1], vm[MAXS - 1], vl[MAXS - 1];
            for (int spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int s = 0; s < MAXS - 1; ++s) {  // splits 0 .. last - 1; surplus slots re-read the last of them (unused below)
                    const u64* ps = pt0 + min(s, last - 1) * kPartStride;
                    vo[s] = __hip_atomic_load(ps + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vm[s] = __hip_atomic_load(ps + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vl[s] = __hip_atomic_load(ps + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(vo[s] >> 32) == tag && (unsigned)(vm[s] >> 32) == tag && (unsigned)(vl[s] >> 32) == tag;
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                if ((unsigned)spins >= a.spin_limit) {  // bounded: never hang the queue — and never pass silently
                    if (lane == 0 && a.fault) __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float ms[MAXS], ls[MAXS], os[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {  // slot s = split s: polled below `last`, this workgroup's own at `last`, weight 0 above
                const bool polled = s < last && s < MAXS - 1;
                ms[s] = polled ? __uint_as_float((unsigned)vm[s < MAXS - 1 ? s : 0]) : s == last ? M : -1e30f;
                ls[s] = polled ? __uint_as_float((unsigned)vl[s < MAXS - 1 ? s : 0]) : l;
                os[s] = polled ? __uint_as_float((unsigned)vo[s < MAXS - 1 ? s : 0]) : o;
            }
            mx = fmaxf(fmaxf(ms[0], ms[1]), fmaxf(ms[2], ms[3]));
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                const float w = __expf(ms[s] - mx);
                lsum = fmaf(w, ls[s], lsum);
                r = fmaf(w, os[s], r);
            }
        } else {  // long contexts: one split at a time, running maximum; this workgroup's own partial last
            float mrun = -1e30f;
            for (int s = 0; s <= last; ++s) {
                float m_s = M, o_s = o, l_s = l;
                if (s < last) {
                    const u64* ps = pt0 + s * kPartStride;
                    u64 vo, vm, vl;
                    for (int spins = 0;; ++spins) {
                        vo = __hip_atomic_load(ps + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vm = __hip_atomic_load(ps + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vl = __hip_atomic_load(ps + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const bool ok = (unsigned)(vo >> 32) == tag && (unsigned)(vm >> 32) == tag && (unsigned)(vl >> 32) == tag;
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                        if ((unsigned)spins >= a.spin_limit) {
                            if (lane == 0 && a.fault) __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    m_s = __uint_as_float((unsigned)vm);
                    o_s = __uint_as_float((unsigned)vo);
                    l_s = __uint_as_float((unsigned)vl);
                }
                const float mnew = fmaxf(mrun, m_s);
                const float w0 = __expf(mrun - mnew), w1 = __expf(m_s - mnew);
                r = fmaf(w1, o_s, r * w0);
                lsum = fmaf(w1, l_s, lsum * w0);
                mrun = mnew;
            }
        }
    } else if (nsplit > 1) {
        typedef __attribute__((address_space(1))) unsigned gu32;
        gu32* gp = (gu32*)part;
        __hip_atomic_store(gp + lane, __float_as_uint(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            __hip_atomic_store(gp + 64, __float_as_uint(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gp + 65, __float_as_uint(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int* cnt = a.merge_cnt + b * n_heads + h;
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != nsplit - 1) return;
        if (lane == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
        const gu32* p0 = (const gu32*)(a.part + ((size_t)b * n_heads + h) * a.max_splits * kPartStride);
        constexpr int MAXS = 4;  // ctx 1024 / 256; more splits take the loop below
        if (nsplit <= MAXS) {
            float ms[MAXS], ls[MAXS], os[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {  // branch-free: surplus splits re-read the last valid one ...
                const gu32* ps = p0 + min(s, nsplit - 1) * kPartStride;
                ms[s] = __uint_as_float(__hip_atomic_load(ps + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                ls[s] = __uint_as_float(__hip_atomic_load(ps + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                os[s] = __uint_as_float(__hip_atomic_load(ps + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s >= nsplit) ms[s] = -1e30f;  // ... and get weight exp(-1e30 - max) == 0
            const float mx = fmaxf(fmaxf(ms[0], ms[1]), fmaxf(ms[2], ms[3]));
            r = 0.0f;
            lsum = 0.0f;
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                const float w = __expf(ms[s] - mx);
                lsum = fmaf(w, ls[s], lsum);
                r = fmaf(w, os[s], r);
            }
        } else {
            float mx = -1e30f;
            for (int s = 0; s < nsplit; ++s)
                mx = fmaxf(mx, __uint_as_float(__hip_atomic_load(p0 + s * kPartStride + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
            r = 0.0f;
            lsum = 0.0f;
            for (int s = 0; s < nsplit; ++s) {
                const gu32* ps = p0 + s * kPartStride;
                const float w = __expf(__uint_as_float(__hip_atomic_load(ps + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - mx);
                lsum = fmaf(w, __uint_as_float(__hip_atomic_load(ps + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), lsum);
                r = fmaf(w, __uint_as_float(__hip_atomic_load(ps + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), r);
            }
        }
    }
    const float v = r * (1.0f / lsum);
    uint32_t hi, mid, lo;
    split3_pk(v, 0.0f, hi, mid, lo);
    const int k = h * 64 + lane;
    a.pl_out[plane_elem(0, b, k)] = (bf16_t)hi;
    a.pl_out[plane_elem(1, b, k)] = (bf16_t)mid;
    a.pl_out[plane_elem(2, b, k)] = (bf16_t)lo;
}

// grid (H, max_splits, B), block 256.  Requires head_dim == 64.
// Arguments (zg_common.h ZG_PIN): the 14 preloaded dwords carry everything a K/V address depends on — q, k, v, the three
// strides (32-bit element counts; bit 31 of st = "sequence length from the control block"), th = t_hi | heads << 20 — plus the two words
// read through a pointer, cw = step control block and ew = epoch of the tags (always readable addresses).  The first
// version took the AttnArgs block alone: its K/V loads were issued behind two dependent scalar round trips (kernarg
// block, then the sequence length behind the control-block pointer in it).
template <typename KV>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ qp, const void* __restrict__ kp,
                                                          const void* __restrict__ vp, unsigned sb, unsigned sh, unsigned st,
                                                          unsigned th, const int* __restrict__ cw, const unsigned* __restrict__ ew,
                                                          const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float s_o[4][64];
    __shared__ float s_m[4], s_l[4];
    const int h = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int t_hi = (int)(th & 0xfffffu), n_heads = (int)(th >> 20);  // (grid sizes are scalar loads from the kernarg segment)
    // t_hi: launch-time upper bound of the sequence length (seq_len itself in the op tier, the
    // 64-position bucket of the captured graph in the model tier).  Every K/V load below depends
    // only on t_hi, so it is in flight while the exact seq_len is still being fetched from the
    // device control block; seq_len is needed for masking alone.
    const int Tc = cw[1];
    const unsigned epoch = ew[0];
    const int T = (st >> 31) ? Tc : t_hi;
    const size_t stride_t = st & 0x7fffffffu;
    const int chunk0 = split * kAttnChunk;
    if (chunk0 >= t_hi) return;  // nothing to attend to in this split (consumer skips it too)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int base = chunk0 + wave * 64;

    const KV* K = reinterpret_cast<const KV*>(kp) + (size_t)b * sb + (size_t)h * sh;
    const KV* V = reinterpret_cast<const KV*>(vp) + (size_t)b * sb + (size_t)h * sh;
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(qp + ((size_t)b * n_heads + h) * 64 + lr * 4);
    const float alpha = 0.125f;  // 1 / sqrt(64), applied to the dot product like sgemm alpha (ops.zig:275)

    float m_w = kNegBig, l_w = 0.0f;
    f32x4 o4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (base < t_hi) {
        // ---- all K and V loads up front: position t = base + 4*i + g, dims 4*lr .. 4*lr+3
        // Branch-free: positions at or beyond t_hi re-read the last valid row (their probability is exactly 0 below: t >= t_hi
        // >= T).  With `if (t < t_hi) load else 0` the compiler merged one loaded component with its zero in a fresh
        // register right behind the second pair of loads — a vmcnt wait, i.e. a whole memory round trip, in the middle of
        // the load sequence, with 14 of the 16 pairs not yet issued.
        f32x4 k4[16], v4[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = min(base + 4 * i + g, t_hi - 1);
            k4[i] = load_kv4<KV>(K + (size_t)t * stride_t + lr * 4);
            v4[i] = load_kv4<KV>(V + (size_t)t * stride_t + lr * 4);
        }
        ZG_PIN(a.progress); ZG_PIN(a.part); ZG_PIN(a.max_splits); ZG_PIN(a.pl_out); ZG_PIN(a.part_tag); ZG_PIN(a.launch_id); ZG_PIN(a.merge_cnt); ZG_PIN(a.fault); ZG_PIN(a.spin_limit);
        pf_count(a.progress);
        // ---- partial dots, then reduce-scatter: lane (g, j) ends with the score of t = base + 4*j + g
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            s[i] = fmaf(q4.x, k4[i].x, fmaf(q4.y, k4[i].y, fmaf(q4.z, k4[i].z, q4.w * k4[i].w)));
        float sc = row16_reduce_scatter(s, lr) * alpha;
        const int t_mine = base + 4 * lr + g;
        if (t_mine >= T) sc = kNegBig;
        // ---- wave softmax statistics
        m_w = wave_allmax(sc);
        const float p = (t_mine < T) ? __expf(sc - m_w) : 0.0f;
        l_w = wave_allsum(p);
        // ---- o += p_t * V_t ; p of position (i, g) lives in lane g*16 + i
        // lane i of each 16-lane row to the whole row: DPP row_newbcast, no LDS crossbar
#define ZG_PV(i)                                                                                                         \
    {                                                                                                                    \
        const float pi = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(p), 0x150 + (i), 0xF, 0xF, false)); \
        o4.x = fmaf(pi, v4[i].x, o4.x);                                                                                  \
        o4.y = fmaf(pi, v4[i].y, o4.y);                                                                                  \
        o4.z = fmaf(pi, v4[i].z, o4.z);                                                                                  \
        o4.w = fmaf(pi, v4[i].w, o4.w);                                                                                  \
    }
        ZG_PV(0) ZG_PV(1) ZG_PV(2) ZG_PV(3) ZG_PV(4) ZG_PV(5) ZG_PV(6) ZG_PV(7)
        ZG_PV(8) ZG_PV(9) ZG_PV(10) ZG_PV(11) ZG_PV(12) ZG_PV(13) ZG_PV(14) ZG_PV(15)
#undef ZG_PV
        // sum the four position groups (lanes 16 apart)
        o4.x += __shfl_xor(o4.x, 16, 64); o4.y += __shfl_xor(o4.y, 16, 64);
        o4.z += __shfl_xor(o4.z, 16, 64); o4.w += __shfl_xor(o4.w, 16, 64);
        o4.x += __shfl_xor(o4.x, 32, 64); o4.y += __shfl_xor(o4.y, 32, 64);
        o4.z += __shfl_xor(o4.z, 32, 64); o4.w += __shfl_xor(o4.w, 32, 64);
    }
    if (lane < 16) *reinterpret_cast<f32x4*>(&s_o[wave][lane * 4]) = o4;
    if (lane == 0) {
        s_m[wave] = m_w;
        s_l[wave] = l_w;
    }
    __syncthreads();
    if (wave == 0) {
        const float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float o = 0.0f, l = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __expf(s_m[w] - M);
            o = fmaf(sc, s_o[w][lane], o);
            l = fmaf(sc, s_l[w], l);
        }
        publish_partial(a, n_heads, (t_hi + kAttnChunk - 1) / kAttnChunk, b, h, split, lane, o, M, l, (epoch << 8) | a.launch_id);
    }
}

// ---- fp16 cache: 16-byte loads.  A 128-byte row (64 halves) is covered by EIGHT lanes of 8 halves each, so a load
// instruction of the wave covers 8 positions and the 64 positions of a wave need 8 + 8 loads of 16 B (the first fp16 path
// kept the fp32 lane map: 8-byte loads, twice the load instructions for the same bytes — and lost to the fp32 cache).
// Lane (g8 = lane / 8, c = lane % 8): dims 8 c .. 8 c + 7 of positions base + 8 i + g8, i = 0..7.  The partial dots are
// reduce-scattered over the 8 lanes of a group (lane c ends with the score of position base + 8 c + g8), the softmax
// statistics are wave reductions as above, p goes back to the group with two row broadcasts and a select.
typedef __attribute__((ext_vector_type(8))) _Float16 h8;

__device__ __forceinline__ float grp8_reduce_scatter(float (&s)[8], int c) {
    float a4[4];
    {
        const bool hi = c & 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float keep = hi ? s[j + 4] : s[j];
            const float send = hi ? s[j] : s[j + 4];
            const float from_lo = dpp_row_ror<4>(send);   // from lane - 4
            const float from_hi = dpp_row_ror<12>(send);  // from lane + 4
            a4[j] = keep + (hi ? from_lo : from_hi);
        }
    }
    float a2[2];
    {
        const bool hi = c & 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = hi ? a4[j + 2] : a4[j];
            const float send = hi ? a4[j] : a4[j + 2];
            const float from_lo = dpp_row_ror<2>(send);
            const float from_hi = dpp_row_ror<14>(send);
            a2[j] = keep + (hi ? from_lo : from_hi);
        }
    }
    const bool hi = c & 1;
    const float keep = hi ? a2[1] : a2[0];
    const float send = hi ? a2[0] : a2[1];
    const float from_lo = dpp_row_ror<1>(send);
    const float from_hi = dpp_row_ror<15>(send);
    return keep + (hi ? from_lo : from_hi);
}

// Eight cached elements of one lane, as loaded: fp16 (16 B), or B24 (16 B of the bf16 plane + 8 B of the byte plane; one
// v_perm_b32 per element rebuilds the fp32 bit pattern {hi16, lo8, 0}, where the fp16 row costs one v_cvt_f32_f16).
template <int MODE>
struct Row8;
template <>
struct Row8<1> {
    h8 v;
    __device__ __forceinline__ void load(const void* base, size_t, size_t elem) { v = *reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(base) + elem); }
    __device__ __forceinline__ float at(int e) const { return (float)v[e]; }
};
template <>
struct Row8<2> {
    u32x4 hi;
    u32x2 lo;
    __device__ __forceinline__ void load(const void* base, size_t lo_off, size_t elem) {
        hi = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + elem);
        lo = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(base) + lo_off + elem);
    }
    __device__ __forceinline__ float at(int e) const {  // (e is a constant after unrolling)
        const uint32_t sel = ((e & 1) ? 0x07060000u : 0x05040000u) | ((uint32_t)(e & 3) << 8) | 0x0cu;
        return __uint_as_float(__builtin_amdgcn_perm(hi[e >> 1], lo[e >> 2], sel));
    }
};

template <int MODE>
__global__ __launch_bounds__(256) void attn_decode_h8_kernel(const float* __restrict__ qp0, const void* __restrict__ kp,
                                                             const void* __restrict__ vp, unsigned sb, unsigned sh, unsigned st,
                                                             unsigned th, const int* __restrict__ cw, const unsigned* __restrict__ ew,
                                                             const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float s_o[4][64];
    __shared__ float s_m[4], s_l[4];
    const int h = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int t_hi = (int)(th & 0xfffffu), n_heads = (int)(th >> 20);  // (grid sizes are scalar loads from the kernarg segment)
    const int Tc = cw[1];
    const unsigned epoch = ew[0];
    const int T = (st >> 31) ? Tc : t_hi;
    const size_t stride_t = st & 0x7fffffffu;
    const int chunk0 = split * kAttnChunk;
    if (chunk0 >= t_hi) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 7, g8 = lane >> 3;
    const int base = chunk0 + wave * 64;
    const size_t head0 = (size_t)b * sb + (size_t)h * sh;  // element index of this head's first row
    const float* qp = qp0 + ((size_t)b * n_heads + h) * 64 + c * 8;
    const f32x4 qa = *reinterpret_cast<const f32x4*>(qp), qb = *reinterpret_cast<const f32x4*>(qp + 4);
    const float q[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    const float alpha = 0.125f;

    float m_w = kNegBig, l_w = 0.0f;
    float o[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (base < t_hi) {
        Row8<MODE> k8[8], v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // branch-free, as above
            const int t = min(base + 8 * i + g8, t_hi - 1);
            k8[i].load(kp, a.kv_lo, head0 + (size_t)t * stride_t + c * 8);
            v8[i].load(vp, a.kv_lo, head0 + (size_t)t * stride_t + c * 8);
        }
        ZG_PIN(a.progress); ZG_PIN(a.part); ZG_PIN(a.max_splits); ZG_PIN(a.pl_out); ZG_PIN(a.part_tag); ZG_PIN(a.launch_id); ZG_PIN(a.merge_cnt); ZG_PIN(a.fault); ZG_PIN(a.spin_limit);
        pf_count(a.progress);
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(q[e], k8[i].at(e), acc);
            s[i] = acc;
        }
        float sc = grp8_reduce_scatter(s, c) * alpha;  // lane (g8, c): position base + 8 c + g8
        const int t_mine = base + 8 * c + g8;
        if (t_mine >= T) sc = kNegBig;
        m_w = wave_allmax(sc);
        const float p = (t_mine < T) ? __expf(sc - m_w) : 0.0f;
        l_w = wave_allsum(p);
        // p of position (i, g8) lives in lane 8 g8 + i: lanes 0-7 of a 16-lane row need lane i of the row, lanes 8-15 lane 8 + i
        const bool up = lane & 8;
#define ZG_PV8(i)                                                                                                          \
    {                                                                                                                      \
        const float plo = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(p), 0x150 + (i), 0xF, 0xF, false));     \
        const float phi = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(p), 0x150 + 8 + (i), 0xF, 0xF, false)); \
        const float pi = up ? phi : plo;                                                                                   \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) o[e] = fmaf(pi, v8[i].at(e), o[e]);                              \
    }
        ZG_PV8(0) ZG_PV8(1) ZG_PV8(2) ZG_PV8(3) ZG_PV8(4) ZG_PV8(5) ZG_PV8(6) ZG_PV8(7)
#undef ZG_PV8
        // sum the eight position groups: lanes 8, 16 and 32 apart
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] += dpp_row_ror<8>(o[e]);
            o[e] += __shfl_xor(o[e], 16, 64);
            o[e] += __shfl_xor(o[e], 32, 64);
        }
    }
    if (lane < 8) {
        *reinterpret_cast<f32x4*>(&s_o[wave][lane * 8]) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(&s_o[wave][lane * 8 + 4]) = f32x4{o[4], o[5], o[6], o[7]};
    }
    if (lane == 0) {
        s_m[wave] = m_w;
        s_l[wave] = l_w;
    }
    __syncthreads();
    if (wave == 0) {
        const float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float ov = 0.0f, l = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __expf(s_m[w] - M);
            ov = fmaf(sc, s_o[w][lane], ov);
            l = fmaf(sc, s_l[w], l);
        }
        publish_partial(a, n_heads, (t_hi + kAttnChunk - 1) / kAttnChunk, b, h, split, lane, ov, M, l, (epoch << 8) | a.launch_id);
    }
}

// Standalone combine of the split partials (op tier; the model tier folds this into c_proj).
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* part, int n_heads, int max_splits,
                                                         int seq_len, float* out) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_heads * 64) return;
    const int h = e >> 6, d = e & 63;
    const int nsplit = (seq_len + kAttnChunk - 1) / kAttnChunk;
    const float* p = part + ((size_t)(b * n_heads + h) * max_splits) * kPartStride;
    float mx = kNegBig;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, p[s * kPartStride + 64]);
    float o = 0.0f, l = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = __expf(p[s * kPartStride + 64] - mx);
        o = fmaf(w, p[s * kPartStride + d], o);
        l = fmaf(w, p[s * kPartStride + 65], l);
    }
    out[(size_t)b * n_heads * 64 + e] = o / l;
}

}  // namespace

int launch_attn_decode(const AttnArgs& a, hipStream_t s) {
    ZG_REQUIRE(a.head_dim == 64, ZG_ERR_UNSUPPORTED, "attention: head_dim %d != 64", a.head_dim);
    ZG_REQUIRE(a.t_hi >= 1, ZG_ERR_ARG, "attention: t_hi not set");
    const int splits = (a.t_hi + kAttnChunk - 1) / kAttnChunk;
    ZG_REQUIRE(splits <= a.max_splits, ZG_ERR_ARG, "attention: t_hi %d needs %d splits > %d", a.t_hi, splits, a.max_splits);
    dim3 grid(a.n_heads, splits, a.batch);
    ZG_REQUIRE(a.stride_b >= 0 && a.stride_h >= 0 && a.stride_t >= 0 && a.stride_b < (1ll << 32) && a.stride_h < (1ll << 32) && a.stride_t < (1ll << 31),
               ZG_ERR_UNSUPPORTED, "attention: strides beyond 32 bits");
    ZG_REQUIRE(a.part_tag == nullptr || a.epoch != nullptr, ZG_ERR_ARG, "attention: tagged partials without an epoch word");
    ZG_REQUIRE(a.t_hi < (1 << 20) && a.n_heads < (1 << 12), ZG_ERR_UNSUPPORTED, "attention: t_hi %d / heads %d too large", a.t_hi, a.n_heads);
    const unsigned th = (unsigned)a.t_hi | ((unsigned)a.n_heads << 20);
    const unsigned sb = (unsigned)a.stride_b, sh = (unsigned)a.stride_h, st = (unsigned)a.stride_t | (a.ctrl ? 0x80000000u : 0u);
    const int* cw = a.ctrl ? reinterpret_cast<const int*>(a.ctrl) : reinterpret_cast<const int*>(ctx().d_zero);
    const unsigned* ew = a.epoch ? a.epoch : reinterpret_cast<const unsigned*>(ctx().d_zero);
    // (fp16 cache: the model tier's head-major layout only — 128-byte rows, 8 lanes x 16 B; the first fp16 path, which kept the
    // fp32 lane map and its 8-byte loads, lost to the fp32 cache and is gone)
    ZG_REQUIRE(!a.kv_mode || a.stride_t == 64, ZG_ERR_UNSUPPORTED, "attention: the 16- / 24-bit caches need contiguous 64-element rows");
    ZG_REQUIRE(a.kv_mode >= 0 && a.kv_mode <= 2, ZG_ERR_ARG, "attention: cache mode %d", a.kv_mode);
    note_kernel(a.kv_mode == 1 ? "attn_decode_h8_kernel<1>" : a.kv_mode == 2 ? "attn_decode_h8_kernel<2>" : "attn_decode_kernel<float>");
    if (a.kv_mode == 1)
        hipLaunchKernelGGL((attn_decode_h8_kernel<1>), grid, dim3(256), 0, s, a.q, a.k, a.v, sb, sh, st, th, cw, ew, a);
    else if (a.kv_mode == 2)
        hipLaunchKernelGGL((attn_decode_h8_kernel<2>), grid, dim3(256), 0, s, a.q, a.k, a.v, sb, sh, st, th, cw, ew, a);
    else
        hipLaunchKernelGGL((attn_decode_kernel<float>), grid, dim3(256), 0, s, a.q, a.k, a.v, sb, sh, st, th, cw, ew, a);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// scaled_dot_product_attention for a head dimension other than 64 (src/ops.zig:249-307 takes any; every GPT-2 configuration has
// 64 and runs the kernels above).  The op tier's general path, not a fast one: one workgroup per (sequence, head); pass 1 finds
// the row maximum of alpha q . k_t, pass 2 goes over the positions 256 at a time — every thread the probability of one position
// into LDS, then thread d the weighted sum of v[., d] over the chunk.  fp32 throughout, alpha = 1 / sqrt(head_dim) applied to the
// dot product like sgemm's alpha (ops.zig:275).
__global__ __launch_bounds__(256) void attn_any_dim_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           long stride_b, long stride_h, long stride_t, int n_heads, int head_dim, int T,
                                                           float* __restrict__ out) {
    __shared__ float s_p[256];
    __shared__ float s_red[8];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* qh = q + ((size_t)b * n_heads + h) * head_dim;
    const float* kh = k + (size_t)b * stride_b + (size_t)h * stride_h;
    const float* vh = v + (size_t)b * stride_b + (size_t)h * stride_h;
    const float alpha = 1.0f / sqrtf((float)head_dim);
    auto score = [&](int t) {
        const float* kt = kh + (size_t)t * stride_t;
        float acc = 0.0f;
        for (int d = 0; d < head_dim; ++d) acc = fmaf(qh[d], kt[d], acc);
        return acc * alpha;
    };
    auto block_max = [&](float x) {
        x = wave_allmax(x);
        if ((tid & 63) == 0) s_red[tid >> 6] = x;
        __syncthreads();
        const float r = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        __syncthreads();
        return r;
    };
    auto block_sum = [&](float x) {
        x = wave_allsum(x);
        if ((tid & 63) == 0) s_red[tid >> 6] = x;
        __syncthreads();
        const float r = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        __syncthreads();
        return r;
    };
    float mx = -3.0e38f;
    for (int t = tid; t < T; t += 256) mx = fmaxf(mx, score(t));
    mx = block_max(mx);
    // output dimensions d = tid, tid + 256, ... (at most 8 per thread: head_dim <= 2048, checked by the launcher)
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
    float l = 0.0f;
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        const float p = t < T ? __expf(score(t) - mx) : 0.0f;
        s_p[tid] = p;
        l += p;
        __syncthreads();
        const int n = min(256, T - t0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = tid + 256 * j;
            if (d < head_dim) {
                float acc = o[j];
                for (int i = 0; i < n; ++i) acc = fmaf(s_p[i], vh[(size_t)(t0 + i) * stride_t + d], acc);
                o[j] = acc;
            }
        }
        __syncthreads();
    }
    l = block_sum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = tid + 256 * j;
        if (d < head_dim) out[((size_t)b * n_heads + h) * head_dim + d] = o[j] * inv;
    }
}

int launch_attn_any_dim(const float* q, const float* k, const float* v, long stride_b, long stride_h, long stride_t, int batch, int n_heads,
                        int head_dim, int seq_len, float* out, hipStream_t s) {
    if (batch == 0) return ZG_OK;  // k.len below one sequence: the reference's loop over the batch does nothing (ops.zig:259-261)
    ZG_REQUIRE(head_dim >= 1 && head_dim <= 2048 && n_heads >= 1 && n_heads < 65536 && batch >= 1 && seq_len >= 1, ZG_ERR_UNSUPPORTED,
               "attention: head_dim %d / heads %d / batch %d", head_dim, n_heads, batch);
    for (int b0 = 0; b0 < batch; b0 += 65535) {  // (the grid's y extent ends at 65535)
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        hipLaunchKernelGGL(attn_any_dim_kernel, dim3(n_heads, nb), dim3(256), 0, s, q + (size_t)b0 * n_heads * head_dim, k + (size_t)b0 * stride_b,
                           v + (size_t)b0 * stride_b, stride_b, stride_h, stride_t, n_heads, head_dim, seq_len, out + (size_t)b0 * n_heads * head_dim);
        ZG_HIP(hipGetLastError());
    }
    return ZG_OK;
}

int launch_attn_merge(const float* part, int batch, int n_heads, int head_dim, int max_splits,
                      int seq_len, float* out, hipStream_t s) {
    ZG_REQUIRE(head_dim == 64, ZG_ERR_UNSUPPORTED, "attention: head_dim %d != 64", head_dim);
    dim3 grid((n_heads * 64 + 255) / 256, batch);
    hipLaunchKernelGGL(attn_merge_kernel, grid, dim3(256), 0, s, part, n_heads, max_splits, seq_len, out);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
