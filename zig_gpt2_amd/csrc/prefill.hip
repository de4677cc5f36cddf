// prefill.hip — whole-prompt forward: the reference feeds a prompt one token at a time through
// GPT.forward (src/main.zig:331-334); here the n prompt positions of every sequence go through each Block
// together, so the Linears become GEMMs on the matrix cores and attention becomes one causal pass.
//
// Numerics.  The decode path keeps activations in fp32 and weights in bf16 (exact products, fp32
// accumulation).  To stay inside the same tolerance the GEMM A operand here is the fp32 activation split
// into three bf16 terms, x = hi + mid + lo EXACTLY (24 = 8 + 8 + 8 mantissa bits; a two-term split leaves
// 2^-18 |x| behind, which showed up as 2e-5 of the logit scale — outside the parity bar for logits that
// happen to lie near zero), laid out [M][3K] = [hi | mid | lo]; every staged weight tile is multiplied with
// the three planes, so C = (hi + mid + lo) W^T accumulates in one fp32 MFMA chain.  Attention (attn_prefill.hip) splits q, k, v
// and the probabilities the same way and multiplies the six plane products above 2^-24.
//
// Kernels
//   embed_prefill      x[m] = wte[token[m]] + wpe[t]                         (src/main.zig:179-183)
//   ln_split           a = split(LayerNorm(x))                               (src/ops.zig:82-104)
//   prefill_gemm       128 x 128 (or 128 x 256) x 64 tile, weight tile shared by the three planes, LDS-DMA double
//                      buffer, epilogues:
//                        F32 store | fp32 residual add | GELU + split        (ops.zig:21-46, main.zig:136-145, :79-80)
//   (cache append)     K / V columns of the qkv rows -> head-major caches, in the c_attn GEMM epilogue (ops.zig:152-158)
//   (attention)        causal softmax(q k^T / 8) v for all positions: attn_prefill.hip           (src/ops.zig:249-307)
#include <stdlib.h>

#include "prefill_epi.h"
#include "zg_kernels.h"

namespace zg {

int launch_ln_split(const float* x, int M, int E, const float* g, const float* b, float eps, bf16_t* out, hipStream_t s);

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// four consecutive values -> 8 B into each of the kSplit planes (plane stride `plane` elements); the split
// is exact: x = hi + mid + lo (zg_common.h split3_pk)
__device__ __forceinline__ void store_split4(bf16_t* dst, size_t plane, f32x4 v) {
    uint32_t a[3], b[3];
    split3_pk(v.x, v.y, a[0], a[1], a[2]);
    split3_pk(v.z, v.w, b[0], b[1], b[2]);
#pragma unroll
    for (int p = 0; p < kSplit; ++p) *reinterpret_cast<u32x2*>(dst + p * plane) = u32x2{a[p], b[p]};
}

// ------------------------------------------------------------------------------------------ embed
__global__ __launch_bounds__(256) void embed_prefill_kernel(const int* __restrict__ tokens, int token_stride, int P,
                                                            const void* __restrict__ wte, const void* __restrict__ wpe,
                                                            int weight_type, int E, float* __restrict__ x) {
    const int m = blockIdx.x, b = m / P, t = m % P;
    const int tok = tokens[(size_t)b * token_stride + t];
    for (int e = threadIdx.x * 4; e < E; e += 1024) {
        f32x4 o;
        if (weight_type == WT_BF16) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(wte) + (size_t)tok * E + e);
            const u32x2 p = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(wpe) + (size_t)t * E + e);
            o.x = bf16_lo(w.x) + bf16_lo(p.x);
            o.y = bf16_hi(w.x) + bf16_hi(p.x);
            o.z = bf16_lo(w.y) + bf16_lo(p.y);
            o.w = bf16_hi(w.y) + bf16_hi(p.y);
        } else {
            o = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(wte) + (size_t)tok * E + e) +
                *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(wpe) + (size_t)t * E + e);
        }
        *reinterpret_cast<f32x4*>(x + (size_t)m * E + e) = o;
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm + split
// One wave per row.  Same arithmetic as LayerNorm.forward: single pass sum / sum of squares,
// std = sqrt(E[x^2] - mean^2 + eps), (x - mean) / std * g + b   (src/ops.zig:88-101).
__global__ __launch_bounds__(256) void ln_split_kernel(const float* __restrict__ x, int M, int E,
                                                       const float* __restrict__ g, const float* __restrict__ bta,
                                                       float eps, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * E;
    float s = 0.0f, ss = 0.0f;
    for (int e = lane * 4; e < E; e += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + e);
        s += (v.x + v.y) + (v.z + v.w);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = wave_allsum(s);
    ss = wave_allsum(ss);
    const float mean = s / (float)E;
    const float sd = sqrtf(ss / (float)E - mean * mean + eps);
    bf16_t* hi = out + (size_t)row * kSplit * E;
    for (int e = lane * 4; e < E; e += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + e);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + e);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bta + e);
        f32x4 o;
        o.x = (v.x - mean) / sd * gg.x + bb.x;
        o.y = (v.y - mean) / sd * gg.y + bb.y;
        o.z = (v.z - mean) / sd * gg.z + bb.z;
        o.w = (v.w - mean) / sd * gg.w + bb.w;
        store_split4(hi + e, E, o);
    }
}

// ------------------------------------------------------------------------------------------ GEMM
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;
constexpr int kStages = 2;
// stage = weight tile (128 NS rows) + the three activation planes: 64 KiB (NS = 1) or 80 KiB (NS = 2)
constexpr int stage_bytes(int ns) { return (ns + kSplit) * kTileBytes; }
constexpr int lds_bytes(int ns) { return kStages * stage_bytes(ns); }  // 128 / 160 KiB; reused as store staging

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + off);
}

__device__ __forceinline__ float gelu_fast(float x) {
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608f, k2 = k1 * 0.044715f;
    const float arg = x * fmaf(x * x, k2, k1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
}

// A: [M][3K] bf16 (hi | mid | lo), B: [N][K] bf16, bias [N].  C: fp32 [M][ldc] (PF_F32, PF_RESID) or bf16
// [M][3N] split planes (PF_GELU_SPLIT).  N % 64 == 0, K % 64 == 0, any M.
// (the body is shared by prefill_gemm_kernel and prefill_gemm_wp_kernel: sp = K slice of this workgroup, slab = the partial
// slab it writes under PF_PARTIAL)
#ifdef ZG_PF_STAMPS  // diagnostic build only (tools/pf_stamps.py): where one wave of a prompt GEMM spends its K loop
__device__ unsigned long long g_pf_stamps[32];
#define ZG_PFS(i, expr)                                                            \
    do {                                                                           \
        if (pfs_on) {                                                              \
            const unsigned long long t_ = __builtin_readcyclecounter();            \
            expr;                                                                  \
            pfs[i] += __builtin_readcyclecounter() - t_;                           \
        } else {                                                                   \
            expr;                                                                  \
        }                                                                          \
    } while (0)
#else
#define ZG_PFS(i, expr) expr
#endif
template <int EPI, int NS, int NSPL>  // NS 64-column strips per wave: the tile is 128 x (128 NS); NSPL activation planes multiplied
__device__ __forceinline__ void prefill_gemm_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias,
                                                  void* __restrict__ C, int M, int N, int K, int ldc, unsigned tp, int n_tiles,
                                                  const PrefillQkv& qa, const int sp, const int slab) {
    // tp = tiles_n | split-K slices << 12 | rows of the XCD grid << 24: the 14 preloaded argument dwords carry everything the
    // first DMA depends on (zg_common.h ZG_PIN; gridDim is a scalar load from the kernarg segment)
    const int tiles_n = (int)(tp & 0xfffu);
    // nsplit = activation planes multiplied: 3 = exact fp32 activations (hi + mid + lo), 2 = hi + mid only
    // (2^-17 relative per activation, ~2e-5 of the logit scale end to end: inside north_star's 1e-3, outside the
    // strict near-zero floor of the tests; 2/3 of the matrix work)
    extern __shared__ __attribute__((aligned(1024))) char lds[];
#ifdef ZG_PF_STAMPS
    const unsigned long long pfs_entry = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x;
    const int n_sp = (int)((tp >> 12) & 0xfffu);  // split-K slices (PF_PARTIAL only; otherwise 1)
    // Workgroups are dealt round robin over the eight XCDs, each with its own L2: XCD x gets workgroups x, x + 8, ...  Which tiles
    // those are decides what every L2 has to fetch.  gr != 0 (tp bits 24-27; the tile counts divide): the XCDs form a gr x 8 / gr
    // grid of equal blocks of tiles_m / gr x tiles_n / (8 / gr) tiles, so an XCD reads that many row tiles of the activation planes
    // and that many column tiles of the weights (c_fc at 1023 tokens, 8 x 24 tiles: 2 x 12 per XCD = 1.2 + 2.4 MB through a 4-MB L2,
    // where one row tile against ALL of W was 0.6 + 4.7 MB: FETCH per launch 32.6 -> 23.4 MB, c_attn 41.8 -> 27.9 MB; the time did not
    // move — 1.50 against 1.52 ms per 1023-token prompt — the loop is not fetch-bound: profiles/round4_prefill_xcd_blocks.txt).  gr == 0: contiguous ranges of the row-major
    // tile order.
    const int xcd = bid & 7, loc = bid >> 3, gr = (int)((tp >> 24) & 0xfu);
    int tm, tn;
    if (gr == 0) {
        const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
        const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        tm = tile / tiles_n;
        tn = tile % tiles_n;
    } else {
        const int gc = 8 / gr, cp = tiles_n / gc, rp = (n_tiles / tiles_n) / gr;
        const int xr = xcd / gc, xc = xcd - xr * gc, lr_ = loc / cp;
        tm = xr * rp + lr_;
        tn = xc * cp + (loc - lr_ * cp);
    }
    constexpr int BNT = BN * NS, NJ = 2 * NS, kStageB = stage_bytes(NS), kBBytes = NS * kTileBytes;
    const int m0 = tm * BM, n0 = tn * BNT;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // A K-step stages ONE 128 x 64 weight tile and the matching tile of each of the three activation planes;
    // the weight fragments are read from LDS once and multiplied with all three planes (the first version
    // walked the weight rows once per plane: 2 tiles per 16 MFMAs, and was bound by LDS-DMA at ~7 TB/s; this
    // is 4 tiles per 48 MFMAs).  Two 64-KiB slots: the DMA of step t + 1 runs under the MFMAs of step t.
    // Split-K slice sp takes the K-steps [t0, nt).
    const int nk = K / BK;
    const int t0 = (int)((long)nk * sp / n_sp), nt = (int)((long)nk * (sp + 1) / n_sp);
    // A K-step's operands arrive as 4 NS + 4 NSPL one-KiB pieces per wave (8 rows x 128 B each), LDS-DMA through buffer
    // descriptors: the per-lane part of an address (row within the piece, swizzled 16-byte chunk) is one VGPR per operand and
    // piece parity, everything else — tile, piece, plane, K-step — a scalar offset, so a piece costs two scalar instructions and
    // the load (the first version computed a 64-bit VGPR address per piece: ~6 vector instructions each, and the matrix pipe
    // waited for them: ~540 of the 2950 cycles of a K-step).  Rows past the operand's end are out of the descriptor's range and
    // arrive as zeros (they are never stored).
    constexpr int NPC = 4 * NS + 4 * NSPL;  // pieces per wave and K-step
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (unsigned)((size_t)M * kSplit * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (unsigned)((size_t)N * K * 2), 0x00020000);
    unsigned rel_a[2], rel_b[2];
    {
        const unsigned l3 = (unsigned)lane >> 3, pos = (unsigned)lane & 7u;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned chunk = pos ^ (((l3 >> 1) + 4u * e) & 7u);
            rel_a[e] = l3 * (unsigned)(kSplit * K * 2) + chunk * 16u;
            rel_b[e] = l3 * (unsigned)(K * 2) + chunk * 16u;
        }
    }
    // kill: 0, or an offset past every descriptor's end — the piece then fetches nothing and writes zeros (the step behind the
    // last one: an unconditional piece costs two scalar adds, a tested one a branch between two MFMAs)
    auto issue_piece = [&](int t, int idx, unsigned kill = 0u) {
        char* slot = lds + ((t - t0) & 1) * kStageB;
        if (idx < 4 * NS) {
            const int piece = wv * (4 * NS) + idx;
            const unsigned soff = ((unsigned)(n0 + piece * 8) * (unsigned)K + (unsigned)(t * BK)) * 2u + kill;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(slot + piece * 1024), 16, rel_b[idx & 1], soff, 0, 0);
        } else {
            const int p = (idx - 4 * NS) >> 2, q = (idx - 4 * NS) & 3, piece = wv * 4 + q;
            const unsigned soff = ((unsigned)(m0 + piece * 8) * (unsigned)(kSplit * K) + (unsigned)(p * K + t * BK)) * 2u + kill;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(slot + kBBytes + p * kTileBytes + piece * 1024), 16, rel_a[q & 1], soff, 0, 0);
        }
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int idx = 0; idx < NPC; ++idx) issue_piece(t, idx);
    };
    if (t0 < nt) issue(t0);
    if (EPI == PF_QKV) {  // the epilogue's argument-block fields, fetched under the first DMA
        ZG_PIN(qa.P); ZG_PIN(qa.E); ZG_PIN(qa.H); ZG_PIN(qa.ctx); ZG_PIN(qa.kv_mode); ZG_PIN(qa.kv_lo); ZG_PIN(qa.k_cache); ZG_PIN(qa.v_cache);
    }

    const int frow = lane & 31, fk = lane >> 5;
#ifdef ZG_PF_STAMPS
    const bool pfs_on = blockIdx.x == 8 && blockIdx.y == 0 && wave == 0;
    unsigned long long pfs[4] = {0, 0, 0, 0};
    const unsigned long long pfs_t0 = __builtin_readcyclecounter();
#endif
    for (int t = t0; t < nt; ++t) {
        ZG_PFS(0, __builtin_amdgcn_s_waitcnt(0x0f70));  // vmcnt(0): this wave's pieces of stage t landed
        ZG_PFS(1, __builtin_amdgcn_s_barrier());        // everyone's did, and nobody still reads the other slot
        // (Two copies of the body, with and without the pieces, were measured too: the compiler then keeps the accumulators in
        // VGPRs across the loop and moves all 64 to the accumulation file and back around the MFMAs of every step — 3257
        // against 2817 cycles per K-step.)
#if defined(ZG_PF_ABL) && (ZG_PF_ABL & 1)
        const bool more = false;  // ablation: no LDS-DMA in the loop (stale operands)
#else
        const bool more = t + 1 < nt;
#endif
        const unsigned kill = more ? 0u : 0x80000000u;  // (operands < 2 GiB: offset + kill never wraps, and lies past every descriptor's end)
        const char* cur = lds + ((t - t0) & 1) * kStageB;
        // 12 groups (16-k slice kk, plane p = 2, 1, 0: smallest plane first) of 2 NJ MFMAs.  The fragments of group g + 1 are
        // read IN FRONT of group g's MFMAs (two register sets, sched barriers pin the order): with one wave per SIMD nothing
        // else covers the LDS latency — the first version read a group's fragments, waited, multiplied, and spent as long
        // waiting as multiplying.  Only the first group of a K-step is exposed (its data arrives with the barrier above).
        // (The plane count is a template parameter: a run-time test around the MFMAs made the compiler wait for ALL LDS reads —
        // the prefetched ones included — in front of every group.)
        // The reads are hand-issued (asm) with COUNTED waits: left to itself the compiler waits for lgkmcnt(0) in front of
        // every second group, i.e. also for the reads it has just issued for the group after.
        bf16x8 bq[2][NJ], aq[2][NSPL][2];
        const unsigned cur_a = (unsigned)(unsigned long)(lds_ptr_t)cur;
        auto rd = [&](bf16x8& dst, unsigned tile_off, int row, int chunk) {
            const unsigned addr = cur_a + tile_off + (unsigned)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
        };
        // fragment r of 16-k slice kk, in the order the slice uses them: the NJ weight-row fragments of the wave's columns, then
        // its two row fragments of every plane, smallest plane first
        constexpr int RS = NJ + 2 * NSPL;   // fragment reads per slice
        constexpr int MS = 2 * NJ * NSPL;   // MFMAs per slice
        constexpr int RSLOT = (RS + 1) / 2; // MFMAs of a slice that carry two reads of the next slice each
        auto rd_item = [&](int kk, int r, bf16x8(&bb)[NJ], bf16x8(&aa)[NSPL][2]) {
            if (r < NJ) rd(bb[r], 0u, wn * 64 * NS + r * 32 + frow, kk * 2 + fk);
            else {
                const int pi = (r - NJ) >> 1, i = (r - NJ) & 1, p = NSPL - 1 - pi;
                rd(aa[p][i], (unsigned)(kBBytes + p * kTileBytes), wm * 64 + i * 32 + frow, kk * 2 + fk);
            }
        };
#pragma unroll
        for (int r = 0; r < RS; ++r) rd_item(0, r, bq[0], aq[0]);  // the step's first slice: in one burst behind the barrier
        // ... and while those reads are on their way (nothing can be multiplied yet) the first pieces of the next stage
        constexpr int PRE = NPC >= 12 ? 4 : 2;  // (8 measured: the wait at the next barrier goes, the body takes it back)
#pragma unroll
        for (int k = 0; k < PRE; ++k) issue_piece(t + 1, k, kill);
        // One instruction stream per wave, nothing else on its SIMD: whatever the wave issues between two MFMAs runs in the 32
        // cycles the matrix pipe is busy with the first, and whatever it issues in a burst leaves the pipe idle.  Stamps and
        // ablations of the loop (tools/pf_stamps.py, -DZG_PF_ABL) priced the bursts of the first version per K-step of 48 MFMAs
        // = 1536 cycles: the 16 LDS-DMA pieces issued in groups of 2-3 behind MFMA groups +540 cycles, the fragment reads of a
        // slice issued in front of its MFMAs +130, on a bare loop of ~1950 (barrier, first slice's read latency).  Now every MFMA
        // carries at most two fragment reads of the NEXT slice (the first RSLOT MFMAs of a slice: the last read then has more
        // than 250 cycles of cover) or one LDS-DMA piece of the next stage (the MFMAs behind them, earliest slices first); the
        // wait in front of a slice is lgkmcnt(0) — everything older is that slice's fragments.  Same MFMA order as before.
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cb = kk & 1;
            // The wait names slice kk's fragments as read-write operands: for the compiler they are ready straight behind the
            // read statements, and only an operand tie keeps a copy, coalesce or spill of those registers from being placed in
            // front of the wait (sched_barrier pins instruction order, not what register allocation inserts).
#define ZG_TIE2(x) "+v"(x[0]), "+v"(x[1])
#define ZG_TIE4(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[NJ > 2 ? 2 : 0]), "+v"(x[NJ > 2 ? 3 : 1])
            if constexpr (NSPL == 3 && NJ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : ZG_TIE2(aq[cb][0]), ZG_TIE2(aq[cb][1]), ZG_TIE2(aq[cb][2]), ZG_TIE2(bq[cb])::"memory");
            else if constexpr (NSPL == 3) asm volatile("s_waitcnt lgkmcnt(0)" : ZG_TIE2(aq[cb][0]), ZG_TIE2(aq[cb][1]), ZG_TIE2(aq[cb][2]), ZG_TIE4(bq[cb])::"memory");
            else if constexpr (NSPL == 2 && NJ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : ZG_TIE2(aq[cb][0]), ZG_TIE2(aq[cb][1]), ZG_TIE2(bq[cb])::"memory");
            else if constexpr (NSPL == 2) asm volatile("s_waitcnt lgkmcnt(0)" : ZG_TIE2(aq[cb][0]), ZG_TIE2(aq[cb][1]), ZG_TIE4(bq[cb])::"memory");
            else if constexpr (NJ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : ZG_TIE2(aq[cb][0]), ZG_TIE2(bq[cb])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : ZG_TIE2(aq[cb][0]), ZG_TIE4(bq[cb])::"memory");
#undef ZG_TIE4
#undef ZG_TIE2
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                const int pi = m / (2 * NJ), i = (m / NJ) & 1, j = m % NJ, p = NSPL - 1 - pi;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[cb][p][i], bq[cb][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (m < RSLOT) {
#if !(defined(ZG_PF_ABL) && (ZG_PF_ABL & 2))
                    if (kk + 1 < 4) {
                        rd_item(kk + 1, 2 * m, bq[cb ^ 1], aq[cb ^ 1]);
                        if (2 * m + 1 < RS) rd_item(kk + 1, 2 * m + 1, bq[cb ^ 1], aq[cb ^ 1]);
                    }
#endif
                } else {
                    // pieces spread evenly over the free MFMAs of the first three slices (packed into the first ones the four waves
                    // asked the CU's one global -> LDS path for a whole 64-KiB stage within a third of the step)
                    constexpr int F = MS - RSLOT, NREST = NPC - PRE, SPAN = (NREST <= 3 * F) ? 3 * F : 4 * F;
                    const int f = kk * F + (m - RSLOT);
                    const int k = (f * NREST + SPAN - 1) / SPAN;  // the piece whose slot floor(k SPAN / NREST) could be f
                    if (f < SPAN && k < NREST && (k * SPAN) / NREST == f) issue_piece(t + 1, PRE + k, kill);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef ZG_PF_STAMPS
    const unsigned long long pfs_t1 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the (empty) pieces of the step behind the last one have written their zeros
    __builtin_amdgcn_s_barrier();        // ring no longer read: it becomes the store staging area

    // epilogue: each 64 x 64 fp32 strip of the wave goes through LDS so that global accesses are 16-B row segments
    float* wtile = reinterpret_cast<float*>(lds + wave * (64 * 64 * 4));
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int nw = n0 + (wn * NS + st) * 64;
        if (nw >= N) break;  // N % 64 == 0: a strip is entirely inside or entirely outside
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + frow;
            const float bv = (EPI != PF_PARTIAL && bias) ? bias[nw + col] : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    float v = acc[i][st * 2 + j][r] + bv;
                    if (EPI == PF_GELU_SPLIT) v = gelu_fast(v);
                    wtile[row * 64 + col] = v;
                }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int c = it * 64 + lane;
            const int row = c >> 4, cc = c & 15;
            const int gm = m0 + wm * 64 + row;
            if (gm >= M) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(wtile + row * 64 + cc * 4);
            if (EPI == PF_GELU_SPLIT) {
                store_split4(reinterpret_cast<bf16_t*>(C) + (size_t)gm * kSplit * N + nw + cc * 4, N, v);
            } else if (EPI == PF_PARTIAL) {  // C is the workspace [n_sp][M][N]
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + ((size_t)slab * M + gm) * N + nw + cc * 4) = v;
            } else {
                float* dst = reinterpret_cast<float*>(C) + (size_t)gm * ldc + nw + cc * 4;
                if (EPI == PF_RESID) v += *reinterpret_cast<const f32x4*>(dst);
                *reinterpret_cast<f32x4*>(dst) = v;
                if (EPI == PF_QKV) qkv_cache_store(qa, gm, nw + cc * 4, v);
            }
        }
    }
#ifdef ZG_PF_STAMPS
    if (pfs_on && lane == 0) {
        g_pf_stamps[0] = (unsigned long long)(nt - t0);
        g_pf_stamps[1] = pfs[0];
        g_pf_stamps[2] = pfs[1];
        g_pf_stamps[3] = pfs_t1 - pfs_t0;                       // the K loop
        g_pf_stamps[4] = __builtin_readcyclecounter() - pfs_t1;  // barrier + epilogue
        g_pf_stamps[5] = (unsigned long long)EPI;
        g_pf_stamps[6] = pfs_t0 - pfs_entry;  // entry .. first K-step (descriptors, zeroed accumulators, first stage issued)
    }
#endif
}

template <int EPI, int NS, int NSPL = kSplit>
__global__ __launch_bounds__(256, 1) void prefill_gemm_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                              const float* __restrict__ bias, void* __restrict__ C, int M,
                                                              int N, int K, int ldc, unsigned tp, int n_tiles,
                                                              const PrefillQkv qa) {
    prefill_gemm_body<EPI, NS, NSPL>(A, B, bias, C, M, N, K, ldc, tp, n_tiles, qa, (int)blockIdx.y, (int)blockIdx.y);
}

// fp32 WEIGHTS: B is the exact three-term bf16 split of the fp32 matrix, plane-major [3][N][K] (hi, mid, lo).  The six plane
// products above 2^-24 of the leading one are three passes over the K slices — w_lo x a_hi, w_mid x (a_mid + a_hi),
// w_hi x (a_lo + a_mid + a_hi): one weight plane each with 1 / 2 / 3 activation planes — dealt over blockIdx.y = pass * n_sp +
// slice and written as partial slabs in that order (smallest terms first), which the reduce kernels sum and finish.  One
// launch per Linear; 128-row tiles and K slices fill the chip at any prompt length.
template <int NS>
__global__ __launch_bounds__(256, 1) void prefill_gemm_wp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                                 const float* __restrict__ bias, void* __restrict__ C, int M, int N,
                                                                 int K, int ldc, unsigned tp, int n_tiles, const PrefillQkv qa) {
    const int n_sp = (int)((tp >> 12) & 0xfffu), y = (int)blockIdx.y;
    // dispatched heaviest first (w_hi with three activation planes), slabs ordered smallest terms first
    const int sched = y / n_sp, sp = y - sched * n_sp, slab = (2 - sched) * n_sp + sp;
    const size_t plane = (size_t)N * K;
    if (sched == 0) prefill_gemm_body<PF_PARTIAL, NS, 3>(A, B, bias, C, M, N, K, ldc, tp, n_tiles, qa, sp, slab);
    else if (sched == 1) prefill_gemm_body<PF_PARTIAL, NS, 2>(A, B + plane, bias, C, M, N, K, ldc, tp, n_tiles, qa, sp, slab);
    else prefill_gemm_body<PF_PARTIAL, NS, 1>(A, B + 2 * plane, bias, C, M, N, K, ldc, tp, n_tiles, qa, sp, slab);
}

// Four columns of one row summed over the K-slice slabs, in slab order.  The loads go out four slabs at a time before the
// first addition: a loop that adds as it loads waits for every slab's round trip in turn (a one-prompt residual Linear has 4-5
// slabs, an fp32-weight one up to 12: the reduce kernel took 7-8 us of which the loads' latencies were most).
__device__ __forceinline__ f32x4 sum_slabs(const float* __restrict__ ws, int n_sp, int M, int N, int row, int col) {
    const size_t slab = (size_t)M * N;
    const float* p = ws + (size_t)row * N + col;
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    for (int sp0 = 1; sp0 < n_sp; sp0 += 4) {
        f32x4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(sp0 + u < n_sp ? sp0 + u : 0) * slab);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (sp0 + u < n_sp) a += t[u];
    }
    return a;
}

// Second half of a split-K GEMM: sum the slices in fixed order (deterministic), add bias, apply the epilogue.
template <int EPI>
__global__ __launch_bounds__(256) void prefill_reduce_kernel(const float* __restrict__ ws, int n_sp,
                                                             const float* __restrict__ bias, void* __restrict__ C, int M,
                                                             int N, int ldc, const PrefillQkv qa) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N / 4;
    if (i >= (size_t)M * n4) return;
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    f32x4 v = sum_slabs(ws, n_sp, M, N, m, n);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
    if (EPI == PF_GELU_SPLIT) {
        v.x = gelu_fast(v.x); v.y = gelu_fast(v.y); v.z = gelu_fast(v.z); v.w = gelu_fast(v.w);
        store_split4(reinterpret_cast<bf16_t*>(C) + (size_t)m * kSplit * N + n, N, v);
    } else {
        float* dst = reinterpret_cast<float*>(C) + (size_t)m * ldc + n;
        if (EPI == PF_RESID) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
        if (EPI == PF_QKV) qkv_cache_store(qa, m, n, v);
    }
}

// Split-K tail of a residual GEMM fused with the LayerNorm that follows it (main.zig:139-140 / the next
// Block's ln_1): one workgroup per row, a float4 (two beyond E = 1024) per thread — the slices are summed in
// fixed order, bias and the residual stream added, x stored, then the row it just produced is normalised and
// written as the three bf16 planes of the next GEMM.  Two launches (reduce, ln_split) become one.
__global__ __launch_bounds__(256) void prefill_reduce_resid_ln_kernel(const float* __restrict__ ws, int n_sp,
                                                                      const float* __restrict__ bias, float* __restrict__ x,
                                                                      int M, int N, const float* __restrict__ g,
                                                                      const float* __restrict__ bta, float eps,
                                                                      bf16_t* __restrict__ out) {
    __shared__ float s_red[8];
    const int row = blockIdx.x, tid = threadIdx.x, n4 = N >> 2;
    float* xr = x + (size_t)row * N;
    f32x4 v[2];
    float s = 0.0f, ss = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + 256 * j;
        v[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (i < n4) {
            const int e = i * 4;
            const f32x4 res = *reinterpret_cast<const f32x4*>(xr + e);
            const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + e) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 a = sum_slabs(ws, n_sp, M, N, row, e);
            if (bias) a += bv;
            a += res;
            *reinterpret_cast<f32x4*>(xr + e) = a;
            v[j] = a;
            s += (a.x + a.y) + (a.z + a.w);
            ss += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        }
    }
    s = wave_allsum(s);
    ss = wave_allsum(ss);
    if ((tid & 63) == 0) {
        s_red[(tid >> 6) * 2] = s;
        s_red[(tid >> 6) * 2 + 1] = ss;
    }
    __syncthreads();
    s = (s_red[0] + s_red[2]) + (s_red[4] + s_red[6]);
    ss = (s_red[1] + s_red[3]) + (s_red[5] + s_red[7]);
    const float mean = s / (float)N;
    const float sd = sqrtf(ss / (float)N - mean * mean + eps);
    bf16_t* hi = out + (size_t)row * kSplit * N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + 256 * j;
        if (i < n4) {
            const int e = i * 4;
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + e);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bta + e);
            f32x4 o;
            o.x = (v[j].x - mean) / sd * gg.x + bb.x;
            o.y = (v[j].y - mean) / sd * gg.y + bb.y;
            o.z = (v[j].z - mean) / sd * gg.z + bb.z;
            o.w = (v[j].w - mean) / sd * gg.w + bb.w;
            store_split4(hi + e, N, o);
        }
    }
}

// Rows of the XCD grid for a tiles_m x tiles_n launch (prefill_gemm_body): the divisor pair that makes an XCD's L2 fetch the
// fewest bytes — a_tile / w_tile = bytes of one row tile of the activation planes / one column tile of the weights; 0 = no pair
// divides both tile counts (contiguous ranges then).
static unsigned xcd_grid_rows(int tiles_m, int tiles_n, size_t a_tile, size_t w_tile) {
    unsigned best = 0;
    size_t best_cost = 0;
    for (int gr = 1; gr <= 8; gr *= 2) {
        const int gc = 8 / gr;
        if (tiles_m % gr != 0 || tiles_n % gc != 0) continue;
        const size_t cost = (size_t)(tiles_m / gr) * a_tile + (size_t)(tiles_n / gc) * w_tile;
        if (best == 0 || cost < best_cost) {
            best = (unsigned)gr;
            best_cost = cost;
        }
    }
    return best;
}

template <int EPI, int NS, int NSPL>
int launch_prefill_gemm_np(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                           float* ws, size_t ws_floats, const PrefillLn* ln, const PrefillQkv& qa, hipStream_t s) {
    constexpr int nsplit = NSPL;
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_gemm_kernel<EPI, NS, NSPL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(NS)));
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_gemm_kernel<PF_PARTIAL, NS, NSPL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(NS)));
        raised = true;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN * NS - 1) / (BN * NS), tiles = tiles_m * tiles_n;
    // Few output tiles (N = n_embed, or a short prompt): slice K so that about one workgroup per CU exists
    // (the LDS ring allows one), at least 3 K-steps per slice, partial sums through the workspace.
    const int nt = K / BK;
    int n_sp = 256 / tiles;
    if (n_sp > nt / 3) n_sp = nt / 3;
    if (n_sp < 1) n_sp = 1;
    while (n_sp > 1 && (size_t)n_sp * M * N > ws_floats) --n_sp;
    const unsigned xg = xcd_grid_rows(tiles_m, tiles_n, (size_t)BM * (K / (n_sp > 1 && ws ? n_sp : 1)) * 2 * nsplit,
                                      (size_t)BN * NS * (K / (n_sp > 1 && ws ? n_sp : 1)) * 2);
    if (n_sp <= 1 || !ws) {
        hipLaunchKernelGGL((prefill_gemm_kernel<EPI, NS, NSPL>), dim3(tiles), dim3(256), lds_bytes(NS), s, A, B, bias, C, M, N, K,
                           ldc, (unsigned)tiles_n | (1u << 12) | (xg << 24), tiles, qa);
    } else {
        hipLaunchKernelGGL((prefill_gemm_kernel<PF_PARTIAL, NS, NSPL>), dim3(tiles, n_sp), dim3(256), lds_bytes(NS), s, A, B, bias,
                           (void*)ws, M, N, K, ldc, (unsigned)tiles_n | ((unsigned)n_sp << 12) | (xg << 24), tiles, qa);
        if (EPI == PF_RESID && ln && ldc == N && N <= 2048) {
            hipLaunchKernelGGL(prefill_reduce_resid_ln_kernel, dim3(M), dim3(256), 0, s, ws, n_sp, bias,
                               reinterpret_cast<float*>(C), M, N, ln->g, ln->b, ln->eps, ln->out);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        }
        const size_t n = (size_t)M * (N / 4);
        hipLaunchKernelGGL((prefill_reduce_kernel<EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, n_sp, bias, C,
                           M, N, ldc, qa);
    }
    ZG_HIP(hipGetLastError());
    if (EPI == PF_RESID && ln) return launch_ln_split(reinterpret_cast<const float*>(C), M, N, ln->g, ln->b, ln->eps, ln->out, s);
    return ZG_OK;
}

// fp32 weights (nsplit = kWeightPlanes): prefill_gemm_wp_kernel + the reduce kernels (which apply the epilogue)
template <int EPI, int NS>
int launch_prefill_gemm_wp(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc, float* ws,
                           size_t ws_floats, const PrefillLn* ln, const PrefillQkv& qa, hipStream_t s) {
    static bool raised = false;
    if (!raised) {
        ZG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_gemm_wp_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(NS)));
        raised = true;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN * NS - 1) / (BN * NS), tiles = tiles_m * tiles_n;
    const int nt = K / BK;
    int n_sp = (256 + 3 * tiles - 1) / (3 * tiles);  // three passes per slice: about one workgroup per CU in all
    if (n_sp > nt / 3) n_sp = nt / 3;
    if (n_sp < 1) n_sp = 1;
    while (n_sp > 1 && (size_t)3 * n_sp * M * N > ws_floats) --n_sp;
    ZG_REQUIRE(ws && (size_t)3 * n_sp * M * N <= ws_floats, ZG_ERR_ARG, "prefill GEMM (fp32 weights): workspace of %zu floats for %d x %d", ws_floats, M, N);
    const unsigned xg = xcd_grid_rows(tiles_m, tiles_n, (size_t)BM * (K / n_sp) * 2 * 3, (size_t)BN * NS * (K / n_sp) * 2);
    hipLaunchKernelGGL((prefill_gemm_wp_kernel<NS>), dim3(tiles, 3 * n_sp), dim3(256), lds_bytes(NS), s, A, B, bias, (void*)ws, M, N, K, ldc,
                       (unsigned)tiles_n | ((unsigned)n_sp << 12) | (xg << 24), tiles, qa);
    if (EPI == PF_RESID && ln && ldc == N && N <= 2048) {
        hipLaunchKernelGGL(prefill_reduce_resid_ln_kernel, dim3(M), dim3(256), 0, s, ws, 3 * n_sp, bias, reinterpret_cast<float*>(C), M, N, ln->g,
                           ln->b, ln->eps, ln->out);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    const size_t n = (size_t)M * (N / 4);
    hipLaunchKernelGGL((prefill_reduce_kernel<EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, 3 * n_sp, bias, C, M, N, ldc, qa);
    ZG_HIP(hipGetLastError());
    if (EPI == PF_RESID && ln) return launch_ln_split(reinterpret_cast<const float*>(C), M, N, ln->g, ln->b, ln->eps, ln->out, s);
    return ZG_OK;
}

template <int EPI, int NS>
int launch_prefill_gemm_ns(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                           float* ws, size_t ws_floats, const PrefillLn* ln, const PrefillQkv& qa, int nsplit, hipStream_t s) {
    if (nsplit == kWeightPlanes) return launch_prefill_gemm_wp<EPI, NS>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, ln, qa, s);
    if (nsplit == 2) return launch_prefill_gemm_np<EPI, NS, 2>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, ln, qa, s);
    return launch_prefill_gemm_np<EPI, NS, kSplit>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, ln, qa, s);
}

// 128 x 256 tiles (the staged weight tile is shared by the three planes, so widening N is the cheap direction:
// 157 FLOP per staged byte against 96) once there are enough rows to fill the chip with them, else 128 x 128.
template <int EPI>
int launch_prefill_gemm_t(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc,
                          float* ws, size_t ws_floats, const PrefillLn* ln, const PrefillQkv& qa, int nsplit, hipStream_t s) {
    const int wide_tiles = ((M + BM - 1) / BM) * ((N + 2 * BN - 1) / (2 * BN));
    const bool wide = wide_tiles >= 256 && N >= 2 * BN;
    if (wide) return launch_prefill_gemm_ns<EPI, 2>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, ln, qa, nsplit, s);
    return launch_prefill_gemm_ns<EPI, 1>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, ln, qa, nsplit, s);
}

}  // namespace

int launch_embed_prefill(const int* tokens, int token_stride, int B, int P, const void* wte, const void* wpe,
                         int weight_type, int E, float* x, hipStream_t s) {
    hipLaunchKernelGGL(embed_prefill_kernel, dim3(B * P), dim3(256), 0, s, tokens, token_stride, P, wte, wpe, weight_type,
                       E, x);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_ln_split(const float* x, int M, int E, const float* g, const float* b, float eps, bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(ln_split_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, M, E, g, b, eps, out);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// Large prompts (several sequences): the Linears run on the persistent four-wave kernel of gemm_s4.hip with the activation
// planes as plane pairs of ONE K loop (K_eff = 3 K between two epilogues) once its 256 x 192 tiles fill most of the chip; the
// N = n_embed Linears (residual adds) get there by slicing K over the tile list, their partial slabs summed by the reduce
// kernels above in fixed order (which also apply bias, residual and the LayerNorm + split that follows).  Returns the slice
// count, 0 = the 128-row kernels of this file take the launch.
static int g_force_kernel = 0, g_force_slices = 0;  // zg_debug_prefill_route / _linear: 1 = gemm_s4, 2 = the 128-row kernels; K slices
void prefill_force_route(int kernel, int slices) {
    g_force_kernel = kernel;
    g_force_slices = slices;
}
static int s4_route(int M, int N, int K, int epi, size_t ws_floats, bool have_ws) {
    int min_tiles = 192;  // tiles of 256 x 192 (x K slices) from which the persistent kernel takes the launch: three quarters of the CUs
    if (g_force_kernel == 1) min_tiles = 1;
    if (g_force_kernel >= 16) min_tiles = g_force_kernel;  // (measurement: another threshold)
    if (g_force_kernel == 2) return 0;
    const int kpp = K / 64;
    if (K % 64 != 0 || kpp < 2 || kSplit * K >= 65536) return 0;
    const long tiles = (long)((M + 255) / 256) * ((N + 191) / 192);
    if (epi != PF_RESID) return tiles >= min_tiles ? 1 : 0;
    if (!have_ws) return 0;
    // (at most four slices: a one-prompt mlp c_proj cut into twelve — 16 tiles — ran 21.8 + 8.0 us of GEMM + reduce against 15.5 + 6.8
    // on the 128-row kernel: profiles/round5_prefill_1x1023_kernel_stats.md of the first pass).  Among the slice counts that divide
    // the K-steps: the one whose items fill most of ONE round of 256 workgroups (four prompts: 64 tiles x 4 rather than x 3); more
    // tiles than that run unsliced.
    if (g_force_slices > 0) return (kpp % g_force_slices == 0 && g_force_slices <= kpp / 2 && (size_t)g_force_slices * M * N <= ws_floats) ? g_force_slices : 0;
    int best = 0;
    long best_items = 0;
    for (int n_sl = 1; n_sl <= 4 && n_sl <= kpp / 2; ++n_sl) {
        if (kpp % n_sl != 0 || (size_t)n_sl * M * N > ws_floats) continue;
        const long items = tiles * n_sl;
        if (items < min_tiles) continue;
        if (best == 0 || (items <= 256 && items > best_items)) {
            best = n_sl;
            best_items = items;
        }
    }
    return best;
}

int launch_prefill_gemm(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, int K, int ldc, int epi,
                        float* ws, size_t ws_floats, const PrefillLn* ln, hipStream_t s, const PrefillQkv* qkv, int nsplit) {
    ZG_REQUIRE(M > 0 && N >= 64 && N % 64 == 0 && K >= 64 && K % 64 == 0, ZG_ERR_UNSUPPORTED, "prefill gemm: M=%d N=%d K=%d", M, N, K);
    {
        const int n_sl = s4_route(M, N, K, epi, ws_floats, ws != nullptr);
        if (n_sl > 0 && epi == PF_RESID && ldc == N) {
            ZG_TRY(launch_gemm_s4_prefill(A, B, nullptr, ws, M, N, K, nsplit, S4_PARTIAL, n_sl, nullptr, s));
            if (ln && N <= 2048) {
                hipLaunchKernelGGL(prefill_reduce_resid_ln_kernel, dim3(M), dim3(256), 0, s, ws, n_sl, bias, reinterpret_cast<float*>(C), M, N, ln->g,
                                   ln->b, ln->eps, ln->out);
                ZG_HIP(hipGetLastError());
                return ZG_OK;
            }
            const size_t n = (size_t)M * (N / 4);
            hipLaunchKernelGGL((prefill_reduce_kernel<PF_RESID>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, n_sl, bias, C, M, N, ldc,
                               PrefillQkv{});
            ZG_HIP(hipGetLastError());
            if (ln) return launch_ln_split(reinterpret_cast<const float*>(C), M, N, ln->g, ln->b, ln->eps, ln->out, s);
            return ZG_OK;
        }
        if (n_sl == 1 && epi == PF_GELU_SPLIT) return launch_gemm_s4_prefill(A, B, bias, C, M, N, K, nsplit, S4_SPLIT3, 1, nullptr, s);
        if (n_sl == 1 && epi == PF_QKV && qkv && ldc == N) return launch_gemm_s4_prefill(A, B, bias, C, M, N, K, nsplit, S4_QKV, 1, qkv, s);
    }
    // (the operands are addressed through 32-bit buffer descriptors; a piece is switched off by adding 2 GiB to its scalar offset,
    // which must then lie past the descriptor's end without wrapping: operands < 2 GiB.  That a raw buffer access is range-checked
    // on voffset + soffset — LLVM documents only voffset — is pinned on the hardware by tools/microbench/soffset_bounds_probe.hip,
    // tests/test_hw_rules_gpu.py)
    ZG_REQUIRE((size_t)M * kSplit * K * 2 < ((size_t)1 << 31) && (size_t)N * K * 2 < ((size_t)1 << 31), ZG_ERR_UNSUPPORTED,
               "prefill gemm: operands of %d x %d x %d beyond 2 GiB", M, N, K);
    ZG_REQUIRE(nsplit == 2 || nsplit == kSplit || nsplit == kWeightPlanes, ZG_ERR_ARG, "prefill gemm: %d activation planes", nsplit);
    const PrefillQkv none{};
    switch (epi) {
        case PF_F32: return launch_prefill_gemm_t<PF_F32>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, nullptr, none, nsplit, s);
        case PF_RESID: return launch_prefill_gemm_t<PF_RESID>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, ln, none, nsplit, s);
        case PF_GELU_SPLIT: return launch_prefill_gemm_t<PF_GELU_SPLIT>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, nullptr, none, nsplit, s);
        case PF_QKV:
            ZG_REQUIRE(qkv && N == 3 * qkv->E && ldc == N, ZG_ERR_ARG, "prefill gemm: PF_QKV needs the cache description");
            return launch_prefill_gemm_t<PF_QKV>(A, B, bias, C, M, N, K, ldc, ws, ws_floats, nullptr, *qkv, nsplit, s);
    }
    ZG_REQUIRE(false, ZG_ERR_ARG, "prefill gemm: epilogue %d", epi);
}

#ifdef ZG_PF_STAMPS
}  // namespace zg
extern "C" int zg_debug_prefill_stamps(unsigned long long* out, size_t n) {
    unsigned long long h[32];
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(zg::g_pf_stamps), sizeof h) != hipSuccess) return -1;
    for (size_t i = 0; i < n && i < 32; ++i) out[i] = h[i];
    return 0;
}
namespace zg {
#endif
}  // namespace zg
