// elementwise.hip — the small ops of reference src/ops.zig as standalone HIP kernels (op tier),
// plus the decode-step head kernel of the model tier.  All are HBM/latency bound; loads are
// 16 B per lane where the layout allows.
#include <algorithm>

#include "zg_kernels.h"

namespace zg {

namespace {

__device__ __forceinline__ float block_allsum(float v, float* s_red) {
    v = wave_allsum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < nw; ++w) t += s_red[w];
    return t;
}

__device__ __forceinline__ float block_allmax(float v, float* s_red) {
    v = wave_allmax(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float t = s_red[0];
    for (int w = 1; w < nw; ++w) t = fmaxf(t, s_red[w]);
    return t;
}

// A single-workgroup kernel of the op tier may announce its own completion: every lane fences its stores to the system, then
// behind a workgroup barrier (waves that have left no longer count) one lane stores the call's sequence number into the pinned
// completion word the host polls (Call::finish) — the separate one-thread launch (2.7 us on the device) is then skipped.
struct DoneWord {
    unsigned* flag;  // nullptr: not asked
    unsigned seq;
};
__device__ inline void announce_done(const DoneWord d) {
    if (d.flag == nullptr) return;
    // every wave waits until its own stores have been acknowledged by the L2 (the workgroup's waves share one XCD's L2), then the
    // barrier, then ONE lane releases to system scope — the L2-wide write-back covers all waves' stores — and stores the word.
    // (A system-scope fence in every lane was measured: +1.3 us on LayerNorm, +4.8 us on gelu — one L2 write-back per wave.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// LayerNorm.forward (src/ops.zig:82-104): one wave per row, in place.  Rows of up to 64 x 32 elements are held in registers
// between the statistics and the normalisation: every element is read once and written once (the op tier hands small host
// buffers over in place, across PCIe).
__global__ __launch_bounds__(256) void layernorm_kernel(float* x, int rows, int n, const float* g,
                                                        const float* b, float eps, DoneWord done, float* shadow) {
    // shadow (op tier, may be null): a device-memory twin of the result — x itself may be pinned host memory the NEXT op's
    // kernels should not read across PCIe (api_ops.hip: the output of one op is the input of the next Linear in src/main.zig)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < rows) {
        float* r = x + (size_t)row * n;
        float s1 = 0.0f, s2 = 0.0f;
        if (n <= 64 * 32) {
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int i = lane + 64 * j;
                v[j] = i < n ? r[i] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                s1 += v[j];
                s2 = fmaf(v[j], v[j], s2);
            }
            s1 = wave_allsum(s1);
            s2 = wave_allsum(s2);
            const float mean = s1 / (float)n;
            const float std_ = sqrtf(s2 / (float)n - mean * mean + eps);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int i = lane + 64 * j;
                if (i < n) {
                    const float y = (v[j] - mean) / std_ * g[i] + b[i];
                    r[i] = y;
                    if (shadow) shadow[(size_t)row * n + i] = y;
                }
            }
        } else {
            for (int i = lane; i < n; i += 64) {
                const float v = r[i];
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
            s1 = wave_allsum(s1);
            s2 = wave_allsum(s2);
            const float mean = s1 / (float)n;
            const float std_ = sqrtf(s2 / (float)n - mean * mean + eps);
            for (int i = lane; i < n; i += 64) {
                const float y = (r[i] - mean) / std_ * g[i] + b[i];
                r[i] = y;
                if (shadow) shadow[(size_t)row * n + i] = y;
            }
        }
    }
    announce_done(done);  // (asked for only when the grid is one workgroup)
}

// gelu (src/ops.zig:221-228), in place.
__global__ __launch_bounds__(1024) void gelu_kernel(float* x, size_t n, DoneWord done, float* shadow) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    f32x4* x4 = reinterpret_cast<f32x4*>(x);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v = x4[i];
        v.x = gelu_ref(v.x); v.y = gelu_ref(v.y); v.z = gelu_ref(v.z); v.w = gelu_ref(v.w);
        x4[i] = v;
        if (shadow) reinterpret_cast<f32x4*>(shadow)[i] = v;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float y = gelu_ref(x[i]);
        x[i] = y;
        if (shadow) shadow[i] = y;
    }
    announce_done(done);  // (asked for only when the grid is one workgroup)
}

// softmax (src/ops.zig:231-241): the whole slice is one vector; single workgroup, in place.  Up to 1024 x 64 elements (the
// logits of every GPT-2 vocabulary, src/main.zig:203) stay in registers across the three passes: read once, written once.
__global__ __launch_bounds__(1024) void softmax_kernel(float* x, size_t n, DoneWord done) {
    __shared__ float s_red[16];
    if (n <= (size_t)1024 * 64) {
        float v[64];
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const size_t i = threadIdx.x + (size_t)1024 * j;
            v[j] = i < n ? x[i] : -3.0e38f;
            mx = fmaxf(mx, v[j]);
        }
        mx = block_allmax(mx, s_red);
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const size_t i = threadIdx.x + (size_t)1024 * j;
            v[j] = i < n ? __expf(v[j] - mx) : 0.0f;
            sum += v[j];
        }
        sum = block_allsum(sum, s_red);
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const size_t i = threadIdx.x + (size_t)1024 * j;
            if (i < n) x[i] = v[j] / sum;
        }
        announce_done(done);
        return;
    }
    float mx = -3.0e38f;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, x[i]);
    mx = block_allmax(mx, s_red);
    float sum = 0.0f;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = __expf(x[i] - mx);
        x[i] = e;
        sum += e;
    }
    sum = block_allsum(sum, s_red);
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) x[i] = x[i] / sum;
    announce_done(done);
}

// Embedding.forward (src/ops.zig:59-67): out[i] = weight[idx[i]].
__global__ __launch_bounds__(256) void embedding_kernel(const float* w, size_t emb_dim, const size_t* idx,
                                                        size_t n_rows, float* out, int* oob, DoneWord done) {
    const size_t i = blockIdx.x;
    const size_t id = idx[i];
    if (id >= n_rows) {
        if (threadIdx.x == 0) *oob = 1;
    } else {
        for (size_t e = threadIdx.x; e < emb_dim; e += blockDim.x) out[i * emb_dim + e] = w[id * emb_dim + e];
    }
    announce_done(done);  // (asked for only when the grid is one workgroup)
}

// split_qkv (src/ops.zig:177-196): rows x [3E] -> rows x [E].
__global__ __launch_bounds__(256) void split_qkv_kernel(const float* in, size_t n_embed, size_t split_idx,
                                                        float* out) {
    const size_t r = blockIdx.x;
    for (size_t e = threadIdx.x; e < n_embed; e += blockDim.x)
        out[r * n_embed + e] = in[r * 3 * n_embed + split_idx * n_embed + e];
}

// transpose (src/ops.zig:199-216): (b,t,n,h) -> (b,n,t,h); one workgroup per (b,n,t) run of h.
__global__ __launch_bounds__(64) void transpose_kernel(const float* in, size_t t, size_t n, size_t h, float* out) {
    const size_t s = blockIdx.x, hh = blockIdx.y, b = blockIdx.z;
    const float* src = in + b * t * n * h + s * n * h + hh * h;
    float* dst = out + b * t * n * h + hh * t * h + s * h;
    for (size_t d = threadIdx.x; d < h; d += blockDim.x) dst[d] = src[d];
}

__global__ __launch_bounds__(256) void copy_kernel(const float* in, float* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* in, bf16_t* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = f32_to_bf16_rne(in[i]);
}

// fp32 [rows][K] -> bf16 planes [rows][3K] = [hi | mid | lo] with hi + mid + lo == x exactly (8 + 8 + 8 mantissa bits):
// the operand format of launch_gemm_planes.  One thread per 4 consecutive elements (16-B load, three 8-B stores).
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t rows, int K) {
    const size_t quads = rows * (size_t)(K / 4), stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) {
        const size_t r = i / (K / 4), k = (i % (K / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + r * K + k);
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pk(v.x, v.y, h0, m0, l0);
        split3_pk(v.z, v.w, h1, m1, l1);
        bf16_t* o = out + r * (size_t)(3 * K) + k;
        *reinterpret_cast<u32x2*>(o) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(o + K) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(o + 2 * K) = u32x2{l0, l1};
    }
}

__device__ __forceinline__ float load_emb(const void* w, int wt, size_t off) {
    if (wt == WT_BF16) return __uint_as_float((uint32_t)reinterpret_cast<const bf16_t*>(w)[off] << 16);
    return reinterpret_cast<const float*>(w)[off];
}

// Head of one decode step (single workgroup).  Follows generate/sample of src/main.zig:322-342
// with greedy argmax: picks the token fed at position s, records outputs, writes
// x = wte[token] + wpe[s] (GPT.forward, src/main.zig:179-183), publishes seq_len = s + 1 and
// advances the step counter so that the same captured graph serves every position.
// Latency-bound: the argmax partials of the previous step are fetched speculatively together with
// the control words (one memory round trip), one wave per sequence; only the wte row depends on
// the chosen token (second round trip).
__global__ __launch_bounds__(512) void embed_step_kernel(const EmbedArgs a) {
    __shared__ int s_tok[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = a.ctrl->step;
    const int mode = a.ctrl->mode;
    const int npart = a.n_partials;
    const int n_waves = a.batch > 4 ? 8 : 4;  // as launched (blockDim is a scalar load from the kernarg segment)
    for (int b = wave; b < a.batch; b += n_waves) {
        // speculative fetch of this sequence's partial maxima (valid memory whether or not needed)
        float bv = -3.0e38f;
        int bi = 0x7fffffff;
        // eight partials per lane per round, loads first and branch-free (clamped): a load-compare loop is one dependent memory
        // round trip per 64 partials — seven of them for the 393 partials of 124M, most of this kernel's time
        for (int base = 0; base < npart; base += 512) {
            float pv[8];
            int pi[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = min(base + lane + 64 * j, npart - 1);
                pv[j] = a.part_val[(size_t)b * a.part_stride + p];
                pi[j] = a.part_idx[(size_t)b * a.part_stride + p];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // (a clamped duplicate of the last partial changes nothing)
                if (pv[j] > bv || (pv[j] == bv && pi[j] < bi)) { bv = pv[j]; bi = pi[j]; }
            }
        }
        {   // the argument-block fields used behind the partial-maxima loads (zg_common.h ZG_PIN)
            ZG_PIN(a.forced); ZG_PIN(a.wte); ZG_PIN(a.wpe); ZG_PIN(a.weight_type); ZG_PIN(a.n_embed); ZG_PIN(a.cur_token); ZG_PIN(a.out_tokens);
            ZG_PIN(a.out_stride); ZG_PIN(a.x); ZG_PIN(a.pl_out); ZG_PIN(a.pl_g); ZG_PIN(a.epoch); ZG_PIN(a.finish_only); ZG_PIN(a.st_out); ZG_PIN(a.vocab);
        }
        if (b == 0 && a.progress != nullptr && a.finish_only == 0 && lane == 0) {
            // a step at sequence length s + 1 starts; block 0 of every launch of this queue lands on the XCD this block is on
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID
            const unsigned long long word = ((unsigned long long)(0x100u | xcc) << 32) | (((unsigned)(s + 1) << 8) | 1u);
            __hip_atomic_store(static_cast<unsigned long long*>(__builtin_assume_aligned(a.progress, 8)), word, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
#ifdef ZG_STAMPS
            reinterpret_cast<PfCtl*>(a.progress)->xcd_log[0] = 0x100u | xcc;
#endif
        }
        const int np = a.prompt_len ? a.prompt_len[b] : 0;
        const int p_cur = a.prompt[(size_t)b * a.prompt_stride + min(s, a.prompt_stride - 1)];
        const int p_last = a.prompt[(size_t)b * a.prompt_stride + max(min(np, a.prompt_stride) - 1, 0)];
        const int forced = a.forced[b];
        const int drawn = a.sampled[b];  // (mode 2; always a readable address)
        // argmax over the wave: DPP row rotations inside the 16-lane rows, then the four row winners through v_readlane
        // (six rounds of ds_bpermute pairs were ~0.3 us of dependent LDS-crossbar hops)
#define ZG_AM_STEP(N)                                                                                                   \
    {                                                                                                                   \
        const float ov = dpp_row_ror<N>(bv);                                                                            \
        const int oi = __builtin_amdgcn_update_dpp(0, bi, 0x120 | N, 0xF, 0xF, true);                                   \
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }                                                     \
    }
        ZG_AM_STEP(8) ZG_AM_STEP(4) ZG_AM_STEP(2) ZG_AM_STEP(1)
#undef ZG_AM_STEP
        {   // every lane folds in the four row winners (its own among them): all lanes end with the wave's pick
            const float rv = bv;
            const int ri = bi;
#pragma unroll
            for (int r = 0; r < 64; r += 16) {
                const float ov = lane_value(rv, r);
                const int oi = __builtin_amdgcn_readlane(ri, r);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
        }
        // logits that are all NaN compare false everywhere and leave the start value standing: index 0 then, as a loop that starts at
        // logits[0] and keeps what compares greater (the oracle, and the reference's order of comparison) — never an index past
        // the vocabulary, which the gather below would follow out of the table (a NaN in a checkpoint must not fault the GPU)
        int g = (unsigned)bi < (unsigned)a.vocab ? bi : 0;
        if (mode == 2 && a.finish_only != 2) g = (unsigned)drawn < (unsigned)a.vocab ? drawn : 0;  // the sampler's draw takes the pick's place
        // the pick of step s-1 exists iff that step ran lm_head (main.zig:337)
        const bool have_pick = a.finish_only == 2 || ((mode != 1) && (s >= 1) && (s - 1 >= np));
        if (lane == 0) {
            if (have_pick) {
                if (a.finish_only == 2) a.cur_token[b] = g;  // zg_gpt_argmax
                else a.out_tokens[(size_t)b * a.out_stride + (s - 1)] = g;
            }
            if (a.finish_only == 0 || a.finish_only == 3) {
                int tok;
                if (mode == 1) tok = forced;
                else if (s < np) tok = p_cur;      // main.zig:331-334: prompt token s
                else if (s == np) tok = p_last;    // main.zig:337: the previous token is fed again
                else tok = g;
                if (mode != 1 && s < np) a.out_tokens[(size_t)b * a.out_stride + s] = tok;
                a.cur_token[b] = tok;
                s_tok[b] = tok;
            }
        }
    }
    if (a.finish_only == 1 || a.finish_only == 2) return;
    __syncthreads();
    const int total4 = a.batch * (a.n_embed >> 2);
    for (int i = threadIdx.x; i < total4; i += 64 * n_waves) {
        const int b = i / (a.n_embed >> 2), e = (i % (a.n_embed >> 2)) * 4;
        const int tok = s_tok[b];
        f32x4 o;
        if (a.weight_type == WT_BF16) {
            const u32x2 t = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(a.wte) + (size_t)tok * a.n_embed + e);
            const u32x2 p = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(a.wpe) + (size_t)s * a.n_embed + e);
            o.x = bf16_lo(t.x) + bf16_lo(p.x);
            o.y = bf16_hi(t.x) + bf16_hi(p.x);
            o.z = bf16_lo(t.y) + bf16_lo(p.y);
            o.w = bf16_hi(t.y) + bf16_hi(p.y);
        } else {
            const f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.wte) + (size_t)tok * a.n_embed + e);
            const f32x4 p = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.wpe) + (size_t)s * a.n_embed + e);
            o = t + p;
        }
        *reinterpret_cast<f32x4*>(a.x + (size_t)b * a.n_embed + e) = o;
        if (a.pl_out) {  // the first Linear of the lock-step batch reads its input as planes of g * x
            const f32x4 v = o * *reinterpret_cast<const f32x4*>(a.pl_g + e);
            uint32_t h0, m0, l0, h1, m1, l1;
            split3_pk(v.x, v.y, h0, m0, l0);
            split3_pk(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<u32x2*>(a.pl_out + plane_elem(0, b, e)) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(a.pl_out + plane_elem(1, b, e)) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(a.pl_out + plane_elem(2, b, e)) = u32x2{l0, l1};
        }
        if (a.st_out) {  // LayerNorm statistics by 16-column tile: four consecutive threads hold one tile of one row
            float s1 = (o.x + o.y) + (o.z + o.w);
            float s2 = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, o.w * o.w)));
            s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
            s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
            if ((threadIdx.x & 3) == 0) *reinterpret_cast<float2*>(a.st_out + ((size_t)b * (a.n_embed >> 4) + (e >> 4)) * 2) = float2{s1, s2};
        }
    }
    if (threadIdx.x == 0 && a.epoch) *a.epoch += 1u;  // a step starts: new tags for its hand-overs
    if (threadIdx.x == 0 && a.finish_only == 0) {
        a.ctrl->seq_len = s + 1;
        a.ctrl->step = s + 1;
    }
}

// softmax(x / temp) and the weighted draw of GPT.sample (src/main.zig:200-206; std.rand weightedIndex: the first index whose
// running sum exceeds u x total), in three small kernels behind lm_head.  History: one workgroup per row with a contiguous chunk of
// 50 logits per lane and a 1024-step serial scan (~100 us per row at V = 50257); then one workgroup with coalesced passes and
// segment sums (28 us: a single CU has ~100 cache lines in flight against ~2 us of memory-side latency for 1600 lines of logits
// that another XCD's lm_head just wrote).  Now:
//   sample_seg_kernel   many workgroups: the row maximum comes from lm_head's argmax partials (no pass over the logits), every
//                       wave sums e = exp(x / temp - max) over SEGMENTS of 64 consecutive elements (786 at V = 50257)
//   sample_pick_kernel  one wave per row walks the segment sums IN INDEX ORDER (64 lanes x consecutive runs of segments, a scan
//                       over the lanes, the owning lane walks its run), then scans the 64 elements of the segment that holds
//                       u x total lane by lane.  Sums inside a segment and over lanes are tree-shaped: the draw can differ from a
//                       strictly left-to-right sum only where u x total lies within rounding (~1e-7) of a boundary.
//   sample_probs_kernel (only when the caller wants them) leaves the probabilities in the logits row, as the reference does.
// NaN probabilities (no interval holds the point) give the last index.  vocab <= 64 x 4096.
constexpr int kSegMax = 4096;  // segment sums per row; [kSegMax] = the row maximum / temp, [kSegMax + 1] = the total

__global__ __launch_bounds__(256) void sample_seg_kernel(const float* __restrict__ logits, int vocab, const SampleParams* params, float inv_temp_arg,
                                                         const float* __restrict__ part_val, int n_part, int part_stride, float* __restrict__ seg_out) {
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const float inv_temp = params ? params->inv_temp : inv_temp_arg;
    const float* x = logits + (size_t)b * vocab;
    float* so = seg_out + (size_t)b * (kSegMax + 2);
    float mxr = -3.0e38f;  // max of the row = max of the per-workgroup maxima lm_head's argmax epilogue left (EPI_ARGMAX)
    for (int p = tid; p < n_part; p += 256) mxr = fmaxf(mxr, part_val[(size_t)b * part_stride + p]);
    mxr = wave_allmax(mxr);
    if (lane == 0) s_red[wave] = mxr;
    __syncthreads();
    const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])) * inv_temp;
    const int nseg = (vocab + 63) >> 6;
    for (int sg = blockIdx.x * 4 + wave; sg < nseg; sg += gridDim.x * 4) {
        const int i = sg * 64 + lane;
        const float e = i < vocab ? __expf(x[i] * inv_temp - mx) : 0.0f;
        const float t = wave_allsum(e);
        if (lane == 0) so[sg] = t;
    }
    if (blockIdx.x == 0 && tid == 0) so[kSegMax] = mx;
}

__global__ __launch_bounds__(64) void sample_pick_kernel(const float* __restrict__ logits, int vocab, const SampleParams* params, float inv_temp_arg,
                                                         const float* __restrict__ u_arr, const StepCtrl* ctrl, float* __restrict__ seg_io,
                                                         int* __restrict__ token_out) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const float inv_temp = params ? params->inv_temp : inv_temp_arg;
    const float* x = logits + (size_t)b * vocab;
    float* so = seg_io + (size_t)b * (kSegMax + 2);
    float u;
    if (u_arr) u = u_arr[b];
    else {  // the counter PRNG of (seed, sequence length of the step whose logits these are, sequence): as zg_gpt_sample derives it on the host
        unsigned long long z = params->seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)ctrl->seq_len * 0xD1B54A32D192ED03ULL + (unsigned long long)b + 1ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        u = (float)(unsigned)(z >> 40) * 5.9604644775390625e-08f;
    }
    const int nseg = (vocab + 63) >> 6;
    const int run = (nseg + 63) >> 6, s0 = lane * run, s1 = min(s0 + run, nseg);
    const float mx = so[kSegMax];
    float local = 0.0f;
    for (int k = s0; k < s1; ++k) local += so[k];
    float incl = local;  // inclusive scan over the lanes, in lane order
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    const float total = __shfl(incl, 63, 64);
    const float point = u * total;
    int pick = vocab - 1;
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(s0 < s1 && point < incl);
    if (hit != 0ull) {
        const int owner = __builtin_ctzll(hit);
        int sg = 0;
        float before = 0.0f;
        if (lane == owner) {  // the owning lane walks its run of segments
            before = incl - local;
            sg = s1 - 1;
            for (int k = s0; k < s1; ++k) {
                const float v = so[k];
                if (point < before + v) {
                    sg = k;
                    break;
                }
                before += v;
            }
        }
        sg = __shfl(sg, owner, 64);
        before = __shfl(before, owner, 64);
        const int i = sg * 64 + lane;
        float c = i < vocab ? __expf(x[i] * inv_temp - mx) : 0.0f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float up = __shfl_up(c, off, 64);
            if (lane >= off) c += up;
        }
        const unsigned long long in = __builtin_amdgcn_ballot_w64(i < vocab && point < before + c);
        // (rounding between the segment sum and its lane-by-lane scan can leave the point just past the last lane: that lane then)
        pick = sg * 64 + (in != 0ull ? __builtin_ctzll(in) : min(63, vocab - 1 - sg * 64));
    }
    if (lane == 0) {
        token_out[b] = pick;
        so[kSegMax + 1] = total;
    }
}

__global__ __launch_bounds__(256) void sample_probs_kernel(float* logits, int vocab, float inv_temp, const float* __restrict__ seg_io) {
    const int b = blockIdx.y;
    float* x = logits + (size_t)b * vocab;
    const float* so = seg_io + (size_t)b * (kSegMax + 2);
    const float mx = so[kSegMax], inv = 1.0f / so[kSegMax + 1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) x[i] = __expf(x[i] * inv_temp - mx) * inv;
}

inline int grid_for(size_t n, int block = 256, int cap = 2048) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}

}  // namespace

// done_flag != nullptr: the caller would like the kernel to announce its own completion; *announced tells whether this launch
// does (a one-workgroup grid), otherwise the caller's one-thread launch follows as usual.
int launch_layernorm(float* x, int rows, int n, const float* g, const float* b, float eps, hipStream_t s, unsigned* done_flag, unsigned done_seq,
                     bool* announced, float* shadow) {
    if (announced) *announced = false;
    if (rows == 0) return ZG_OK;
    const bool own = done_flag != nullptr && rows <= 4;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, n, g, b, eps, DoneWord{own ? done_flag : nullptr, done_seq}, shadow);
    ZG_HIP(hipGetLastError());
    if (announced) *announced = own;
    return ZG_OK;
}

int launch_gelu(float* x, size_t n, hipStream_t s, unsigned* done_flag, unsigned done_seq, bool* announced, float* shadow) {
    if (announced) *announced = false;
    if (n == 0) return ZG_OK;
    const bool own = done_flag != nullptr && n <= 4096;  // one workgroup of 1024 lanes, one vector of four each: a single pass
    hipLaunchKernelGGL(gelu_kernel, dim3(own ? 1 : grid_for(n / 4 + 1)), dim3(own ? 1024 : 256), 0, s, x, n, DoneWord{own ? done_flag : nullptr, done_seq}, shadow);
    ZG_HIP(hipGetLastError());
    if (announced) *announced = own;
    return ZG_OK;
}

int launch_softmax(float* x, size_t n, hipStream_t s, unsigned* done_flag, unsigned done_seq, bool* announced) {
    if (announced) *announced = false;
    if (n == 0) return ZG_OK;
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(1024), 0, s, x, n, DoneWord{done_flag, done_seq});
    ZG_HIP(hipGetLastError());
    if (announced) *announced = done_flag != nullptr;
    return ZG_OK;
}

// The completion word of an op-tier call: the host polls this pinned word instead of paying hipStreamSynchronize's wake-up
// (tools/microbench/sync_cost.hip: 9.6-11.0 against 12.0-13.3 us per one-kernel call).  In stream order behind the call's
// kernels, whose results the kernel boundary has already released to the system.
__global__ void done_flag_kernel(unsigned* flag, unsigned seq) {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The outputs of an op-tier call that live in device memory (read back by later kernels of the call, e.g. _qkv and _q of
// CausalSelfAttention.forward) leave for the pinned arena AND the call's completion is announced in ONE launch: a single
// workgroup copies up to four small segments and then stores the completion word — instead of a DMA plus the one-thread launch.
__global__ __launch_bounds__(1024) void copy_out_done_kernel(CopySegs segs, DoneWord done) {
    for (int k = 0; k < segs.n; ++k) {
        const f32x4* src = reinterpret_cast<const f32x4*>(segs.src[k]);
        f32x4* dst = reinterpret_cast<f32x4*>(segs.dst[k]);
        const unsigned n4 = segs.bytes[k] / 16;
        for (unsigned i = threadIdx.x; i < n4; i += 1024) dst[i] = src[i];
        const unsigned tail = segs.bytes[k] & 15u;  // (multiples of 4 bytes: the op tier's element types are 4 or 8 bytes wide)
        if (threadIdx.x < tail / 4) reinterpret_cast<float*>(segs.dst[k])[n4 * 4 + threadIdx.x] = reinterpret_cast<const float*>(segs.src[k])[n4 * 4 + threadIdx.x];
    }
    announce_done(done);
}
int launch_copy_out_done(const CopySegs& segs, unsigned* flag, unsigned seq, hipStream_t s) {
    hipLaunchKernelGGL(copy_out_done_kernel, dim3(1), dim3(1024), 0, s, segs, DoneWord{flag, seq});
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// two equally long copies in one launch (the k and v rows of a cache append, ops.zig:151-152, :156-157)
__global__ __launch_bounds__(256) void copy2_kernel(const float* a, float* da, const float* b, float* db, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        da[i] = a[i];
        db[i] = b[i];
    }
}
int launch_copy2_f32(const float* a, float* da, const float* b, float* db, size_t n, hipStream_t s) {
    if (n == 0) return ZG_OK;
    hipLaunchKernelGGL(copy2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, da, b, db, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_done_flag(unsigned* flag, unsigned seq, hipStream_t s) {
    hipLaunchKernelGGL(done_flag_kernel, dim3(1), dim3(1), 0, s, flag, seq);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

__global__ void epoch_bump_kernel(unsigned* epoch) { *epoch += 1u; }
int launch_epoch_bump(unsigned* epoch, hipStream_t s) {
    hipLaunchKernelGGL(epoch_bump_kernel, dim3(1), dim3(1), 0, s, epoch);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_embedding(const float* w, size_t emb_dim, const size_t* idx, size_t n_idx, size_t n_rows,
                     float* out, int* d_oob, hipStream_t s, unsigned* done_flag, unsigned done_seq, bool* announced) {
    if (announced) *announced = false;
    if (n_idx == 0) return ZG_OK;
    // (d_oob is a pinned host word: the caller reads it behind its own drain of the stream)
    const bool own = done_flag != nullptr && n_idx == 1;
    hipLaunchKernelGGL(embedding_kernel, dim3((unsigned)n_idx), dim3(256), 0, s, w, emb_dim, idx, n_rows, out, d_oob, DoneWord{own ? done_flag : nullptr, done_seq});
    ZG_HIP(hipGetLastError());
    if (announced) *announced = own;
    return ZG_OK;
}

int launch_split_qkv(const float* in, size_t rows, size_t n_embed, size_t split_idx, float* out, hipStream_t s) {
    if (rows == 0) return ZG_OK;
    hipLaunchKernelGGL(split_qkv_kernel, dim3((unsigned)rows), dim3(256), 0, s, in, n_embed, split_idx, out);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_transpose(const float* in, size_t batch, size_t t, size_t n, size_t h, float* out, hipStream_t s) {
    if (batch * t * n * h == 0) return ZG_OK;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)t, (unsigned)n, (unsigned)batch), dim3(64), 0, s, in, t, n, h, out);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_copy_f32(const float* in, float* out, size_t n, hipStream_t s) {
    if (n == 0) return ZG_OK;
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) {
    if (n == 0) return ZG_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static __global__ __launch_bounds__(256) void kv_clear_tail_kernel(char* base, size_t cache_stride, size_t plane_off, int row_bytes, int strips, int ctx,
                                                            int from_row) {
    const int strip = blockIdx.x, cache = blockIdx.y;
    char* p = base + (size_t)cache * cache_stride + plane_off + ((size_t)strip * ctx + from_row) * row_bytes;
    const size_t n16 = (size_t)(ctx - from_row) * row_bytes / 16;  // (row_bytes is a multiple of 64)
    for (size_t i = threadIdx.x; i < n16; i += 256) reinterpret_cast<u32x4*>(p)[i] = u32x4{0u, 0u, 0u, 0u};
}

int launch_kv_clear_tail(void* base, int n_caches, size_t cache_stride, size_t plane_off, int row_bytes, int strips, int ctx, int from_row,
                         hipStream_t s) {
    if (from_row >= ctx || n_caches == 0 || strips == 0) return ZG_OK;
    ZG_REQUIRE(row_bytes % 64 == 0 && strips < 65536 * 32 && n_caches < 65536, ZG_ERR_UNSUPPORTED, "kv clear: %d-byte rows, %d strips", row_bytes, strips);
    hipLaunchKernelGGL(kv_clear_tail_kernel, dim3(strips, n_caches), dim3(256), 0, s, reinterpret_cast<char*>(base), cache_stride, plane_off, row_bytes,
                       strips, ctx, from_row);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int launch_split3(const float* in, size_t rows, int K, bf16_t* out, hipStream_t s) {
    if (rows == 0) return ZG_OK;
    ZG_REQUIRE(K % 4 == 0, ZG_ERR_UNSUPPORTED, "split3: K=%d must be a multiple of 4", K);
    hipLaunchKernelGGL(split3_kernel, dim3(grid_for(rows * (size_t)(K / 4))), dim3(256), 0, s, in, out, rows, K);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int sample_launches(float* logits, int batch, int vocab, const SampleParams* params, float inv_temp, const float* u, const StepCtrl* ctrl,
                           const float* part_val, int n_part, int part_stride, float* seg_ws, int* token_out, bool write_probs, hipStream_t s) {
    ZG_REQUIRE(vocab >= 1 && vocab <= 64 * kSegMax, ZG_ERR_UNSUPPORTED, "sampler: vocabulary of %d beyond %d", vocab, 64 * kSegMax);
    ZG_REQUIRE(part_val && n_part >= 1 && seg_ws && token_out, ZG_ERR_ARG, "sampler: missing argmax partials / workspace");
    const int nseg = (vocab + 63) / 64, nb = std::min(128, (nseg + 3) / 4);
    hipLaunchKernelGGL(sample_seg_kernel, dim3(nb, batch), dim3(256), 0, s, logits, vocab, params, inv_temp, part_val, n_part, part_stride, seg_ws);
    ZG_HIP(hipGetLastError());
    hipLaunchKernelGGL(sample_pick_kernel, dim3(batch), dim3(64), 0, s, logits, vocab, params, inv_temp, u, ctrl, seg_ws, token_out);
    ZG_HIP(hipGetLastError());
    if (write_probs) {
        hipLaunchKernelGGL(sample_probs_kernel, dim3(std::min(256, (vocab + 255) / 256), batch), dim3(256), 0, s, logits, vocab, inv_temp, seg_ws);
        ZG_HIP(hipGetLastError());
    }
    return ZG_OK;
}

int launch_sample(float* logits, int batch, int vocab, float temp, const float* u, const float* part_val, int n_part, int part_stride, float* seg_ws,
                  int* token_out, bool write_probs, hipStream_t s) {
    return sample_launches(logits, batch, vocab, nullptr, 1.0f / temp, u, nullptr, part_val, n_part, part_stride, seg_ws, token_out, write_probs, s);
}

int launch_sample_step(float* logits, int batch, int vocab, const SampleParams* params, const StepCtrl* ctrl, const float* part_val, int n_part,
                       int part_stride, float* seg_ws, int* token_out, hipStream_t s) {
    return sample_launches(logits, batch, vocab, params, 0.0f, nullptr, ctrl, part_val, n_part, part_stride, seg_ws, token_out, false, s);
}

size_t sample_workspace_floats(int batch) { return (size_t)batch * (kSegMax + 2); }

int launch_embed_step(const EmbedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(embed_step_kernel, dim3(1), dim3(a.batch > 4 ? 512 : 256), 0, s, a);  // one wave per sequence
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
