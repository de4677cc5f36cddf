// Side-stream L2 prefetcher for the token-at-a-time decode chain (GPT.forward, src/main.zig:178-195, one launch per
// Linear / attention of src/ops.zig).
//
// Why: a decode kernel's weights are read once per token, 62 launches (124M) apart, so every launch finds them on
// the memory side (Infinity Cache or HBM), never in its XCD's L2: tools/cold_chain.py prices that at +0.35..0.85 us
// per launch (+13 % of a token) against the same kernel re-reading the same layer.  A launch cannot fetch for its
// successor (s_endpgm waits for a wave's outstanding loads, so the fetch would only move into the earlier
// kernel), and branches of a captured graph are not co-resident (tools/microbench/handoff_probe.hip).  What does overlap
// with the chain is a SEPARATE, low-priority stream: one persistent kernel of a few workgroups per XCD that
// follows the chain's progress counter and touches, one or two kernels ahead, the cache lines the upcoming kernel's
// workgroups ON THE SAME XCD will read.  Blocks are dealt round robin over the XCDs starting at a queue-dependent
// one (observed: XCD 7 for a torch stream, 0 for a stream of a fresh process), so the step's first kernel publishes the
// XCC_ID its block 0 runs on and the prefetcher reads its own: a different placement costs speed, never results.
// Placement matters both ways: lines fetched into ANOTHER XCD's L2 make the consumer slower than no prefetch at all
// (284 against 241 us per token at 124M), lines in its own L2 faster (225).
//
// Only lines that are final are touched: weights (constant during generation) and KV-cache rows of EARLIER
// positions; the row the current step appends is left to the attention kernel itself.
//
// Every wait is bounded: the kernel leaves when the host-enqueued stop arrives, or when the counter has not moved
// for `idle_limit` polls (a stalled or aborted generation), so it cannot outlive its generation call.
#include "zg_common.h"
#include "zg_kernels.h"

namespace zg {

namespace {

// One 4-byte load per 128-byte line: the line lands in this XCD's L2, the value is discarded.  A plain load: the
// compiler waits for it only where the value is consumed (a `volatile` one is waited for on the spot).
__device__ __forceinline__ unsigned touch(const char* p) { return *reinterpret_cast<const unsigned*>(p); }

// Lines of job j that the blocks hosted by this XCD (b = mine, mine + 8, ...) of the upcoming launch will read; this
// thread takes lines gt, gt + stride, ... of that set.  Loads are fire and forget: a value is consumed (so that the
// load stays alive) only when its slot of `pend` is reused, normally by a later job, so neither the job nor the next
// poll waits for the memory round trip.
constexpr int kPend = 8;
struct Pending {
    unsigned v[kPend];
};
#define ZG_PF_SLOT(u, cond, addr)                \
    {                                             \
        sink ^= pend.v[u];                        \
        pend.v[u] = (cond) ? touch(addr) : 0u;    \
    }

__device__ __forceinline__ unsigned prefetch_job(const PfJob& j, unsigned mine, unsigned gt, unsigned stride, unsigned T, unsigned ls,
                                                 unsigned cap_bytes, Pending& pend) {
    unsigned sink = 0;
    if (j.kind == PF_WEIGHTS) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const unsigned nl = (j.aux_bytes[v] + 127u) >> 7;
            for (unsigned l = gt; l < nl; l += stride) ZG_PF_SLOT(v, true, j.aux[v] + ((size_t)l << 7))
        }
        unsigned lpt = (j.touch_bytes + (1u << ls) - 1u) >> ls;  // lines fetched per workgroup tile
        if (cap_bytes != 0 && j.n_wg != 0) {
            const unsigned cap_lpt = max((cap_bytes / j.n_wg) >> ls, 1u);
            lpt = min(lpt, cap_lpt);
        }
        const unsigned ntile = j.n_wg > mine ? (j.n_wg - mine + 7u) >> 3 : 0u;
        const unsigned nlines = ntile * lpt;
        for (unsigned l0 = gt; l0 < nlines; l0 += kPend * stride) {
#pragma unroll
            for (int u = 0; u < kPend; ++u) {
                const unsigned l = l0 + u * stride;
                const unsigned t = l / lpt, r = l - t * lpt;
                const size_t off = (size_t)(mine + 8u * t) * j.wg_bytes + ((size_t)r << ls);
                ZG_PF_SLOT(u, l < nlines && off < j.total_bytes, j.base + off)
            }
        }
    } else if (j.kind == PF_KV && T >= 2) {
        // attention grid (n_heads, splits, batch), block id = h + H (split + splits b); its K and V rows of
        // positions [split * 256, min(T - 1, (split + 1) * 256)): row T - 1 is being appended by this very step
        const unsigned H = j.n_heads, ctx = j.ctx;
        unsigned t_hi = ((T + 63u) >> 6) << 6;
        if (t_hi > ctx) t_hi = ctx;
        const unsigned splits = (t_hi + kAttnChunk - 1) / kAttnChunk;
        const unsigned nblk = H * splits * j.batch;
        const unsigned ntile = nblk > mine ? (nblk - mine + 7u) >> 3 : 0u;
        const unsigned row_lines = j.row_bytes >> ls;             // lines per position (fp32: 2 of 128 bytes)
        const unsigned lpt = kAttnChunk * row_lines * 2;          // K and V
        const unsigned nlines = ntile * lpt;
        for (unsigned l0 = gt; l0 < nlines; l0 += kPend * stride) {
#pragma unroll
            for (int u = 0; u < kPend; ++u) {
                const unsigned l = l0 + u * stride;
                const unsigned t = l / lpt, r = l - t * lpt;
                const unsigned blk = mine + 8u * t;
                const unsigned h = blk % H, sb = blk / H, split = sb % splits, b = sb / splits;
                const unsigned kv = r >= lpt / 2, rr = kv ? r - lpt / 2 : r;
                const unsigned pos = split * kAttnChunk + rr / row_lines;
                const size_t off = ((size_t)b * H * ctx + (size_t)h * ctx + pos) * j.row_bytes + ((size_t)(rr % row_lines) << ls);
                ZG_PF_SLOT(u, l < nlines && pos + 1 < T, (kv ? j.base2 : j.base) + off)
            }
        }
    }
    return sink;
}
#undef ZG_PF_SLOT

// Wave 0 of a workgroup follows the chain (agent-scope loads of the progress word) and publishes what it sees in LDS;
// waves 1..3 spin on that LDS word and do the fetching.  The roles are split so that the poll never waits for the
// fetchers' loads in flight and the fetchers never wait for the poll's round trip; no barrier after the start-up.
constexpr unsigned kPfWorkers = 192;  // fetching threads per workgroup (waves 1..3)
constexpr unsigned long long kPfLeave = 0xffffffffffffffffull;

__global__ __launch_bounds__(256) void prefetch_kernel(const PfArgs a) {
    __shared__ unsigned s_word;
    __shared__ unsigned long long s_seen;  // {base_xcd, progress} as last read by wave 0; kPfLeave = leave
    __shared__ PfJob s_jobs[256];
    // HW_REG_XCC_ID (id 20), bits [3:0]: simm16 = (size - 1) << 11 | offset << 6 | id
    const unsigned xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    if (threadIdx.x == 0) {
        s_word = __hip_atomic_fetch_add(&a.ctl->ticket[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_seen = 0;
    }
    const unsigned njobs = (unsigned)a.njobs;
    for (unsigned i = threadIdx.x; i < njobs * (unsigned)(sizeof(PfJob) / 4); i += 256)
        reinterpret_cast<unsigned*>(s_jobs)[i] = reinterpret_cast<const unsigned*>(a.jobs)[i];
    __syncthreads();
    const unsigned sub = s_word;
    if (sub >= (unsigned)a.nsub) return;  // more workgroups on this XCD than planned: the others cover it
    volatile unsigned long long* seen = &s_seen;

    if (threadIdx.x < 64) {  // ---- wave 0: the poller
        unsigned last = 0, idle = 0, reason = 0;
        for (;;) {
            const unsigned long long w = __hip_atomic_load(static_cast<const unsigned long long*>(__builtin_assume_aligned(&a.ctl->progress, 8)),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned P = (unsigned)w;
            if (P == PF_STOP) { reason = 1; break; }
            if (P == last) {
                if (++idle > a.idle_limit) { reason = 2; break; }
                for (unsigned i = 0; i < a.sleep; ++i) __builtin_amdgcn_s_sleep(8);
                continue;
            }
            idle = 0;
            last = P;
            if (threadIdx.x == 0) *seen = w;
        }
        if (threadIdx.x == 0) {
            *seen = kPfLeave;
            if (sub == 0) a.ctl->exit_reason[xcd] = reason;
        }
        return;
    }

    // ---- waves 1..3: the fetchers
    const unsigned gt = sub * kPfWorkers + (threadIdx.x - 64u), stride = (unsigned)a.nsub * kPfWorkers;
    unsigned long long cursor = 0;  // next job to fetch, as T * njobs + index
    unsigned long long last = 0;
    unsigned sink = 0, jobs_done = 0;
    Pending pend;
#pragma unroll
    for (int u = 0; u < kPend; ++u) pend.v[u] = 0;
    for (;;) {
        const unsigned long long w = *seen;
        if (w == kPfLeave) break;
        if (w == last) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        last = w;
        const unsigned P = (unsigned)w, base = (unsigned)(w >> 32);
        const unsigned T = P >> 8, started = P & 255u;
        if (started == 0 || !(base & 0x100u)) continue;
        // block b of a launch runs on XCD (base + b) % 8: this XCD hosts the blocks b = mine - base (mod 8)
        const unsigned mine = (xcd - base) & 7u;
        const unsigned long long running = (unsigned long long)T * njobs + (started - 1u);
        if (cursor <= running) cursor = running + 1;
        const unsigned long long until = running + (unsigned)a.lead;
        while (cursor <= until) {
            const unsigned idx = (unsigned)(cursor % njobs);
            const unsigned Tj = (unsigned)(cursor / njobs);
            if (Tj <= (unsigned)a.max_T && ((a.cls_mask >> s_jobs[idx].cls) & 1u))
                sink ^= prefetch_job(s_jobs[idx], mine, gt, stride, Tj, a.line_shift, a.cap_bytes, pend);
            ++cursor;
            ++jobs_done;
        }
    }
#pragma unroll
    for (int u = 0; u < kPend; ++u) sink ^= pend.v[u];
    if (gt == 0) a.ctl->jobs_done[xcd] = jobs_done;
    if (sink == 0x9e3779b9u && a.ctl->sink_guard) a.ctl->sink = sink;  // keeps the loads alive; never taken in practice
}

}  // namespace

int launch_prefetcher(const PfArgs& a, hipStream_t s) {
    ZG_REQUIRE(a.ctl && a.jobs && a.njobs >= 1 && a.njobs <= 255 && a.nsub >= 1 && a.lead >= 1, ZG_ERR_ARG, "prefetcher: bad arguments");
    hipLaunchKernelGGL(prefetch_kernel, dim3(8 * a.nsub), dim3(256), 0, s, a);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
