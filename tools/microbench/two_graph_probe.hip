// two_graph_probe.hip — can consecutive dependent M = 1 Linear kernels overlap their launch + weight-fetch ramps when they
// are replayed from TWO SEPARATELY LAUNCHED hipGraphs on two streams (even stages in one, odd stages in the other) and hand
// their activation vector over by tagged data instead of a kernel boundary?  (handoff_probe.hip mode 2 forked two branches
// INSIDE one captured graph: those were not co-resident.  Two launches on two streams are how prefetch.hip co-runs today.)
//   mode 0: one graph, one stream, plain fp32 vector                      (today's decode step: a boundary per stage)
//   mode 1: one graph, one stream, {value, tag} vector gathered through LDS (what the tagging alone costs)
//   mode 2: two graphs, two streams, tagged                                (stage k + 1 resident — weights in registers,
//                                                                           polling — while stage k runs)
//   mode 3 / 4: no graphs — eager launches on one stream / alternating between two streams, tagged (host-bound, but shows
//               whether kernels of two streams are co-resident at all)
// A stage: out[n] = sum_k W[n][k] in[k], N = K, bf16 weights streamed once; 256 threads, wave w owns K quarter w: it gathers
// its quarter of the input ONCE (three 8-byte sc1 loads per lane), checks the tags, leaves fp32 in LDS; lanes then read their
// 48 values.  Polls are bounded: a time-out sets err and the stage proceeds — a bug cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct Ctrl { unsigned seq[2]; unsigned err; unsigned spins; };
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline float bf(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ void begin_kernel(Ctrl* c, int which, u64* vec, float* fvec, int K) {
    const unsigned s = c->seq[which] + 1;  // this graph's step number (both graphs are launched once per step)
    if (vec)
        for (int i = threadIdx.x; i < K; i += blockDim.x) {
            const float v = 0.25f + 0.001f * (float)(i % 17);
            __hip_atomic_store(vec + i, ((u64)(s * 256u) << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fvec[i] = v;
        }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&c->seq[which], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// KQ = 48 values per lane; QL lanes share a row: rows per workgroup RW = 64 / QL; K = 4 * QL * 48
template <int QL, bool TAGGED>
__global__ __launch_bounds__(256) void stage_kernel(const unsigned short* __restrict__ W, const u64* in, u64* out, const float* fin, float* fout,
                                                    Ctrl* c, int which, int K, int stage) {
    constexpr int KQ = 48, RW = 64 / QL, KW = QL * KQ;  // KW: K elements per wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane / QL, q = lane % QL;
    const int row = blockIdx.x * RW + r;
    const int kb = (w * QL + q) * KQ;
    __shared__ float xs[4][KW];
    __shared__ float part[4][16];
    u32x4 wr[KQ / 8];
#pragma unroll
    for (int i = 0; i < KQ / 8; ++i) wr[i] = __builtin_nontemporal_load((const u32x4*)(W + (size_t)row * K + kb) + i);
    float x[KQ];
    if (TAGGED) {
        const unsigned want = __hip_atomic_load(&c->seq[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 256u + (unsigned)stage;
        constexpr int PL = KW / 64;  // granules per lane: 3 (QL = 4) or 6 (QL = 8)
        unsigned spins = 0;
        for (;;) {
            u64 p[PL];
            bool ok = true;
#pragma unroll
            for (int i = 0; i < PL; ++i) p[i] = __hip_atomic_load(in + w * KW + i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < PL; ++i) ok &= (unsigned)(p[i] >> 32) == want;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#pragma unroll
                for (int i = 0; i < PL; ++i) xs[w][i * 64 + lane] = __uint_as_float((unsigned)p[i]);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 13)) { if (lane == 0) c->err = 1000u + (unsigned)stage; break; }
        }
        if (spins && lane == 0 && w == 0 && blockIdx.x == 0) atomicAdd(&c->spins, spins);
        // (wave-private strip: LDS operations of one wave complete in order — no barrier)
#pragma unroll
        for (int i = 0; i < KQ; ++i) x[i] = xs[w][q * KQ + i];
    } else {
#pragma unroll
        for (int i = 0; i < KQ; ++i) x[i] = fin[kb + i];
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < KQ / 8; ++i) {
        const unsigned u[4] = {wr[i].x, wr[i].y, wr[i].z, wr[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc += bf((unsigned short)(u[j] & 0xffff)) * x[i * 8 + 2 * j];
            acc += bf((unsigned short)(u[j] >> 16)) * x[i * 8 + 2 * j + 1];
        }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (QL == 8) acc += __shfl_xor(acc, 4);
    if (q == 0) part[w][r] = acc;
    __syncthreads();
    if (threadIdx.x < RW) {
        const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        const int n = blockIdx.x * RW + threadIdx.x;
        if (TAGGED) {
            const unsigned seq = __hip_atomic_load(&c->seq[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(out + n, ((u64)(seq * 256u + (unsigned)stage + 1u) << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
            fout[n] = v;
    }
}

template <int QL>
static void launch_stage(bool tagged, hipStream_t s, int G, const unsigned short* W, const u64* in, u64* out, const float* fin, float* fout, Ctrl* c,
                         int which, int K, int stage) {
    if (tagged) hipLaunchKernelGGL((stage_kernel<QL, true>), dim3(G), dim3(256), 0, s, W, in, out, fin, fout, c, which, K, stage);
    else hipLaunchKernelGGL((stage_kernel<QL, false>), dim3(G), dim3(256), 0, s, W, in, out, fin, fout, c, which, K, stage);
}

int main() {
    const int L = 61, R = 200;  // (odd: the last stage runs in the graph that also holds the next step's begin kernel, which rewrites the vector the last-but-one stage writes)
    CK(hipSetDevice(0));
    for (int K : {768, 1536}) {
        const int N = K, QL = K / 192, G = N / (64 / QL);
        std::vector<unsigned short> hw((size_t)N * K);
        unsigned bits; float inv = 1.0f / (float)K; memcpy(&bits, &inv, 4);
        for (auto& v : hw) v = (unsigned short)(bits >> 16);
        unsigned short* W; CK(hipMalloc(&W, (size_t)L * N * K * 2));
        for (int l = 0; l < L; ++l) CK(hipMemcpy(W + (size_t)l * N * K, hw.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
        u64* vec[2]; float* fvec[2]; Ctrl* c;
        for (int i = 0; i < 2; ++i) { CK(hipMalloc(&vec[i], K * 8)); CK(hipMemset(vec[i], 0, K * 8)); CK(hipMalloc(&fvec[i], K * 4)); CK(hipMemset(fvec[i], 0, K * 4)); }
        CK(hipMalloc(&c, sizeof(Ctrl))); CK(hipMemset(c, 0, sizeof(Ctrl)));
        for (int mode = 0; mode <= 6; ++mode) {
            const int S = (mode == 2 || mode >= 4) ? 2 : 1;
            const bool eager = mode == 3 || mode == 4;  // modes 5 / 6: as 2, the second stream at lower / higher priority
            const bool tagged = mode >= 1;
            CK(hipMemset(c, 0, sizeof(Ctrl)));  // both graphs count their steps from zero: their tags must agree
            for (int i = 0; i < 2; ++i) CK(hipMemset(vec[i], 0, K * 8));
            CK(hipDeviceSynchronize());
            hipStream_t st[2];
            int plo = 0, phi = 0;
            CK(hipDeviceGetStreamPriorityRange(&plo, &phi));  // plo = lowest priority (largest number)
            CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
            if (mode == 5) CK(hipStreamCreateWithPriority(&st[1], hipStreamNonBlocking, plo));
            else if (mode == 6) CK(hipStreamCreateWithPriority(&st[1], hipStreamNonBlocking, phi));
            else CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
            hipGraph_t g[2]; hipGraphExec_t ge[2];
            for (int gi = 0; gi < S; ++gi) {
                CK(hipStreamBeginCapture(st[gi], hipStreamCaptureModeThreadLocal));
                hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(256), 0, st[gi], c, gi, gi == 0 ? vec[0] : nullptr, fvec[0], K);
                for (int k = gi; k < L; k += S) {
                    const unsigned short* Wk = W + (size_t)k * N * K;
                    if (QL == 4) launch_stage<4>(tagged, st[gi], G, Wk, vec[k & 1], vec[(k + 1) & 1], fvec[k & 1], fvec[(k + 1) & 1], c, gi, K, k);
                    else launch_stage<8>(tagged, st[gi], G, Wk, vec[k & 1], vec[(k + 1) & 1], fvec[k & 1], fvec[(k + 1) & 1], c, gi, K, k);
                }
                CK(hipStreamEndCapture(st[gi], &g[gi]));
                CK(hipGraphInstantiate(&ge[gi], g[gi], nullptr, nullptr, 0));
            }
            auto step = [&]() {
                if (!eager) {
                    for (int gi = 0; gi < S; ++gi) CK(hipGraphLaunch(ge[gi], st[gi]));
                    return;
                }
                for (int gi = 0; gi < S; ++gi) hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(256), 0, st[gi], c, gi, gi == 0 ? vec[0] : nullptr, fvec[0], K);
                for (int k = 0; k < L; ++k) {
                    const int gi = k % S;
                    const unsigned short* Wk = W + (size_t)k * N * K;
                    if (QL == 4) launch_stage<4>(tagged, st[gi], G, Wk, vec[k & 1], vec[(k + 1) & 1], fvec[k & 1], fvec[(k + 1) & 1], c, gi, K, k);
                    else launch_stage<8>(tagged, st[gi], G, Wk, vec[k & 1], vec[(k + 1) & 1], fvec[k & 1], fvec[(k + 1) & 1], c, gi, K, k);
                }
            };
            for (int i = 0; i < 20; ++i) step();
            for (int gi = 0; gi < S; ++gi) CK(hipStreamSynchronize(st[gi]));
            float best = 1e9f;
            hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, st[0]));
                for (int i = 0; i < R; ++i) step();
                if (S == 2) { CK(hipEventRecord(ej, st[1])); CK(hipStreamWaitEvent(st[0], ej, 0)); }
                CK(hipEventRecord(e1, st[0]));
                for (int gi = 0; gi < S; ++gi) CK(hipStreamSynchronize(st[gi]));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            Ctrl hc; CK(hipMemcpy(&hc, c, sizeof(Ctrl), hipMemcpyDeviceToHost));
            std::vector<u64> hv(K); CK(hipMemcpy(hv.data(), vec[L & 1], K * 8, hipMemcpyDeviceToHost));
            std::vector<float> hf(K); CK(hipMemcpy(hf.data(), fvec[L & 1], K * 4, hipMemcpyDeviceToHost));
            float o0; unsigned lo = (unsigned)hv[0]; memcpy(&o0, &lo, 4);
            printf("{\"K\": %d, \"workgroups\": %d, \"mode\": %d, \"graphs_streams\": %d, \"tagged\": %d, \"us_per_stage\": %.3f, \"us_per_step\": %.1f, \"err\": %u, \"spins\": %u, \"out0\": %g}\n",
                   K, G, mode, S, (int)tagged, best * 1e3f / (R * L), best * 1e3f / R, hc.err, hc.spins, tagged ? o0 : hf[0]);
            fflush(stdout);
            CK(hipMemset(&c->err, 0, 8));
            for (int gi = 0; gi < S; ++gi) { CK(hipGraphExecDestroy(ge[gi])); CK(hipGraphDestroy(g[gi])); }
            for (int i = 0; i < 2; ++i) CK(hipStreamDestroy(st[i]));
        }
        CK(hipFree(W));
    }
    return 0;
}
