// paired_mlp_probe.hip — VERDICT round 3, item 3(b): does "column -> row pairing" of the MLP half of a Block beat the two
// dependent launches the decode step runs today (ln_2 + c_fc + GELU, then mlp c_proj + residual: src/main.zig:140-146)?
//
//   mode 0 (two launches per layer, today's dataflow):
//       A: h[4E]   = gelu(W_fc ln_2(x) + b_fc)            grid 4E/U workgroups, U hidden units each
//       B: x'[E]   = x + W_proj h + b_proj                 one wave per output row, 4 rows per workgroup
//   mode 1 (one launch per layer, paired; the partials are laid out [E/16][G][16] so that a reducer reads ONE contiguous run):
//       workgroup w owns hidden units [U w, U w + U): it computes them (as A does) and at once multiplies them with the
//       matching K-slab of W_proj — re-tiled at load to [4E/U][E][U], so the slab is one contiguous 2 U E bytes — giving a
//       partial E-vector, stored as (value, tag) words; the LAST E/16 workgroups of the grid (dispatched after their
//       writers) poll the tags and reduce 16 output rows each over the G partials IN A FIXED ORDER (workgroup order: sixteen
//       segment sums, then a serial sum of the sixteen), add bias and residual, write x'.  Deterministic; polls bounded.
//   Both modes issue every weight load of a workgroup before anything else.  A chain is L layers with distinct weights
//   (12 x 9.4 MB: more than the L2s hold, as in the real step), captured in one hipGraph with one epoch-bump kernel in front.
//   The result of both modes is compared with a host double-precision chain.
// usage: paired_mlp_probe [E=768] [L=12] [reps=200]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Ctrl { unsigned epoch; unsigned err; unsigned spins; unsigned pad; };
constexpr int E = 768, H = 4 * E;

__device__ __forceinline__ float lo16(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi16(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ float gelu(float x) {  // src/ops.zig:225 in the x / (1 + exp(-2u)) form
    const float u = x * 0.7978845608f * (1.0f + 0.044715f * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}
template <int W>
__device__ __forceinline__ float group_sum(float v) {  // all-reduce over W = 8, 16 or 32 consecutive lanes
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    if (W >= 16) v += __shfl_xor(v, 8, 64);
    if (W == 32) v += __shfl_xor(v, 16, 64);
    return v;
}

struct Layer {
    const bf16_t* wfc;    // [H][E]
    const bf16_t* wproj;  // mode 0: [E][H]; mode 1: [H/U][E][U]
    const float *bfc, *bproj, *g, *b;
};

__global__ void bump_kernel(Ctrl* c) { if (threadIdx.x == 0) c->epoch = c->epoch + 1; }

// Hidden units [U blockIdx.x, +U) of one layer: LPR = 256 / U lanes per fc row, 96 / LPR chunks of 8 weights per lane.
// Returns the unit's value in every lane of its group.
template <int U>
struct FcRegs { u32x4 w[96 * U / 256]; };
template <int U>
__device__ __forceinline__ void fc_issue(const Layer& y, FcRegs<U>& r) {
    constexpr int LPR = 256 / U, NC = 96 / LPR;
    const int row = blockIdx.x * U + threadIdx.x / LPR, l = threadIdx.x % LPR;
#pragma unroll
    for (int i = 0; i < NC; ++i) r.w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y.wfc + (size_t)row * E) + l + LPR * i);
}
template <int U>
__device__ __forceinline__ float fc_finish(const Layer& y, const float* __restrict__ x, const FcRegs<U>& r) {
    constexpr int LPR = 256 / U, NC = 96 / LPR;
    const int row = blockIdx.x * U + threadIdx.x / LPR, l = threadIdx.x % LPR;
    f32x4 xv[NC][2], gv[NC][2], bv[NC][2];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (l + LPR * i) * 8 + 4 * h;
            xv[i][h] = *reinterpret_cast<const f32x4*>(x + k);
            gv[i][h] = *reinterpret_cast<const f32x4*>(y.g + k);
            bv[i][h] = *reinterpret_cast<const f32x4*>(y.b + k);
        }
    const float bias = y.bfc[row];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) s += xv[i][h].x + xv[i][h].y + xv[i][h].z + xv[i][h].w;
    const float mean = group_sum<LPR>(s) * (1.0f / E);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 d = xv[i][h] - mean;
            q += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
    const float rstd = 1.0f / sqrtf(group_sum<LPR>(q) * (1.0f / E) + 1e-5f);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const f32x4 n0 = (xv[i][0] - mean) * rstd * gv[i][0] + bv[i][0], n1 = (xv[i][1] - mean) * rstd * gv[i][1] + bv[i][1];
        const u32x4 w = r.w[i];
        acc += lo16(w.x) * n0.x + hi16(w.x) * n0.y + lo16(w.y) * n0.z + hi16(w.y) * n0.w;
        acc += lo16(w.z) * n1.x + hi16(w.z) * n1.y + lo16(w.w) * n1.z + hi16(w.w) * n1.w;
    }
    return gelu(group_sum<LPR>(acc) + bias);
}

template <int U>
__global__ __launch_bounds__(256) void fc_kernel(Layer y, const float* __restrict__ x, float* __restrict__ h) {
    FcRegs<U> r;
    fc_issue<U>(y, r);
    const float v = fc_finish<U>(y, x, r);
    constexpr int LPR = 256 / U;
    if (threadIdx.x % LPR == 0) h[blockIdx.x * U + threadIdx.x / LPR] = v;
}

// mode 0, second launch: one wave per output row (K = 3072: 6 chunks of 8 per lane), 4 rows per workgroup
__global__ __launch_bounds__(256) void proj_kernel(Layer y, const float* __restrict__ h, const float* __restrict__ x, float* __restrict__ xo) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    u32x4 w[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y.wproj + (size_t)n * H) + lane + 64 * i);
    const float add = y.bproj[n] + x[n];
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float* hp = h + (lane + 64 * i) * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(hp), b = *reinterpret_cast<const f32x4*>(hp + 4);
        acc += lo16(w[i].x) * a.x + hi16(w[i].x) * a.y + lo16(w[i].y) * a.z + hi16(w[i].y) * a.w;
        acc += lo16(w[i].z) * b.x + hi16(w[i].z) * b.y + lo16(w[i].w) * b.z + hi16(w[i].w) * b.w;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) xo[n] = acc + add;
}

// mode 1: the pair in one launch.  part: [E / RO][G][RO] (value, tag) words; RO = output rows per reducing workgroup.
template <int U, int RO>
__global__ __launch_bounds__(256) void paired_kernel(Layer y, const float* __restrict__ x, float* __restrict__ xo, u64* part, Ctrl* c, int id,
                                                     unsigned spin_limit, int abl) {
    constexpr int G = H / U, LPR = 256 / U, NRED = E / RO, NP = 256 / RO, SEG = G / NP;
    static_assert(SEG * NP == G && NRED <= G, "shape");
    __shared__ float hs[U];
    __shared__ float seg[NP][RO];
    FcRegs<U> r;
    fc_issue<U>(y, r);
    // the slab [E][U]: thread t owns output rows t, t + 256, t + 512 (2 U bytes each)
    u32x4 ws[3][U / 8];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < U / 8; ++q)
            ws[j][q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y.wproj + ((size_t)blockIdx.x * E + threadIdx.x + 256 * j) * U) + q);
    const unsigned tag = (__hip_atomic_load(&c->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 8) | (unsigned)id;
    const float v = fc_finish<U>(y, x, r);
    if (threadIdx.x % LPR == 0) hs[threadIdx.x / LPR] = v;
    __syncthreads();
    // word of (workgroup w, output n): ((n / RO) * G + w) * RO + n % RO — RO lanes store one 8 RO-byte run, a reducer's input is contiguous
    u64* mine = part + (size_t)blockIdx.x * RO;
#define PIDX(n) ((size_t)((n) / RO) * (G * RO) + ((n) % RO))
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < U / 8; ++q) {
            const u32x4 w = ws[j][q];
            const float* hp = hs + 8 * q;
            acc += lo16(w.x) * hp[0] + hi16(w.x) * hp[1] + lo16(w.y) * hp[2] + hi16(w.y) * hp[3];
            acc += lo16(w.z) * hp[4] + hi16(w.z) * hp[5] + lo16(w.w) * hp[6] + hi16(w.w) * hp[7];
        }
        if (abl & 2) mine[PIDX(threadIdx.x + 256 * j)] = ((u64)tag << 32) | __float_as_uint(acc);  // (ablation: plain stores — the reduce is then skipped)
        else if (!(abl & 4)) __hip_atomic_store(mine + PIDX(threadIdx.x + 256 * j), ((u64)tag << 32) | __float_as_uint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (acc == 123.456f) mine[0] = 0;  // (ablation: no partial stores at all)
    }
    const int red = (int)blockIdx.x - (G - NRED);  // the last NRED workgroups reduce RO output rows each
    if (red < 0) return;
    if (abl) {  // ablations 1 / 2 / 4: no poll, no reduce
        if (threadIdx.x < RO) xo[red * RO + threadIdx.x] = y.bproj[red * RO + threadIdx.x] + x[red * RO + threadIdx.x];
        return;
    }
    const int o = threadIdx.x % RO, p = threadIdx.x / RO;
    const u64* src = part + (size_t)red * (G * RO) + (size_t)(p * SEG) * RO + o;
    float s = 0.0f;
    unsigned spins = 0;
    u64 g[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) g[i] = __hip_atomic_load(src + (size_t)i * RO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
        while ((unsigned)(g[i] >> 32) != tag) {
            if (++spins > spin_limit) { c->err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
            g[i] = __hip_atomic_load(src + (size_t)i * RO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s += __uint_as_float((unsigned)g[i]);
    }
    seg[p][o] = s;
    __syncthreads();
    if (threadIdx.x < RO) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < NP; ++q) t += seg[q][threadIdx.x];
        const int nn = red * RO + threadIdx.x;
        xo[nn] = t + y.bproj[nn] + x[nn];
    }
    if (spins && threadIdx.x == 0) atomicAdd(&c->spins, spins);
}

static bf16_t f2bf(float f) { unsigned b; memcpy(&b, &f, 4); b += 0x7fffu + ((b >> 16) & 1u); return (bf16_t)(b >> 16); }
static float bf2f(bf16_t h) { unsigned b = (unsigned)h << 16; float f; memcpy(&f, &b, 4); return f; }
static unsigned rng_state = 12345u;
static float rnd() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f; }

template <int U, int RO>
static void run_paired(int L, int reps, const std::vector<Layer>& lay, float* x0, float* xa, float* xb, u64* part, Ctrl* c, hipStream_t s,
                       const std::vector<double>& want, float base_us, const char* name, int abl = 0) {
    constexpr int G = H / U;
    CK(hipMemsetAsync(c, 0, sizeof(Ctrl), s));
    CK(hipMemsetAsync(part, 0, (size_t)G * E * 8, s));
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s, c);
    CK(hipMemcpyAsync(xa, x0, E * 4, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL((paired_kernel<U, RO>), dim3(G), dim3(256), 0, s, lay[l], (l & 1) ? xb : xa, (l & 1) ? xa : xb, part, c, l + 1, 1u << 18, abl);
    CK(hipStreamEndCapture(s, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> out(E); Ctrl hc;
    CK(hipMemcpy(out.data(), (L & 1) ? xb : xa, E * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hc, c, sizeof(Ctrl), hipMemcpyDeviceToHost));
    double worst = 0, rms = 0;
    for (int i = 0; i < E; ++i) { worst = fmax(worst, fabs(out[i] - want[i])); rms += want[i] * want[i]; }
    const float us = ms * 1e3f / reps;
    printf("%-34s %8.2f us per chain, %6.3f us per layer (chain minus the %0.2f us of bump + copy)  err %d spins/launch %.0f  max|dx| %.2e of rms %.3f\n", name, us,
           (us - base_us) / L, base_us, hc.err, (double)hc.spins / ((reps + 10.0) * L), worst, sqrt(rms / E));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
}

int main(int argc, char** argv) {
    const int L = argc > 2 ? atoi(argv[2]) : 12, reps = argc > 3 ? atoi(argv[3]) : 200;
    if (argc > 1 && atoi(argv[1]) != E) { fprintf(stderr, "E is compiled in (%d)\n", E); return 2; }
    hipStream_t s; CK(hipStreamCreate(&s));
    // ---- weights: host fp32 (bf16-representable) + device layouts
    std::vector<std::vector<float>> wfc(L), wpr(L), bfc(L), bpr(L), gg(L), bb(L);
    std::vector<Layer> l0(L), l8(L), l16(L), l32(L);
    auto up = [&](const void* p, size_t n) { void* d; CK(hipMalloc(&d, n)); CK(hipMemcpy(d, p, n, hipMemcpyHostToDevice)); return d; };
    for (int l = 0; l < L; ++l) {
        wfc[l].resize((size_t)H * E); wpr[l].resize((size_t)E * H); bfc[l].resize(H); bpr[l].resize(E); gg[l].resize(E); bb[l].resize(E);
        std::vector<bf16_t> dfc((size_t)H * E), dpr((size_t)E * H), t8((size_t)E * H), t16((size_t)E * H), t32((size_t)E * H);
        for (size_t i = 0; i < wfc[l].size(); ++i) { dfc[i] = f2bf(rnd() * 0.08f); wfc[l][i] = bf2f(dfc[i]); }
        for (size_t i = 0; i < wpr[l].size(); ++i) { dpr[i] = f2bf(rnd() * 0.04f); wpr[l][i] = bf2f(dpr[i]); }
        for (int i = 0; i < H; ++i) bfc[l][i] = rnd() * 0.1f;
        for (int i = 0; i < E; ++i) { bpr[l][i] = rnd() * 0.1f; gg[l][i] = 1.0f + rnd() * 0.2f; bb[l][i] = rnd() * 0.1f; }
        for (int n = 0; n < E; ++n)
            for (int k = 0; k < H; ++k) {
                t8[((size_t)(k / 8) * E + n) * 8 + k % 8] = dpr[(size_t)n * H + k];
                t16[((size_t)(k / 16) * E + n) * 16 + k % 16] = dpr[(size_t)n * H + k];
                t32[((size_t)(k / 32) * E + n) * 32 + k % 32] = dpr[(size_t)n * H + k];
            }
        Layer y{};
        y.wfc = (const bf16_t*)up(dfc.data(), dfc.size() * 2);
        y.bfc = (const float*)up(bfc[l].data(), H * 4); y.bproj = (const float*)up(bpr[l].data(), E * 4);
        y.g = (const float*)up(gg[l].data(), E * 4); y.b = (const float*)up(bb[l].data(), E * 4);
        l0[l] = y; l0[l].wproj = (const bf16_t*)up(dpr.data(), dpr.size() * 2);
        l8[l] = y; l8[l].wproj = (const bf16_t*)up(t8.data(), t8.size() * 2);
        l16[l] = y; l16[l].wproj = (const bf16_t*)up(t16.data(), t16.size() * 2);
        l32[l] = y; l32[l].wproj = (const bf16_t*)up(t32.data(), t32.size() * 2);
    }
    std::vector<float> x0(E);
    for (int i = 0; i < E; ++i) x0[i] = rnd() * 2.0f;
    // ---- host chain (double)
    std::vector<double> x(x0.begin(), x0.end()), h(H);
    for (int l = 0; l < L; ++l) {
        double m = 0, v = 0;
        for (int k = 0; k < E; ++k) m += x[k];
        m /= E;
        for (int k = 0; k < E; ++k) v += (x[k] - m) * (x[k] - m);
        const double rs = 1.0 / sqrt(v / E + 1e-5);
        std::vector<double> nx(E);
        for (int k = 0; k < E; ++k) nx[k] = (x[k] - m) * rs * gg[l][k] + bb[l][k];
        for (int r = 0; r < H; ++r) {
            double a = bfc[l][r];
            for (int k = 0; k < E; ++k) a += (double)wfc[l][(size_t)r * E + k] * nx[k];
            const double u = a * 0.7978845608 * (1.0 + 0.044715 * a * a);
            h[r] = a / (1.0 + exp(-2.0 * u));
        }
        for (int n = 0; n < E; ++n) {
            double a = bpr[l][n] + x[n];
            for (int k = 0; k < H; ++k) a += (double)wpr[l][(size_t)n * H + k] * h[k];
            nx[n] = a;
        }
        x = nx;
    }
    float *dx0 = (float*)up(x0.data(), E * 4), *xa, *xb, *dh;
    CK(hipMalloc(&xa, E * 4)); CK(hipMalloc(&xb, E * 4)); CK(hipMalloc(&dh, H * 4));
    u64* part; CK(hipMalloc(&part, (size_t)(H / 8) * E * 8));
    Ctrl* c; CK(hipMalloc(&c, sizeof(Ctrl))); CK(hipMemset(c, 0, sizeof(Ctrl)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- what the chain's fixed part costs: bump + copy only
    float base_us = 0;
    {
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s, c);
        CK(hipMemcpyAsync(xa, dx0, E * 4, hipMemcpyDeviceToDevice, s));
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        base_us = ms * 1e3f / reps;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
    }
    // ---- mode 0: two launches per layer
    for (int U : {8, 16}) {
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s, c);
        CK(hipMemcpyAsync(xa, dx0, E * 4, hipMemcpyDeviceToDevice, s));
        for (int l = 0; l < L; ++l) {
            const float* xi = (l & 1) ? xb : xa; float* xo = (l & 1) ? xa : xb;
            if (U == 8) hipLaunchKernelGGL((fc_kernel<8>), dim3(H / 8), dim3(256), 0, s, l0[l], xi, dh);
            else hipLaunchKernelGGL((fc_kernel<16>), dim3(H / 16), dim3(256), 0, s, l0[l], xi, dh);
            hipLaunchKernelGGL(proj_kernel, dim3(E / 4), dim3(256), 0, s, l0[l], dh, xi, xo);
        }
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> out(E);
        CK(hipMemcpy(out.data(), (L & 1) ? xb : xa, E * 4, hipMemcpyDeviceToHost));
        double worst = 0, rms = 0;
        for (int i = 0; i < E; ++i) { worst = fmax(worst, fabs(out[i] - x[i])); rms += x[i] * x[i]; }
        const float us = ms * 1e3f / reps;
        char name[64]; snprintf(name, sizeof name, "two launches (fc U=%d, %d + %d WGs)", U, H / U, E / 4);
        printf("%-34s %8.2f us per chain, %6.3f us per layer (chain minus the %0.2f us of bump + copy)  max|dx| %.2e of rms %.3f\n", name, us, (us - base_us) / L,
               base_us, worst, sqrt(rms / E));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
    }
    run_paired<8, 16>(L, reps, l8, dx0, xa, xb, part, c, s, x, base_us, "paired (U=8, 384 WGs, 48 reduce)");
    run_paired<8, 4>(L, reps, l8, dx0, xa, xb, part, c, s, x, base_us, "paired (U=8, 384 WGs, 192 reduce)");
    run_paired<8, 2>(L, reps, l8, dx0, xa, xb, part, c, s, x, base_us, "paired (U=8, 384 WGs, 384 reduce)");
    run_paired<16, 16>(L, reps, l16, dx0, xa, xb, part, c, s, x, base_us, "paired (U=16, 192 WGs, 48 reduce)");
    run_paired<16, 4>(L, reps, l16, dx0, xa, xb, part, c, s, x, base_us, "paired (U=16, 192 WGs, 192 reduce)");
    run_paired<32, 16>(L, reps, l32, dx0, xa, xb, part, c, s, x, base_us, "paired (U=32, 96 WGs, 48 reduce)");
    run_paired<32, 8>(L, reps, l32, dx0, xa, xb, part, c, s, x, base_us, "paired (U=32, 96 WGs, 96 reduce)");
    puts("ablations (results wrong by construction): 1 = tagged stores, no poll / reduce; 2 = plain stores, no reduce; 4 = no partial stores, no reduce");
    for (int abl : {1, 2, 4}) {
        char nm[64];
        snprintf(nm, sizeof nm, "paired U=8  ablation %d", abl);  run_paired<8, 16>(L, reps, l8, dx0, xa, xb, part, c, s, x, base_us, nm, abl);
        snprintf(nm, sizeof nm, "paired U=16 ablation %d", abl); run_paired<16, 16>(L, reps, l16, dx0, xa, xb, part, c, s, x, base_us, nm, abl);
        snprintf(nm, sizeof nm, "paired U=32 ablation %d", abl); run_paired<32, 16>(L, reps, l32, dx0, xa, xb, part, c, s, x, base_us, nm, abl);
    }
    return 0;
}
