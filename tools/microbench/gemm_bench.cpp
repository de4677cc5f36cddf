// gemm_bench.cpp — stand-alone driver of zg_gemm_bf16_nt (the C ABI of libzgpt2_hip.so) for fast kernel iteration on the GPU
// box: no Python / torch start-up.  Checks a sample of outputs against a CPU fp64 reference, compares the two kernel
// generations bitwise-tolerantly, screens for races by repeat runs, and times back-to-back launches (min and mean of batches).
//   gemm_bench [M N K] [-k s4|p8] [-b batches] [-i iters] [-nocheck] [-stamps]
// Environment passes through (ZGPT2_GEMM_KERNEL, ZGPT2_GEMM_DBG, ZGPT2_GEMM_BN ...; read per call by the library).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/zgpt2.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define ZK(x) do { int s_ = (x); if (s_ != 0) { printf("zg error %d (%s) at line %d\n", s_, zg_last_error(), __LINE__); exit(1);} } while (0)

extern "C" int zg_debug_gemm_stamps(unsigned long long* out, size_t n_words) __attribute__((weak));

static uint16_t f2bf(float x) { uint32_t b; memcpy(&b, &x, 4); b += 0x7FFFu + ((b >> 16) & 1u); return (uint16_t)(b >> 16); }
static float bf2f(uint16_t h) { uint32_t b = (uint32_t)h << 16; float x; memcpy(&x, &b, 4); return x; }
static uint32_t rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) * (1.0f / 16777216.0f); }
static float nrand() { float s = 0; for (int i = 0; i < 12; ++i) s += urand(); return s - 6.0f; }
static double gelu_ref(double x) { const double u = x * 0.7978845608 * (1.0 + 0.044715 * x * x); return 0.5 * x * (1.0 + tanh(u)); }

int main(int argc, char** argv) {
    int M = 8192, N = 3072, K = 768, batches = 5, iters = 50, npos = 0;
    bool check = true, gelu = true, out_bf16 = true, stamps = false;
    int fill = 0;  // 0 random (A uniform(-1,1), B normal(0, 0.02)), 1 all zero, 2 constant 1.0 / 0.02, 3 random sign only
    const char* kern = nullptr;
    const char* refk = nullptr;  // -ref <kernel>: the output must be bitwise that of this other kernel generation
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-k")) kern = argv[++i];
        else if (!strcmp(argv[i], "-ref")) refk = argv[++i];
        else if (!strcmp(argv[i], "-b")) batches = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-i")) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-nocheck")) check = false;
        else if (!strcmp(argv[i], "-nogelu")) gelu = false;
        else if (!strcmp(argv[i], "-f32")) out_bf16 = false;
        else if (!strcmp(argv[i], "-stamps")) stamps = true;
        else if (!strcmp(argv[i], "-fill")) fill = atoi(argv[++i]);
        else { const int v = atoi(argv[i]); if (npos == 0) M = v; else if (npos == 1) N = v; else K = v; ++npos; }
    }
    if (kern) setenv("ZGPT2_GEMM_KERNEL", kern, 1);
    ZK(zg_init(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    ZK(zg_set_stream(st));
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    std::vector<float> hbias(N);
    for (auto& v : hA) v = f2bf(fill == 1 ? 0.0f : fill == 2 ? 1.0f : fill == 3 ? (urand() < 0.5f ? -1.0f : 1.0f) : urand() * 2.0f - 1.0f);
    for (auto& v : hB) v = f2bf(fill == 1 ? 0.0f : fill == 2 ? 0.02f : fill == 3 ? (urand() < 0.5f ? -0.02f : 0.02f) : nrand() * 0.02f);
    for (auto& v : hbias) v = nrand() * 0.02f;
    uint16_t *dA, *dB; float* dbias; void* dC;
    const size_t esz = out_bf16 ? 2 : 4;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dC, (size_t)M * N * esz));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    auto run = [&]() { ZK(zg_gemm_bf16_nt(dA, dB, dbias, dC, M, N, K, gelu, out_bf16)); };
    auto fetch = [&](std::vector<uint8_t>& h) { h.resize((size_t)M * N * esz); CK(hipStreamSynchronize(st)); CK(hipMemcpy(h.data(), dC, h.size(), hipMemcpyDeviceToHost)); };
    auto val = [&](const std::vector<uint8_t>& h, size_t i) -> float {
        if (out_bf16) return bf2f(reinterpret_cast<const uint16_t*>(h.data())[i]);
        return reinterpret_cast<const float*>(h.data())[i];
    };
    if (check) {
        CK(hipMemset(dC, 0xFF, (size_t)M * N * esz));
        CK(hipDeviceSynchronize());  // the library launches on its own non-blocking stream: the memset must be done first
        run();
        std::vector<uint8_t> h0, h1;
        fetch(h0);
        // sampled fp64 reference: whole rows spread over the matrix (every tile row / column position is hit)
        int bad = 0; double worst = 0;
        const int nrows = 48;
        for (int s = 0; s < nrows; ++s) {
            const int m = (int)(((long long)s * 1103515245LL + 12345) % M + M) % M;
            for (int n = 0; n < N; ++n) {
                double acc = hbias[n];
                for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hB[(size_t)n * K + k]);
                if (gelu) acc = gelu_ref(acc);
                const double got = val(h0, (size_t)m * N + n), err = fabs(got - acc);
                const double tol = (out_bf16 ? 0.01 : 2e-4) * fabs(acc) + (out_bf16 ? 2e-3 : 2e-4);
                if (!(err <= tol)) { if (bad < 5) printf("  mismatch at (%d,%d): got %g want %g\n", m, n, got, acc); ++bad; }
                worst = std::max(worst, err);
            }
        }
        bool same = true;
        for (int r = 0; r < 3; ++r) { run(); fetch(h1); same &= h0 == h1; }
        printf("check %dx%dx%d %s: %d rows sampled, bad %d, max abs err %.3g, repeatable %s\n", M, N, K, getenv("ZGPT2_GEMM_KERNEL") ? getenv("ZGPT2_GEMM_KERNEL") : "default",
               nrows, bad, worst, same ? "yes" : "NO");
        if (bad || !same) return 1;
        if (refk) {
            const char* mine = getenv("ZGPT2_GEMM_KERNEL");
            std::string keep = mine ? mine : "";
            setenv("ZGPT2_GEMM_KERNEL", refk, 1);
            CK(hipMemset(dC, 0xFF, (size_t)M * N * esz));
            CK(hipDeviceSynchronize());
            run();
            fetch(h1);
            size_t diff = 0, first = 0;
            for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) { if (!diff) first = i; ++diff; }
            printf("bitwise against %s: %zu differing bytes of %zu%s\n", refk, diff, h0.size(), diff ? "  <-- MISMATCH" : "");
            if (diff) printf("  first at element %zu (row %zu, col %zu): %g vs %g\n", first / esz, first / esz / N, first / esz % N, val(h0, first / esz), val(h1, first / esz));
            if (mine) setenv("ZGPT2_GEMM_KERNEL", keep.c_str(), 1); else unsetenv("ZGPT2_GEMM_KERNEL");
            if (diff) return 1;
        }
    }
    for (int i = 0; i < 300; ++i) run();  // clocks settle
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30, sum = 0;
    for (int b = 0; b < batches; ++b) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        best = std::min(best, us); sum += us;
    }
    const double flops = 2.0 * M * N * K, mean = sum / batches;
    printf("time %dx%dx%d %s dbg=%s: min %.2f us (%.1f TF, %.1f%%)  mean %.2f us (%.1f TF, %.1f%% of 2.5 PF)\n", M, N, K,
           getenv("ZGPT2_GEMM_KERNEL") ? getenv("ZGPT2_GEMM_KERNEL") : "default", getenv("ZGPT2_GEMM_DBG") ? getenv("ZGPT2_GEMM_DBG") : "0", best,
           flops / best / 1e6, flops / best / 1e6 / 25.0, mean, flops / mean / 1e6, flops / mean / 1e6 / 25.0);
    if (stamps && zg_debug_gemm_stamps) {
        std::vector<unsigned long long> w(1 + 4 * 256 + 16);
        ZK(zg_debug_gemm_stamps(w.data(), w.size()));
        const int g = (int)std::min<unsigned long long>(w[0], 256);
        // per XCD (workgroup b runs on XCD b % 8; the cycle counter is per XCD): span from the first start to the last end
        double dmin = 1e30, dmax = 0, dsum = 0, span_max = 0;
        for (int x = 0; x < 8 && x < g; ++x) {
            unsigned long long s0 = ~0ull, e1 = 0;
            for (int b = x; b < g; b += 8) { s0 = std::min(s0, w[1 + 2 * b]); e1 = std::max(e1, w[2 + 2 * b]); }
            span_max = std::max(span_max, (double)(e1 - s0));
        }
        for (int b = 0; b < g; ++b) { const double d = (double)(w[2 + 2 * b] - w[1 + 2 * b]); dmin = std::min(dmin, d); dmax = std::max(dmax, d); dsum += d; }
        // wall time inside the launch (100 MHz s_memrealtime: one counter for the whole chip): first start .. last end
        unsigned long long ws = ~0ull, we = 0, wls = 0;
        double wsum = 0;
        for (int b = 0; b < g; ++b) { ws = std::min(ws, w[513 + 2 * b]); we = std::max(we, w[514 + 2 * b]); wls = std::max(wls, w[513 + 2 * b]); wsum += (double)(w[514 + 2 * b] - w[513 + 2 * b]); }
        const double wall_us = (double)(we - ws) / 100.0, wg_us = wsum / g / 100.0;
        printf("stamps: %d workgroups, cycles min %.0f avg %.0f max %.0f; widest XCD span %.0f cycles\n", g, dmin, dsum / g, dmax, span_max);
        printf("stamps: wall inside the launch %.2f us (first start .. last end; last start %.2f us after the first), mean workgroup %.2f us -> shader clock %.2f GHz; "
               "launch-to-launch %.2f us -> %.2f us outside the workgroups\n", wall_us, (double)(wls - ws) / 100.0, wg_us, dsum / g / wg_us / 1e3, mean, mean - wall_us);
        printf("stamps: workgroup 0 phases (cycles since start):");
        for (int i = 1; i < 16 && w[1025 + i]; ++i) printf(" %llu", w[1025 + i] - w[1025]);
        printf("\n");
    }
    return 0;
}
