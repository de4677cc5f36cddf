// zg_runtime.h — process-wide runtime state of libzgpt2_hip (device, stream, host-staging arena).
#pragma once
#include <unordered_map>

#include "zg_kernels.h"

namespace zg {

struct Registered {
    void* dev;
    size_t bytes;
};

struct Ctx {
    bool inited = false;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the stream every launch uses (own_stream unless zg_set_stream)
    char* stage = nullptr;         // device staging arena for host-pointer callers (op tier)
    size_t stage_cap = 0;
    size_t stage_off = 0;
    int* d_flag = nullptr;         // device int used for index-range checks
    float* d_zero = nullptr;       // 64 zero bytes (stand-in operand for absent bias / residual)
    float* attn_part = nullptr;    // op-tier attention partials (sized at init)
    size_t attn_part_floats = 0;
    std::unordered_map<const void*, Registered> registry;
    unsigned long long* dbg = nullptr;  // -DZG_STAMPS diagnostic buffer
};

Ctx& ctx();
int require_init();
bool is_device_ptr(const void* p);

// Per-call staging scope for the op tier: host pointers are copied into the arena (inputs) and
// copied back on finish (outputs); device pointers pass through untouched.
class Call {
  public:
    Call();
    // Activations (inputs, indices, q / k / v): always the caller's bytes of THIS call.
    template <typename T>
    int in(const T* p, size_t n, const T** dev) {
        return stage_in(p, n * sizeof(T), false, reinterpret_cast<const void**>(dev));
    }
    // Borrowed parameters (weights, biases, LayerNorm vectors): a zg_register_tensor mirror of exactly this
    // pointer AND size is used instead of staging; anything else is staged like an activation.
    template <typename T>
    int param(const T* p, size_t n, const T** dev) {
        return stage_in(p, n * sizeof(T), true, reinterpret_cast<const void**>(dev));
    }
    // Device scratch from the same arena (released by finish()).
    template <typename T>
    int scratch(size_t n, T** dev) {
        return alloc(n * sizeof(T), reinterpret_cast<void**>(dev));
    }
    size_t arena_left() const;
    template <typename T>
    int out(T* p, size_t n, T** dev) {
        return stage_out(p, n * sizeof(T), false, reinterpret_cast<void**>(dev));
    }
    template <typename T>
    int inout(T* p, size_t n, T** dev) {
        return stage_out(p, n * sizeof(T), true, reinterpret_cast<void**>(dev));
    }
    // Copies outputs back (if any were staged), synchronises the stream, releases the arena.
    int finish();
    hipStream_t stream() const { return s_; }

  private:
    int stage_in(const void* p, size_t bytes, bool is_param, const void** dev);
    int stage_out(void* p, size_t bytes, bool copy_in, void** dev);
    int alloc(size_t bytes, void** dev);
    struct Out {
        void* host;
        void* dev;
        size_t bytes;
    };
    Out outs_[16];
    int n_outs_ = 0;
    size_t mark_;
    hipStream_t s_;
};

}  // namespace zg
