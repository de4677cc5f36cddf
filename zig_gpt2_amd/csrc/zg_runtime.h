// zg_runtime.h — process-wide runtime state of libzgpt2_hip (device, stream, host-staging arena).
#pragma once
#include <unordered_map>

#include "zg_kernels.h"

namespace zg {

struct Registered {
    void* dev;
    size_t bytes;
};

// Device mirror of a caller-owned host KV cache (zg_attn_forward, ops.zig:129-173): the caller appends one row per call and
// never reads the cache itself, so the rows it handed over once stay on the device (keyed by the cache's host address).
struct KvMirror {
    float* dev = nullptr;
    size_t cap_floats = 0;
    size_t n_embed = 0;
    size_t rows = 0;  // rows 0 .. rows-1 of the mirror equal the caller's cache
};

struct Ctx {
    bool inited = false;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the stream every launch uses (own_stream unless zg_set_stream)
    char* stage = nullptr;         // device staging arena for host-pointer callers (op tier)
    size_t stage_cap = 0;
    size_t stage_off = 0;
    // pinned host arena of the op tier: the caller's pageable buffers are copied here by the CPU, and the GPU either reads /
    // writes this memory in place (buffers a kernel touches once) or moves it with ONE asynchronous DMA each way
    char* pin = nullptr;
    size_t pin_cap = 0;
    size_t pin_off = 0;
    // device pool of the KV-cache mirrors (allocated in zg_init: no forward allocates)
    char* kv_pool = nullptr;
    size_t kv_pool_cap = 0;
    size_t kv_pool_off = 0;
    std::unordered_map<const void*, KvMirror> kv_mirrors;
    // the result of the last in-place elementwise op on a HOST buffer, kept twice: in device memory (what the next Linear reads
    // instead of uploading the same bytes again — in src/main.zig every Linear's input is the previous op's output) and in host
    // memory (what the caller's buffer is compared with before the device twin is trusted)
    float* shadow_dev = nullptr;
    char* shadow_host = nullptr;
    size_t shadow_cap = 0;
    const void* shadow_ptr = nullptr;
    size_t shadow_bytes = 0;
    unsigned* done_flag = nullptr;  // pinned completion word of op-tier calls (d_flag + 8), its sequence number, calls since a real drain
    unsigned done_seq = 0;
    unsigned calls_since_sync = 0;
    int* d_flag = nullptr;         // pinned host int the kernels raise for index-range checks (read after the call's drain)
    float* d_zero = nullptr;       // 64 zero bytes (stand-in operand for absent bias / residual)
    float* attn_part = nullptr;    // op-tier attention partials (sized at init)
    size_t attn_part_floats = 0;
    std::unordered_map<const void*, Registered> registry;
    unsigned long long* dbg = nullptr;  // -DZG_STAMPS diagnostic buffer
};

Ctx& ctx();
int require_init();
bool is_device_ptr(const void* p);

// Per-call staging scope for the op tier.  Device pointers pass through untouched.  Host pointers:
//   in / out / inout      the buffer must live in device memory while the kernels run (read by many workgroups, or read back):
//                         CPU copy into the pinned arena, one asynchronous DMA into the device arena (and back at finish)
//   *_once                a kernel reads / writes every element ONCE: the kernel works on the pinned arena in place (PCIe
//                         zero-copy), no DMA at all — the common case for the small activation vectors of src/main.zig:119-195
// Buffers that do not fit the pinned arena go through hipMemcpyAsync on the caller's pageable memory, as before.
class Call {
  public:
    Call();
    // Activations (inputs, indices, q / k / v): always the caller's bytes of THIS call.
    template <typename T>
    int in(const T* p, size_t n, const T** dev) {
        return stage_in(p, n * sizeof(T), false, false, reinterpret_cast<const void**>(dev));
    }
    template <typename T>
    int in_once(const T* p, size_t n, const T** dev) {
        return stage_in(p, n * sizeof(T), false, true, reinterpret_cast<const void**>(dev));
    }
    // Borrowed parameters (weights, biases, LayerNorm vectors): a zg_register_tensor mirror of exactly this
    // pointer AND size is used instead of staging; anything else is staged like an activation.
    template <typename T>
    int param(const T* p, size_t n, const T** dev) {
        return stage_in(p, n * sizeof(T), true, false, reinterpret_cast<const void**>(dev));
    }
    // Device scratch from the same arena (released by finish()).
    template <typename T>
    int scratch(size_t n, T** dev) {
        return alloc(n * sizeof(T), reinterpret_cast<void**>(dev));
    }
    size_t arena_left() const;
    template <typename T>
    int out(T* p, size_t n, T** dev) {
        return stage_out(p, n * sizeof(T), false, false, reinterpret_cast<void**>(dev));
    }
    template <typename T>
    int out_once(T* p, size_t n, T** dev) {
        return stage_out(p, n * sizeof(T), false, true, reinterpret_cast<void**>(dev));
    }
    template <typename T>
    int inout(T* p, size_t n, T** dev) {
        return stage_out(p, n * sizeof(T), true, false, reinterpret_cast<void**>(dev));
    }
    template <typename T>
    int inout_once(T* p, size_t n, T** dev) {
        return stage_out(p, n * sizeof(T), true, true, reinterpret_cast<void**>(dev));
    }
    // A second host destination for a part of an output already staged with out() (the cache rows inside _qkv).
    int also_out(void* host, const void* dev_part, size_t bytes);
    // A call whose LAST kernel is a single workgroup writing zero-copy outputs may let that kernel store the completion word
    // itself: reserve() hands out the word and the sequence number finish() will poll for (flag == nullptr: not available —
    // polling is off, the periodic real drain is due, or an output of this call still needs a DMA behind the kernel);
    // announced(true) tells finish() that the kernel took it, so the one-thread launch is skipped.
    void reserve(unsigned** flag, unsigned* seq);
    void announced(bool yes) { announced_ = yes; }
    // Copies outputs back (if any were staged), synchronises the stream, releases the arenas.
    int finish();
    hipStream_t stream() const { return s_; }

  private:
    int stage_in(const void* p, size_t bytes, bool is_param, bool once, const void** dev);
    int stage_out(void* p, size_t bytes, bool copy_in, bool once, void** dev);
    int alloc(size_t bytes, void** dev);
    char* pin_alloc(size_t bytes);
    struct Out {
        void* host;
        void* dev;   // device buffer to copy from (nullptr: the pinned slot already holds the result)
        char* pin;   // pinned slot (nullptr: pageable path, straight into host)
        size_t bytes;
    };
    Out outs_[24];
    int n_outs_ = 0;
    unsigned reserved_seq_ = 0;
    bool reserved_ = false, announced_ = false;
    size_t mark_, pin_mark_;
    hipStream_t s_;
};

}  // namespace zg
