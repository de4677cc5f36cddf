// zgpt2_ops.hpp — C++ host-side mirror of the reference's src/ops.zig public interface over the C ABI
// of zgpt2.h.  The reference's host language (Zig) has no toolchain in the build environment, so
// this header is the compiled-language caller of the drop-in boundary: same decl names, the same
// fields, `forward` with the same argument order and meaning, slices as (ptr, len) pairs, ops that
// never allocate, `void` returns (a non-zero status throws, where Zig would panic).
#pragma once
#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>

#include "zgpt2.h"

namespace ops {

// A Zig slice: pointer + element count.
template <typename T>
struct Slice {
    T* ptr = nullptr;
    size_t len = 0;
    Slice() = default;
    Slice(T* p, size_t n) : ptr(p), len(n) {}
    template <typename C, typename = decltype(std::declval<C&>().data())>
    Slice(C& c) : ptr(c.data()), len(c.size()) {}  // NOLINT: std::vector / std::array
    Slice sub(size_t begin, size_t end) const { return Slice(ptr + begin, end - begin); }
    operator Slice<const T>() const { return Slice<const T>(ptr, len); }
};

inline void check(int status) {
    if (status != ZG_OK) throw std::runtime_error(std::string("libzgpt2_hip: ") + zg_last_error());
}

// ops.Linear — src/ops.zig:4-47
struct Linear {
    size_t in_features = 0, out_features = 0;
    Slice<const float> weight;  // [out_features, in_features]
    Slice<const float> bias;    // ptr == nullptr: no bias

    static Linear init(size_t in_features, size_t out_features, Slice<const float> weight, Slice<const float> bias) {
        check(zg_register_tensor(weight.ptr, weight.len));
        if (bias.ptr) check(zg_register_tensor(bias.ptr, bias.len));
        return Linear{in_features, out_features, weight, bias};
    }
    // the reference frees borrowed weights with `defer allocator.free` (src/tests.zig); the mirror goes with them
    void deinit() const {
        zg_unregister_tensor(weight.ptr);
        if (bias.ptr) zg_unregister_tensor(bias.ptr);
    }
    void forward(Slice<const float> inputs, Slice<float> outputs) const {
        check(zg_linear_forward(in_features, out_features, weight.ptr, bias.ptr, inputs.ptr, inputs.len, outputs.ptr,
                                outputs.len));
    }
};

// ops.Embedding — src/ops.zig:49-68
struct Embedding {
    size_t emb_dim = 0;
    Slice<const float> weight;
    static Embedding init(size_t emb_dim, Slice<const float> weight) {
        check(zg_register_tensor(weight.ptr, weight.len));
        return Embedding{emb_dim, weight};
    }
    void deinit() const { zg_unregister_tensor(weight.ptr); }
    void forward(Slice<const size_t> idxs, Slice<float> embeddings) const {
        check(zg_embedding_forward(emb_dim, weight.ptr, weight.len, idxs.ptr, idxs.len, embeddings.ptr, embeddings.len));
    }
};

// ops.LayerNorm — src/ops.zig:70-105
struct LayerNorm {
    size_t n_features = 0;
    Slice<const float> weight, bias;
    float eps = 1e-5f;
    static LayerNorm init(size_t n_features, Slice<const float> weight, Slice<const float> bias) {
        check(zg_register_tensor(weight.ptr, weight.len));
        check(zg_register_tensor(bias.ptr, bias.len));
        LayerNorm l;
        l.n_features = n_features;
        l.weight = weight;
        l.bias = bias;
        return l;
    }
    void deinit() const {
        zg_unregister_tensor(weight.ptr);
        zg_unregister_tensor(bias.ptr);
    }
    void forward(Slice<float> inputs) const {
        check(zg_layernorm_forward(n_features, weight.ptr, bias.ptr, eps, inputs.ptr, inputs.len));
    }
};

// ops.CausalSelfAttention — src/ops.zig:107-217
struct CausalSelfAttention {
    size_t n_heads = 0, n_embed = 0, head_dim = 0;
    Linear c_attn, c_proj;
    static CausalSelfAttention init(size_t n_heads, size_t n_embed, Linear c_attn, Linear c_proj) {
        return CausalSelfAttention{n_heads, n_embed, n_embed / n_heads, c_attn, c_proj};
    }
    void forward(size_t seq_len, Slice<const float> inputs, Slice<float> k_cache, Slice<float> v_cache,
                 Slice<float> outputs, Slice<float> _qkv, Slice<float> _q, Slice<float> _k, Slice<float> _v,
                 Slice<float> _attn) const {
        check(zg_attn_forward(n_heads, n_embed, c_attn.weight.ptr, c_attn.bias.ptr, c_proj.weight.ptr, c_proj.bias.ptr,
                              seq_len, inputs.ptr, inputs.len, k_cache.ptr, k_cache.len, v_cache.ptr, v_cache.len,
                              outputs.ptr, outputs.len, _qkv.ptr, _qkv.len, _q.ptr, _q.len, _k.ptr, _k.len, _v.ptr,
                              _v.len, _attn.ptr, _attn.len));
    }
    void split_qkv(size_t seq_len, Slice<const float> inputs, size_t split_idx, Slice<float> outputs) const {
        check(zg_split_qkv(n_embed, seq_len, inputs.ptr, inputs.len, split_idx, outputs.ptr, outputs.len));
    }
    static void transpose(const size_t (&shape)[3], Slice<const float> inputs, Slice<float> outputs) {
        check(zg_transpose(shape[0], shape[1], shape[2], inputs.ptr, inputs.len, outputs.ptr, outputs.len));
    }
};

inline void gelu(Slice<float> inputs) { check(zg_gelu(inputs.ptr, inputs.len)); }        // src/ops.zig:221-228
inline void softmax(Slice<float> inputs) { check(zg_softmax(inputs.ptr, inputs.len)); }  // src/ops.zig:231-241

// src/ops.zig:249-307
inline void scaled_dot_product_attention(Slice<const float> q, Slice<const float> k, Slice<const float> v,
                                         size_t n_heads, size_t seq_len, size_t head_dim, Slice<float> outputs,
                                         Slice<float> _attn) {
    check(zg_scaled_dot_product_attention(q.ptr, q.len, k.ptr, k.len, v.ptr, v.len, n_heads, seq_len, head_dim,
                                          outputs.ptr, outputs.len, _attn.ptr, _attn.len));
}

}  // namespace ops
