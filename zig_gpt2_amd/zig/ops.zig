//! Drop-in replacement for zig_gpt2's `src/ops.zig` that forwards every op to libzgpt2_hip.so
//! (hand-written HIP kernels for MI355X / gfx950) through the C ABI declared in include/zgpt2.h.
//!
//! Every public decl of the reference file keeps its name, fields and `forward` signature, so
//! `src/main.zig` and `src/tests.zig` compile against this file unchanged
//! (`const ops = @import("ops.zig");`).  Slices are passed as (ptr, len); ops still allocate
//! nothing.  Status codes are turned into panics because the reference ops return `void`.
//!
//! NOTE: written for Zig 0.11 (the reference's dialect) but UNCOMPILED — there is no zig toolchain
//! in the build environment.  The executed callers of the same C ABI are the Python mirror
//! (zig_gpt2_amd/ops.py, zig_gpt2_amd/gpt.py) and the tests.  Call `ops.init(0)` once at the top
//! of `main` (or rely on the lazy init in `check`), and link with
//!     exe.addLibraryPath(.{ .path = "zig_gpt2_amd/lib" });
//!     exe.linkSystemLibrary("zgpt2_hip");
//! in place of `exe.linkFramework("Accelerate")` (build.zig:30, :71).
//!
//! What the library does behind these calls so that an unchanged main.zig is not slow (include/zgpt2.h, INTEGRATION.md §4):
//! weights are mirrored on the device once (`zg_register_tensor` in `init`); the small activation slices of `State` travel through
//! a pinned arena (kernels that touch a slice once work on it in place); the k/v caches a `Block` owns are mirrored on the device,
//! keyed by their address, so `CausalSelfAttention.forward` uploads nothing while `seq_len` advances by one and returns only the
//! appended row.  A caller that rewrites cache rows it has already handed over must call `zg_unregister_tensor(cache.ptr)` first
//! (main.zig never does).  The C++ restatement of main.zig over the same ABI runs ~0.8 k tokens/s at GPT-2 124M.
const std = @import("std");

const c = struct {
    extern "c" fn zg_init(device: c_int) c_int;
    extern "c" fn zg_last_error() [*:0]const u8;
    extern "c" fn zg_register_tensor(host_ptr: [*]const f32, len: usize) c_int;
    extern "c" fn zg_unregister_tensor(host_ptr: [*]const f32) c_int;
    extern "c" fn zg_linear_forward(in_features: usize, out_features: usize, weight: [*]const f32, bias: ?[*]const f32, inputs: [*]const f32, inputs_len: usize, outputs: [*]f32, outputs_len: usize) c_int;
    extern "c" fn zg_embedding_forward(emb_dim: usize, weight: [*]const f32, weight_len: usize, idxs: [*]const usize, idxs_len: usize, embeddings: [*]f32, embeddings_len: usize) c_int;
    extern "c" fn zg_layernorm_forward(n_features: usize, weight: [*]const f32, bias: [*]const f32, eps: f32, inputs: [*]f32, inputs_len: usize) c_int;
    extern "c" fn zg_attn_forward(n_heads: usize, n_embed: usize, c_attn_w: [*]const f32, c_attn_b: ?[*]const f32, c_proj_w: [*]const f32, c_proj_b: ?[*]const f32, seq_len: usize, inputs: [*]const f32, inputs_len: usize, k_cache: [*]f32, k_cache_len: usize, v_cache: [*]f32, v_cache_len: usize, outputs: [*]f32, outputs_len: usize, _qkv: [*]f32, _qkv_len: usize, _q: [*]f32, _q_len: usize, _k: [*]f32, _k_len: usize, _v: [*]f32, _v_len: usize, _attn: [*]f32, _attn_len: usize) c_int;
    extern "c" fn zg_split_qkv(n_embed: usize, seq_len: usize, inputs: [*]const f32, inputs_len: usize, split_idx: usize, outputs: [*]f32, outputs_len: usize) c_int;
    extern "c" fn zg_transpose(seq_len: usize, n_heads: usize, head_dim: usize, inputs: [*]const f32, inputs_len: usize, outputs: [*]f32, outputs_len: usize) c_int;
    extern "c" fn zg_scaled_dot_product_attention(q: [*]const f32, q_len: usize, k: [*]const f32, k_len: usize, v: [*]const f32, v_len: usize, n_heads: usize, seq_len: usize, head_dim: usize, outputs: [*]f32, outputs_len: usize, _attn: [*]f32, _attn_len: usize) c_int;
    extern "c" fn zg_gelu(inputs: [*]f32, inputs_len: usize) c_int;
    extern "c" fn zg_softmax(inputs: [*]f32, inputs_len: usize) c_int;
};

var initialised: bool = false;

/// Select the GPU.  Optional: the first op call initialises device 0.
pub fn init(device: c_int) void {
    if (c.zg_init(device) != 0) @panic(std.mem.span(c.zg_last_error()));
    initialised = true;
}

fn check(status: c_int) void {
    if (status != 0) @panic(std.mem.span(c.zg_last_error()));
}

fn ensureInit() void {
    if (!initialised) init(0);
}

pub const Linear = struct {
    const Self = @This();

    in_features: usize,
    out_features: usize,
    weight: []const f32, // [out_features, in_features] row-major, as in the reference
    bias: ?[]const f32,

    pub fn init(in_features: usize, out_features: usize, weight: []const f32, bias: ?[]const f32) Self {
        ensureInit();
        // Weights are borrowed for the life of the model (src/main.zig:349-351): keep one device
        // mirror per host tensor instead of re-staging it on every forward.
        check(c.zg_register_tensor(weight.ptr, weight.len));
        if (bias) |b| check(c.zg_register_tensor(b.ptr, b.len));
        return Self{ .in_features = in_features, .out_features = out_features, .weight = weight, .bias = bias };
    }

    /// Not in the reference (its structs only borrow): drops the device mirrors before the caller frees the
    /// weights (`defer allocator.free(...)` in src/tests.zig), so that a reused host address can never meet a
    /// stale mirror.  Optional: `init` on a new tensor at the same address replaces the mirror anyway.
    pub fn deinit(self: Self) void {
        check(c.zg_unregister_tensor(self.weight.ptr));
        if (self.bias) |b| check(c.zg_unregister_tensor(b.ptr));
    }

    pub fn forward(self: Self, inputs: []const f32, outputs: []f32) void {
        check(c.zg_linear_forward(self.in_features, self.out_features, self.weight.ptr, if (self.bias) |b| b.ptr else null, inputs.ptr, inputs.len, outputs.ptr, outputs.len));
    }
};

pub const Embedding = struct {
    const Self = @This();

    emb_dim: usize,
    weight: []const f32,

    pub fn init(emb_dim: usize, weight: []const f32) Self {
        ensureInit();
        check(c.zg_register_tensor(weight.ptr, weight.len));
        return Self{ .emb_dim = emb_dim, .weight = weight };
    }

    pub fn forward(self: Self, idxs: []const usize, embeddings: []f32) void {
        check(c.zg_embedding_forward(self.emb_dim, self.weight.ptr, self.weight.len, idxs.ptr, idxs.len, embeddings.ptr, embeddings.len));
    }
};

pub const LayerNorm = struct {
    const Self = @This();

    n_features: usize,
    weight: []const f32,
    bias: []const f32,
    eps: f32 = 1e-5,

    pub fn init(n_features: usize, weight: []const f32, bias: []const f32) Self {
        ensureInit();
        check(c.zg_register_tensor(weight.ptr, weight.len));
        check(c.zg_register_tensor(bias.ptr, bias.len));
        return Self{ .n_features = n_features, .weight = weight, .bias = bias };
    }

    pub fn forward(self: Self, inputs: []f32) void {
        check(c.zg_layernorm_forward(self.n_features, self.weight.ptr, self.bias.ptr, self.eps, inputs.ptr, inputs.len));
    }
};

pub const CausalSelfAttention = struct {
    const Self = @This();

    n_heads: usize,
    n_embed: usize,
    head_dim: usize,
    c_attn: Linear,
    c_proj: Linear,

    pub fn init(n_heads: usize, n_embed: usize, c_attn: Linear, c_proj: Linear) Self {
        return Self{ .n_heads = n_heads, .n_embed = n_embed, .head_dim = n_embed / n_heads, .c_attn = c_attn, .c_proj = c_proj };
    }

    pub fn forward(
        self: Self,
        seq_len: usize,
        inputs: []const f32,
        k_cache: []f32,
        v_cache: []f32,
        outputs: []f32,
        _qkv: []f32,
        _q: []f32,
        _k: []f32,
        _v: []f32,
        _attn: []f32,
    ) void {
        check(c.zg_attn_forward(self.n_heads, self.n_embed, self.c_attn.weight.ptr, if (self.c_attn.bias) |b| b.ptr else null, self.c_proj.weight.ptr, if (self.c_proj.bias) |b| b.ptr else null, seq_len, inputs.ptr, inputs.len, k_cache.ptr, k_cache.len, v_cache.ptr, v_cache.len, outputs.ptr, outputs.len, _qkv.ptr, _qkv.len, _q.ptr, _q.len, _k.ptr, _k.len, _v.ptr, _v.len, _attn.ptr, _attn.len));
    }

    pub fn split_qkv(self: Self, seq_len: usize, inputs: []const f32, split_idx: usize, outputs: []f32) void {
        ensureInit();
        check(c.zg_split_qkv(self.n_embed, seq_len, inputs.ptr, inputs.len, split_idx, outputs.ptr, outputs.len));
    }

    pub fn transpose(shape: [3]usize, inputs: []const f32, outputs: []f32) void {
        ensureInit();
        check(c.zg_transpose(shape[0], shape[1], shape[2], inputs.ptr, inputs.len, outputs.ptr, outputs.len));
    }
};

pub fn gelu(inputs: []f32) void {
    ensureInit();
    check(c.zg_gelu(inputs.ptr, inputs.len));
}

pub fn softmax(inputs: []f32) void {
    ensureInit();
    check(c.zg_softmax(inputs.ptr, inputs.len));
}

pub fn scaled_dot_product_attention(
    q: []const f32,
    k: []const f32,
    v: []const f32,
    n_heads: usize,
    seq_len: usize,
    head_dim: usize,
    outputs: []f32,
    _attn: []f32,
) void {
    ensureInit();
    check(c.zg_scaled_dot_product_attention(q.ptr, q.len, k.ptr, k.len, v.ptr, v.len, n_heads, seq_len, head_dim, outputs.ptr, outputs.len, _attn.ptr, _attn.len));
}

// Host-side file helpers with the reference's names and signatures (src/ops.zig:309-326) so that
// main.zig's loaders and tests.zig keep compiling; they never touch the GPU.  Unlike the reference,
// a short read is reported instead of being ignored.
pub fn load_tensor(path: []const u8, shape: []const usize, comptime dtype: type, allocator: std.mem.Allocator) ![]dtype {
    var count: usize = 1;
    for (shape) |dim| count *= dim;
    const out = try allocator.alloc(dtype, count);
    errdefer allocator.free(out);
    var file = try std.fs.cwd().openFile(path, .{ .mode = .read_only });
    defer file.close();
    const want = std.mem.sliceAsBytes(out);
    const got = try file.readAll(want);
    if (got != want.len) return error.UnexpectedEndOfFile;
    return out;
}

pub fn load_json(path: []const u8, allocator: std.mem.Allocator) !std.json.Value {
    const max_bytes = 4 << 20;
    const text = try std.fs.cwd().readFileAlloc(allocator, path, max_bytes);
    return std.json.parseFromSliceLeaky(std.json.Value, allocator, text, .{});
}
