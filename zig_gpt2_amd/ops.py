"""Python mirror of the reference's src/ops.zig public interface over the C ABI.

Same decl names, field names and `forward` argument order as ops.zig; every "slice" is a numpy
float32 array (host memory) or a torch tensor (host or device memory) that the caller allocates —
ops never allocate, exactly like the reference (README.md:6).  Outputs are written in place into
the caller's buffers.  All compute runs in libzgpt2_hip.so; there is no Python/CPU fallback.
"""
import numpy as np

from . import _lib
from ._lib import check, ptr


def _n(a):
    return 0 if a is None else int(a.numel() if hasattr(a, "numel") else a.size)


def _chk(a, name):
    if a is None:
        return
    if hasattr(a, "is_contiguous"):
        assert a.is_contiguous(), f"{name} must be contiguous"
    else:
        assert a.flags["C_CONTIGUOUS"], f"{name} must be C-contiguous"


class Linear:
    """ops.Linear (src/ops.zig:4-47): weight is [out_features, in_features] row-major."""

    def __init__(self, in_features, out_features, weight, bias=None):
        self.in_features, self.out_features, self.weight, self.bias = in_features, out_features, weight, bias

    def forward(self, inputs, outputs):
        for a, n in ((self.weight, "weight"), (self.bias, "bias"), (inputs, "inputs"), (outputs, "outputs")):
            _chk(a, n)
        check(_lib.load().zg_linear_forward(self.in_features, self.out_features, ptr(self.weight), ptr(self.bias),
                                            ptr(inputs), _n(inputs), ptr(outputs), _n(outputs)))


class Embedding:
    """ops.Embedding (src/ops.zig:49-68); idxs are usize (numpy uint64 / torch int64)."""

    def __init__(self, emb_dim, weight):
        self.emb_dim, self.weight = emb_dim, weight

    def forward(self, idxs, embeddings):
        check(_lib.load().zg_embedding_forward(self.emb_dim, ptr(self.weight), _n(self.weight), ptr(idxs), _n(idxs),
                                               ptr(embeddings), _n(embeddings)))


class LayerNorm:
    """ops.LayerNorm (src/ops.zig:70-105), in place."""

    def __init__(self, n_features, weight, bias, eps=1e-5):
        self.n_features, self.weight, self.bias, self.eps = n_features, weight, bias, eps

    def forward(self, inputs):
        _chk(inputs, "inputs")
        check(_lib.load().zg_layernorm_forward(self.n_features, ptr(self.weight), ptr(self.bias), self.eps,
                                               ptr(inputs), _n(inputs)))


class CausalSelfAttention:
    """ops.CausalSelfAttention (src/ops.zig:107-217)."""

    def __init__(self, n_heads, n_embed, c_attn, c_proj):
        self.n_heads, self.n_embed, self.head_dim = n_heads, n_embed, n_embed // n_heads
        self.c_attn, self.c_proj = c_attn, c_proj

    def forward(self, seq_len, inputs, k_cache, v_cache, outputs, _qkv, _q, _k, _v, _attn):
        L = _lib.load()
        check(L.zg_attn_forward(
            self.n_heads, self.n_embed, ptr(self.c_attn.weight), ptr(self.c_attn.bias), ptr(self.c_proj.weight),
            ptr(self.c_proj.bias), seq_len, ptr(inputs), _n(inputs), ptr(k_cache), _n(k_cache), ptr(v_cache),
            _n(v_cache), ptr(outputs), _n(outputs), ptr(_qkv), _n(_qkv), ptr(_q), _n(_q), ptr(_k), _n(_k),
            ptr(_v), _n(_v), ptr(_attn), _n(_attn)))

    def split_qkv(self, seq_len, inputs, split_idx, outputs):
        check(_lib.load().zg_split_qkv(self.n_embed, seq_len, ptr(inputs), _n(inputs), split_idx, ptr(outputs),
                                       _n(outputs)))

    @staticmethod
    def transpose(shape, inputs, outputs):
        t, n, h = shape
        check(_lib.load().zg_transpose(t, n, h, ptr(inputs), _n(inputs), ptr(outputs), _n(outputs)))


def gelu(inputs):
    """ops.gelu (src/ops.zig:221-228), in place."""
    check(_lib.load().zg_gelu(ptr(inputs), _n(inputs)))


def softmax(inputs):
    """ops.softmax (src/ops.zig:231-241), in place; the whole slice is one vector."""
    check(_lib.load().zg_softmax(ptr(inputs), _n(inputs)))


def scaled_dot_product_attention(q, k, v, n_heads, seq_len, head_dim, outputs, _attn):
    """ops.scaled_dot_product_attention (src/ops.zig:249-307)."""
    check(_lib.load().zg_scaled_dot_product_attention(ptr(q), _n(q), ptr(k), _n(k), ptr(v), _n(v), n_heads, seq_len,
                                                      head_dim, ptr(outputs), _n(outputs), ptr(_attn), _n(_attn)))
