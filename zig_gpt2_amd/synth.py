"""Synthetic, bit-reproducible GPT-2 weights and inputs (no checkpoints exist offline).

A counter-based integer PRNG (splitmix64 finaliser) whose only floating-point steps are one fp32
multiply and one fp32 add, so numpy here and plain C in the test oracle produce identical bits.
Distributions follow SURVEY.md §8(d): every tensor ~ N(mean, 0.02^2) (LayerNorm gains have
mean 1), optionally rounded to the nearest bf16 so that the fp32 reference arithmetic and the
bf16-storage device path see exactly the same weight values.
"""
from dataclasses import dataclass

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_IH4_STD = 37837.22753904532  # std of the sum of four independent 16-bit uniforms


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _key(seed):
    with np.errstate(over="ignore"):
        return _mix64(np.uint64(seed) + _GOLD)


def _rand_u64(seed, n, offset=0):
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        return _mix64(_key(seed) + i * _GOLD)


def round_bf16(x):
    """Round fp32 array to the nearest bf16 (ties to even), returned as fp32."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    b = (b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return b.view(np.float32)


def to_bf16_bits(x):
    """fp32 array -> uint16 bf16 bit patterns (RNE)."""
    return (round_bf16(x).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def _chunks(n, step=1 << 24):
    for o in range(0, n, step):
        yield o, min(step, n - o)


def fill_normal(seed, n, mean=0.0, std=1.0, bf16=False):
    n = int(n)
    out = np.empty(n, dtype=np.float32)
    scale = np.float32(float(std) / _IH4_STD)
    mean = np.float32(mean)
    m16 = np.uint64(0xFFFF)
    for o, c in _chunks(n):
        r = _rand_u64(seed, c, o)
        s = (r & m16) + ((r >> np.uint64(16)) & m16) + ((r >> np.uint64(32)) & m16) + (r >> np.uint64(48))
        v = (s.astype(np.int64) - 131070).astype(np.float32) * scale
        v = v + mean
        out[o : o + c] = v
    return round_bf16(out) if bf16 else out


def fill_uniform(seed, n, lo=0.0, hi=1.0, bf16=False):
    n = int(n)
    out = np.empty(n, dtype=np.float32)
    lo = np.float32(lo)
    width = np.float32(hi) - lo
    for o, c in _chunks(n):
        r = _rand_u64(seed, c, o)
        u = (r >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
        v = u * width
        out[o : o + c] = v + lo
    return round_bf16(out) if bf16 else out


def rand_tokens(seed, n, vocab_size):
    return (_rand_u64(seed, n) % np.uint64(vocab_size)).astype(np.uint64)


@dataclass(frozen=True)
class GPTConfig:
    """Field-for-field mirror of GPTConfig in src/main.zig:5-23."""

    vocab_size: int
    context_size: int
    n_layer: int
    n_heads: int
    n_embed: int

    @property
    def head_dim(self):
        return self.n_embed // self.n_heads


# The configurations BASELINE.json names (XL and nano-char sizes are not in the reference).
CONFIGS = {
    "124M": GPTConfig(50257, 1024, 12, 12, 768),  # src/main.zig:346
    "xl": GPTConfig(50257, 1024, 48, 25, 1600),
    "nano-char": GPTConfig(65, 256, 6, 6, 384),
    "tiny": GPTConfig(257, 64, 2, 2, 128),  # test-only
    "tiny3": GPTConfig(131, 48, 3, 3, 192),  # test-only, odd sizes
    "xl-slice": GPTConfig(1031, 96, 2, 25, 1600),  # test-only: GPT-2 XL's layer shapes (E = 1600, 25 heads), 2 layers
    "max-slice": GPTConfig(515, 72, 1, 32, 2048),  # test-only: the widest supported model (4 E = 8192), 1 layer
    "medium-slice": GPTConfig(521, 80, 2, 16, 1024),  # test-only: GPT-2 medium's layer shapes (E = 1024, 16 heads), 2 layers
}

BLOCK_TENSORS = [
    # name, shape fn (E), mean
    ("ln_1_g", lambda e: (e,), 1.0),
    ("ln_1_b", lambda e: (e,), 0.0),
    ("c_attn_w", lambda e: (3 * e, e), 0.0),
    ("c_attn_b", lambda e: (3 * e,), 0.0),
    ("c_proj_w", lambda e: (e, e), 0.0),
    ("c_proj_b", lambda e: (e,), 0.0),
    ("ln_2_g", lambda e: (e,), 1.0),
    ("ln_2_b", lambda e: (e,), 0.0),
    ("c_fc_w", lambda e: (4 * e, e), 0.0),
    ("c_fc_b", lambda e: (4 * e,), 0.0),
    ("mlp_proj_w", lambda e: (e, 4 * e), 0.0),
    ("mlp_proj_b", lambda e: (e,), 0.0),
]


def tensor_specs(cfg):
    """[(name, shape, mean, tensor_index)] — Linear weights are [out, in] like ops.Linear.weight."""
    e = cfg.n_embed
    specs = [
        ("wte", (cfg.vocab_size, e), 0.0, 0),
        ("wpe", (cfg.context_size, e), 0.0, 1),
        ("ln_f_g", (e,), 1.0, 2),
        ("ln_f_b", (e,), 0.0, 3),
    ]
    for l in range(cfg.n_layer):
        for s, (name, shp, mean) in enumerate(BLOCK_TENSORS):
            specs.append((f"h{l}.{name}", shp(e), mean, 16 + 16 * l + s))
    return specs


def make_weights(cfg, seed=0, bf16=True, std=0.02):
    """All model tensors as fp32 numpy arrays (bf16-representable when bf16=True)."""
    w = {}
    for name, shape, mean, idx in tensor_specs(cfg):
        n = int(np.prod(shape))
        w[name] = fill_normal(seed * 4096 + idx, n, mean=mean, std=std, bf16=bf16).reshape(shape)
    return w
