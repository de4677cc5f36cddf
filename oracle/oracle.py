"""ctypes wrapper over oracle/libzgpt2_oracle.so — TEST INFRASTRUCTURE ONLY.

The product package (zig_gpt2_amd) never imports this module.  Arrays are numpy float32,
C-contiguous, host memory.  Function names mirror src/ops.zig / src/main.zig of the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzgpt2_oracle.so")
_lib = None

f32p = C.POINTER(C.c_float)
szp = C.POINTER(C.c_size_t)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
        os.path.join(_HERE, "zgpt2_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libzgpt2_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def host_cores():
    """Usable host cores: affinity mask, clipped by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
        # tests: a bounded OpenMP team (oversubscribed teams make the many tiny sgemm calls crawl)
        _lib.orc_set_num_threads(min(host_cores(), int(os.environ.get("ZGPT2_ORACLE_THREADS", "16"))))
    return _lib


def _declare(L):
    sz, f, i, u64 = C.c_size_t, C.c_float, C.c_int, C.c_uint64
    L.orc_fill_normal.argtypes = [u64, sz, f, f, i, f32p]
    L.orc_fill_uniform.argtypes = [u64, sz, f, f, i, f32p]
    L.orc_use_cblas.argtypes = [C.c_char_p, C.c_char_p]
    L.orc_use_cblas.restype = i
    L.orc_set_accum_double.argtypes = [i]
    L.orc_num_threads.restype = i
    L.orc_set_num_threads.argtypes = [i]
    L.orc_linear_forward.argtypes = [sz, sz, f32p, f32p, f32p, sz, f32p]
    L.orc_embedding_forward.argtypes = [sz, f32p, szp, sz, f32p]
    L.orc_layernorm_forward.argtypes = [sz, f32p, f32p, f, f32p, sz]
    L.orc_gelu.argtypes = [f32p, sz]
    L.orc_softmax.argtypes = [f32p, sz]
    L.orc_split_qkv.argtypes = [sz, sz, f32p, sz, sz, f32p]
    L.orc_transpose.argtypes = [sz, sz, sz, f32p, sz, f32p]
    L.orc_sdpa.argtypes = [f32p, f32p, sz, f32p, sz, sz, sz, f32p, f32p]
    L.orc_attn_forward.argtypes = [sz, sz, f32p, f32p, f32p, f32p, sz, f32p] + [f32p] * 8
    L.orc_gpt_create.argtypes = [sz] * 5
    L.orc_gpt_create.restype = C.c_void_p
    L.orc_gpt_destroy.argtypes = [C.c_void_p]
    L.orc_gpt_set_block_tensor.argtypes = [C.c_void_p, sz, i, f32p]
    L.orc_gpt_set_block_tensor.restype = i
    L.orc_gpt_set_tensor.argtypes = [C.c_void_p, i, f32p]
    L.orc_gpt_set_tensor.restype = i
    L.orc_gpt_logits.argtypes = [C.c_void_p]
    L.orc_gpt_logits.restype = f32p
    L.orc_gpt_x.argtypes = [C.c_void_p]
    L.orc_gpt_x.restype = f32p
    L.orc_gpt_forward.argtypes = [C.c_void_p, sz, sz, i]
    L.orc_gpt_sample_greedy.argtypes = [C.c_void_p, sz, sz]
    L.orc_gpt_sample_greedy.restype = sz
    L.orc_gpt_sample.argtypes = [C.c_void_p, sz, sz, f, f]
    L.orc_gpt_sample.restype = sz
    L.orc_gpt_generate_greedy.argtypes = [C.c_void_p, szp, sz, sz, szp, f32p]
    L.orc_gpt_forced_logits.argtypes = [C.c_void_p, szp, sz, sz, f32p]


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(f32p)


def _fo(a):
    return None if a is None else _f(a)


# ----------------------------------------------------------------------------- synthetic data
def fill_normal(seed, n, mean=0.0, std=1.0, round_bf16=False):
    out = np.empty(int(n), dtype=np.float32)
    lib().orc_fill_normal(seed, out.size, mean, std, int(round_bf16), _f(out))
    return out


def fill_uniform(seed, n, lo=0.0, hi=1.0, round_bf16=False):
    out = np.empty(int(n), dtype=np.float32)
    lib().orc_fill_uniform(seed, out.size, lo, hi, int(round_bf16), _f(out))
    return out


def use_cblas(path, symbol="cblas_sgemm"):
    return lib().orc_use_cblas(None if path is None else path.encode(), symbol.encode())


def find_cblas():
    """Locate a CBLAS for the timed CPU baseline: (path, symbol, label) or None."""
    import glob

    try:
        import scipy

        for p in glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so")):
            return os.path.abspath(p), "scipy_cblas_sgemm", "OpenBLAS (scipy.libs)"
    except Exception:
        pass
    try:
        for p in glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas*.so")):
            return os.path.abspath(p), "scipy_cblas_sgemm", "OpenBLAS (numpy.libs)"
    except Exception:
        pass
    return None


def set_accum_double(on):
    lib().orc_set_accum_double(int(on))


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ----------------------------------------------------------------------------- src/ops.zig
def linear_forward(in_features, out_features, weight, bias, inputs):
    """Linear.forward (src/ops.zig:21-46). Returns outputs [M, out]."""
    m = inputs.size // in_features
    out = np.empty((m, out_features), dtype=np.float32)
    lib().orc_linear_forward(in_features, out_features, _f(weight), _fo(bias), _f(inputs), inputs.size, _f(out))
    return out


def embedding_forward(emb_dim, weight, idxs):
    idxs = np.ascontiguousarray(idxs, dtype=np.uint64)
    out = np.empty((idxs.size, emb_dim), dtype=np.float32)
    lib().orc_embedding_forward(emb_dim, _f(weight), idxs.ctypes.data_as(szp), idxs.size, _f(out))
    return out


def layernorm_forward(n_features, weight, bias, inputs, eps=1e-5):
    """LayerNorm.forward (src/ops.zig:82-104), in place on a copy; returns the result."""
    x = np.array(inputs, dtype=np.float32, copy=True, order="C")
    lib().orc_layernorm_forward(n_features, _f(weight), _f(bias), eps, _f(x), x.size)
    return x


def gelu(inputs):
    x = np.array(inputs, dtype=np.float32, copy=True, order="C")
    lib().orc_gelu(_f(x), x.size)
    return x


def softmax(inputs):
    """softmax (src/ops.zig:231-241) over the WHOLE array as one vector."""
    x = np.array(inputs, dtype=np.float32, copy=True, order="C")
    lib().orc_softmax(_f(x), x.size)
    return x


def split_qkv(n_embed, seq_len, inputs, split_idx):
    out = np.empty(inputs.size // 3, dtype=np.float32)
    lib().orc_split_qkv(n_embed, seq_len, _f(inputs), inputs.size, split_idx, _f(out))
    return out


def transpose(seq_len, n_heads, head_dim, inputs):
    out = np.empty(inputs.size, dtype=np.float32)
    lib().orc_transpose(seq_len, n_heads, head_dim, _f(inputs), inputs.size, _f(out))
    return out


def sdpa(q, k, v, n_heads, seq_len, head_dim):
    batch = k.size // (n_heads * seq_len * head_dim)
    out = np.empty(batch * n_heads * head_dim, dtype=np.float32)
    attn = np.empty(seq_len, dtype=np.float32)
    lib().orc_sdpa(_f(q), _f(k), k.size, _f(v), n_heads, seq_len, head_dim, _f(out), _f(attn))
    return out


class CausalSelfAttention:
    """Decode-step attention with caller-owned caches, mirroring src/ops.zig:107-173."""

    def __init__(self, n_heads, n_embed, c_attn_w, c_attn_b, c_proj_w, c_proj_b, context_size):
        self.n_heads, self.n_embed = n_heads, n_embed
        self.w = [np.ascontiguousarray(a, dtype=np.float32) for a in (c_attn_w, c_attn_b, c_proj_w, c_proj_b)]
        e = n_embed
        self.k_cache = np.zeros(context_size * e, np.float32)
        self.v_cache = np.zeros(context_size * e, np.float32)
        self._qkv = np.zeros(3 * e, np.float32)
        self._q = np.zeros(e, np.float32)
        self._k = np.zeros(context_size * e, np.float32)
        self._v = np.zeros(context_size * e, np.float32)
        self._attn = np.zeros(context_size, np.float32)

    def forward(self, seq_len, inputs):
        out = np.empty(self.n_embed, np.float32)
        lib().orc_attn_forward(
            self.n_heads, self.n_embed, _f(self.w[0]), _f(self.w[1]), _f(self.w[2]), _f(self.w[3]), seq_len,
            _f(np.ascontiguousarray(inputs, dtype=np.float32)), _f(self.k_cache), _f(self.v_cache), _f(out),
            _f(self._qkv), _f(self._q), _f(self._k), _f(self._v), _f(self._attn),
        )
        return out


# ----------------------------------------------------------------------------- src/main.zig
BLOCK_SLOTS = [
    "ln_1_g", "ln_1_b", "c_attn_w", "c_attn_b", "c_proj_w", "c_proj_b",
    "ln_2_g", "ln_2_b", "c_fc_w", "c_fc_b", "mlp_proj_w", "mlp_proj_b",
]
TOP_SLOTS = ["wte", "wpe", "ln_f_g", "ln_f_b"]


class GPT:
    """GPT over borrowed fp32 weights (dict from zig_gpt2_amd.synth.make_weights)."""

    def __init__(self, cfg, weights):
        self.cfg = cfg
        self.weights = weights  # keep alive: the C side borrows the pointers
        L = lib()
        self.h = L.orc_gpt_create(cfg.vocab_size, cfg.context_size, cfg.n_layer, cfg.n_heads, cfg.n_embed)
        for s, name in enumerate(TOP_SLOTS):
            assert L.orc_gpt_set_tensor(self.h, s, _f(weights[name])) == 0
        for l in range(cfg.n_layer):
            for s, name in enumerate(BLOCK_SLOTS):
                assert L.orc_gpt_set_block_tensor(self.h, l, s, _f(weights[f"h{l}.{name}"])) == 0

    def __del__(self):
        try:
            lib().orc_gpt_destroy(self.h)
        except Exception:
            pass

    def forward(self, seq_len, token, compute_logits=True):
        lib().orc_gpt_forward(self.h, seq_len, token, int(compute_logits))
        if compute_logits:
            return np.ctypeslib.as_array(lib().orc_gpt_logits(self.h), (self.cfg.vocab_size,)).copy()
        return None

    def sample(self, seq_len, token, temp, u):
        """GPT.sample (src/main.zig:198-207) with the uniform draw u supplied; returns (token, probs)."""
        t = lib().orc_gpt_sample(self.h, seq_len, token, temp, u)
        return int(t), np.ctypeslib.as_array(lib().orc_gpt_logits(self.h), (self.cfg.vocab_size,)).copy()

    def hidden(self):
        return np.ctypeslib.as_array(lib().orc_gpt_x(self.h), (self.cfg.n_embed,)).copy()

    def generate_greedy(self, prompt, n_steps, want_logits=False):
        prompt = np.ascontiguousarray(prompt, dtype=np.uint64)
        out = np.zeros(n_steps, dtype=np.uint64)
        logits = None
        if want_logits:
            logits = np.zeros((n_steps - prompt.size, self.cfg.vocab_size), np.float32)
        lib().orc_gpt_generate_greedy(
            self.h, prompt.ctypes.data_as(szp), prompt.size, n_steps, out.ctypes.data_as(szp), _fo(logits)
        )
        return (out, logits) if want_logits else out

    def forced_logits(self, forced, first_logit_step):
        forced = np.ascontiguousarray(forced, dtype=np.uint64)
        n = forced.size
        logits = np.zeros((n - first_logit_step, self.cfg.vocab_size), np.float32)
        lib().orc_gpt_forced_logits(self.h, forced.ctypes.data_as(szp), n, first_logit_step, _f(logits))
        return logits
