"""Python mirror of the model side of the reference (src/main.zig) over the C ABI.

`GPT` wraps the device-resident model tier (zg_gpt_*): State.init + load_gpt become the
constructor and `load_weights`, GPT.forward keeps its (seq_len, token, compute_logits) meaning,
`generate` is the reference decode loop with greedy argmax in place of the sampler.
`HostGPT` is the op-tier composition: main.zig's State/MLP/Block/GPT written against ops.py
exactly as main.zig is written against ops.zig (used to test the drop-in boundary).
"""
import ctypes as C

import numpy as np

from . import _lib, ops
from ._lib import check, ptr
from .synth import GPTConfig


class GPT:
    def __init__(self, config: GPTConfig, batch=1, weights_f32=False, use_graph=True, kv_f16=False, prefill=True,
                 prefill_planes=3, prefetch=True, kv_b24=False, share_weights_with=None, own_stream=False, stream_priority=0,
                 sampled_generate=False):
        """share_weights_with / own_stream / stream_priority: zg_gpt_options of zg_gpt_create_ex (a handle of an independent
        prompt group on the same GPU: private stream, weight region borrowed from another GPT of the same config)."""
        self.config, self.batch = config, batch
        L = _lib.load()
        flags = (_lib.GPT_WEIGHTS_F32 if weights_f32 else 0) | (0 if use_graph else _lib.GPT_NO_GRAPH)
        flags |= _lib.GPT_KV_F16 if kv_f16 else 0
        flags |= _lib.GPT_KV_B24 if kv_b24 else 0
        flags |= 0 if prefill else _lib.GPT_NO_PREFILL
        flags |= _lib.GPT_PREFILL_2PLANE if prefill_planes == 2 else 0
        flags |= 0 if prefetch else _lib.GPT_NO_PREFETCH
        flags |= _lib.GPT_SAMPLED_GENERATE if sampled_generate else 0
        cfg = _lib.GptConfig(config.vocab_size, config.context_size, config.n_layer, config.n_heads, config.n_embed)
        h = C.c_void_p()
        if share_weights_with is None and not own_stream:
            check(L.zg_gpt_create(C.byref(h), C.byref(cfg), batch, flags))
        else:
            opt = _lib.GptOptions(share_weights_with.h if share_weights_with is not None else None, int(bool(own_stream)), int(stream_priority))
            check(L.zg_gpt_create_ex(C.byref(h), C.byref(cfg), batch, flags, C.byref(opt)))
        self._weight_owner = share_weights_with  # keeps the owner alive: it must be destroyed last
        self.h = h
        self._L = L

    @classmethod
    def from_raw_dir(cls, path, config: GPTConfig, **kw):
        """load_gpt (src/main.zig:304-314) from a reference-format weight directory.  The matrices keep the reference's fp32
        unless every one of them is bf16-representable (weights_io.flags_for_checkpoint: bf16 storage of an ordinary fp32
        checkpoint is 6e-3 of the logit scale away from the fp32 result, outside the 1e-3 bound)."""
        from . import weights_io

        w = weights_io.load_raw_dir(path, config)
        flags = weights_io.flags_for_checkpoint(w)
        flags.update(kw)
        m = cls(config, **flags)
        m.load_weights(w)
        return m

    def close(self):
        if getattr(self, "h", None):
            self._L.zg_gpt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights):
        """weights: dict name -> fp32 array (numpy or torch), names as synth.tensor_specs."""
        L = self._L
        for s, name in enumerate(_lib.TOP_SLOTS):
            w = weights[name]
            check(L.zg_gpt_load_tensor(self.h, s, ptr(w), ops._n(w)))
        for l in range(self.config.n_layer):
            for s, name in enumerate(_lib.BLOCK_SLOTS):
                w = weights[f"h{l}.{name}"]
                check(L.zg_gpt_load_block_tensor(self.h, l, s, ptr(w), ops._n(w)))

    def weight_arena(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self._L.zg_gpt_weight_arena(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def step_bytes(self, seq_len):
        w, kv = C.c_size_t(), C.c_size_t()
        check(self._L.zg_gpt_step_bytes(self.h, seq_len, C.byref(w), C.byref(kv)))
        return w.value, kv.value

    # ------------------------------------------------------------------ GPT.forward / sample
    def forward(self, seq_len, tokens, compute_logits=True, want_logits=True):
        """GPT.forward (src/main.zig:178-195) for `batch` sequences; returns logits [batch, V] or None."""
        tokens = np.ascontiguousarray(np.atleast_1d(tokens), dtype=np.uint64)
        logits = None
        if compute_logits and want_logits:
            logits = np.empty((self.batch, self.config.vocab_size), np.float32)
        check(self._L.zg_gpt_forward(self.h, seq_len, ptr(tokens), tokens.size, int(compute_logits), ptr(logits),
                                     ops._n(logits)))
        return logits

    def prefill(self, tokens, compute_logits=True, want_logits=True):
        """The prompt loop of generate (src/main.zig:331-334) as one pass: tokens [batch, n] are positions
        0..n-1; returns the logits of position n-1 ([batch, V]) or None."""
        tokens = np.ascontiguousarray(np.atleast_2d(tokens), dtype=np.uint64)
        assert tokens.shape[0] == self.batch
        logits = None
        if compute_logits and want_logits:
            logits = np.empty((self.batch, self.config.vocab_size), np.float32)
        check(self._L.zg_gpt_prefill(self.h, ptr(tokens), tokens.shape[1], tokens.shape[1], int(compute_logits),
                                     ptr(logits), ops._n(logits)))
        return logits

    def sample(self, seq_len, tokens, temp, uniforms=None, seed=0, want_probs=False):
        """GPT.sample (src/main.zig:198-207) with reproducible uniforms; returns tokens [batch] (and probs)."""
        tokens = np.ascontiguousarray(np.atleast_1d(tokens), dtype=np.uint64)
        u = None if uniforms is None else np.ascontiguousarray(np.atleast_1d(uniforms), dtype=np.float32)
        out = np.zeros(self.batch, np.uint64)
        probs = np.empty((self.batch, self.config.vocab_size), np.float32) if want_probs else None
        check(self._L.zg_gpt_sample(self.h, seq_len, ptr(tokens), tokens.size, temp, ptr(u), seed, ptr(out), ptr(probs),
                                    ops._n(probs)))
        return (out, probs) if want_probs else out

    def argmax(self):
        out = np.zeros(self.batch, np.uint64)
        check(self._L.zg_gpt_argmax(self.h, ptr(out), out.size))
        return out

    def hidden(self):
        x = np.empty((self.batch, self.config.n_embed), np.float32)
        check(self._L.zg_gpt_hidden(self.h, ptr(x), x.size))
        return x

    # ------------------------------------------------------------------ generate
    def _prompts(self, prompts):
        prompts = [np.atleast_1d(np.asarray(p, dtype=np.uint64)) for p in prompts]
        assert len(prompts) == self.batch
        stride = max(len(p) for p in prompts)
        mat = np.zeros((self.batch, stride), np.uint64)
        lens = np.zeros(self.batch, np.uint64)
        for b, p in enumerate(prompts):
            mat[b, : len(p)] = p
            lens[b] = len(p)
        return mat, lens, stride

    def generate(self, prompts, n_steps):
        """generate (src/main.zig:322-342), greedy; returns tokens [batch, n_steps]."""
        mat, lens, stride = self._prompts(prompts)
        out = np.zeros((self.batch, n_steps), np.uint64)
        check(self._L.zg_gpt_generate_greedy(self.h, ptr(mat), stride, ptr(lens), n_steps, ptr(out), out.size))
        return out

    def generate_enqueue(self, prompts, n_steps):
        mat, lens, stride = self._prompts(prompts)
        check(self._L.zg_gpt_generate_enqueue(self.h, ptr(mat), stride, ptr(lens), n_steps))

    def generate_fetch(self, n_steps):
        out = np.zeros((self.batch, n_steps), np.uint64)
        check(self._L.zg_gpt_generate_fetch(self.h, n_steps, ptr(out), out.size))
        return out

    def generate_sample(self, prompts, n_steps, temp, seed=0):
        """generate (src/main.zig:322-342) as the reference runs it — every token behind the prompt drawn by GPT.sample — with the
        loop on the device; the tokens of the host loop over `sample(T, tok, temp, seed=seed)`."""
        mat, lens, stride = self._prompts(prompts)
        out = np.zeros((self.batch, n_steps), np.uint64)
        check(self._L.zg_gpt_generate_sample(self.h, ptr(mat), stride, ptr(lens), n_steps, temp, seed, ptr(out), out.size))
        return out

    def generate_sample_enqueue(self, prompts, n_steps, temp, seed=0):
        mat, lens, stride = self._prompts(prompts)
        check(self._L.zg_gpt_generate_sample_enqueue(self.h, ptr(mat), stride, ptr(lens), n_steps, temp, seed))

    PROFILE_CLASSES = ["embed", "ln1_c_attn_kv", "attention", "merge_attn_proj_resid", "ln2_c_fc_gelu",
                       "mlp_proj_resid", "lnf_lm_head_argmax", "step_total"]

    def profile_step(self, seq_len, iters):
        """Average microseconds per kernel class of one eager decode step (zg_gpt_profile_step)."""
        out = np.zeros(9, np.float32)
        check(self._L.zg_gpt_profile_step(self.h, seq_len, iters, out.ctypes.data_as(_lib.f32p), out.size))
        return dict(zip(self.PROFILE_CLASSES + ["null_kernel_interval"], (float(v) for v in out)))

    def prefetch_stats(self):
        """zg_debug_prefetch_stats: {"on", "workgroups", "exit", "jobs"} (lists per XCD) of the last generate call."""
        out = np.zeros(25 + 256, np.uint32)
        check(self._L.zg_debug_prefetch_stats(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return {"on": bool(out[0]), "stalled": int(out[0]) == 2, "workgroups": out[1:9].tolist(), "exit": out[9:17].tolist(), "jobs": out[17:25].tolist(),
                "xcd_of_block0": [int(v) & 15 for v in out[25:] if v & 0x100]}

    def time_kernel(self, which, iters, walk_layers=False, at=0):
        """walk_layers: launch i of the chain takes layer i mod n_layer (weights / KV from the memory side, as in the real step);
        at: the sequence length the chain runs at (0: mid-context).  ZG_TIME_WALK_LAYERS / ZG_TIME_AT of include/zgpt2.h."""
        us, nbytes = C.c_float(), C.c_size_t()
        check(self._L.zg_gpt_time_kernel(self.h, which | (0x100 if walk_layers else 0) | (int(at) << 16), iters, C.byref(us), C.byref(nbytes)))
        return us.value, nbytes.value


class GPTGroups:
    """`n_prompts` independent prompts on ONE GPU as `groups` handles of n_prompts / groups sequences each, every handle on
    its own stream, all reading one weight region (zg_gpt_create_ex) — the embarrassingly parallel case of the reference's
    generate loop (src/main.zig:322-342) with the batch == 1 restriction of src/ops.zig:126-128 lifted by running chains side
    by side instead of in lock step.  groups == 1 is a plain GPT of batch n_prompts.  Tokens equal GPT's row for row."""

    # priorities dealt to the groups' streams: streams of different priorities never share a hardware queue
    PRIORITIES = (0, 1, -1)

    def __init__(self, config: GPTConfig, n_prompts, groups, priorities=None, **kw):
        assert groups >= 1 and n_prompts % groups == 0, (n_prompts, groups)
        self.config, self.n_prompts, self.groups = config, n_prompts, groups
        self._L = _lib.load()
        per = n_prompts // groups
        # default: priorities dealt in turn up to four groups (three distinct hardware queues: 452 against 667 us per step round at
        # 4 x 2), all alike beyond that (eight streams over three priority levels serialise completely: 2.1 ms against 0.94 at 8 x 1;
        # profiles/round6_corun_ab_final.jsonl)
        if priorities is not None:
            pr = priorities
        elif groups <= 4:
            pr = [self.PRIORITIES[i % len(self.PRIORITIES)] for i in range(groups)]
        else:
            pr = [0] * groups
        self.members = []
        for i in range(groups):
            self.members.append(GPT(config, batch=per, share_weights_with=self.members[0] if i else None,
                                    own_stream=groups > 1, stream_priority=pr[i], **kw))
        self._harr = (C.c_void_p * groups)(*[m.h for m in self.members])

    @property
    def batch(self):
        return self.n_prompts

    def close(self):
        for m in reversed(getattr(self, "members", [])):  # the weight owner last
            m.close()
        self.members = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, weights):
        self.members[0].load_weights(weights)

    def _prompts(self, prompts):
        prompts = [np.atleast_1d(np.asarray(p, dtype=np.uint64)) for p in prompts]
        assert len(prompts) == self.n_prompts
        stride = max(len(p) for p in prompts)
        mat = np.zeros((self.n_prompts, stride), np.uint64)
        lens = np.zeros(self.n_prompts, np.uint64)
        for b, p in enumerate(prompts):
            mat[b, : len(p)] = p
            lens[b] = len(p)
        return mat, lens, stride

    def generate_enqueue(self, prompts, n_steps):
        mat, lens, stride = self._prompts(prompts)
        check(self._L.zg_gpt_generate_enqueue_many(self._harr, self.groups, ptr(mat), stride, ptr(lens), n_steps))

    def generate_fetch(self, n_steps):
        out = np.zeros((self.n_prompts, n_steps), np.uint64)
        check(self._L.zg_gpt_generate_fetch_many(self._harr, self.groups, n_steps, ptr(out), out.size))
        return out

    def generate(self, prompts, n_steps):
        self.generate_enqueue(prompts, n_steps)
        return self.generate_fetch(n_steps)

    def synchronize(self):
        """Drain every member's stream (hipStreamSynchronize through zg_gpt_hidden's drain would copy; use the streams)."""
        import torch

        for m in self.members:
            s = C.c_void_p()
            check(self._L.zg_gpt_stream(m.h, C.byref(s)))
            torch.cuda.ExternalStream(s.value).synchronize()


# --------------------------------------------------------------------------------------------
# Op-tier composition: src/main.zig written against ops.py the way it is written against ops.zig.
# --------------------------------------------------------------------------------------------
class State:
    """State (src/main.zig:26-65): every buffer allocated once, by the caller."""

    def __init__(self, config: GPTConfig, alloc=None):
        z = alloc or (lambda n: np.zeros(n, np.float32))
        e, c = config.n_embed, config.context_size
        self.pos_emb, self.x, self.o = z(e), z(e), z(e)
        self.logits = z(config.vocab_size)
        self._h, self._4xh, self._qkv, self._q = z(e), z(4 * e), z(3 * e), z(e)
        self._k, self._v, self._attn = z(c * e), z(c * e), z(c)


class HostGPT:
    """GPT/Block/MLP of src/main.zig:67-208 over the op tier, with host (numpy) buffers."""

    def __init__(self, config: GPTConfig, weights):
        self.config = config
        e = config.n_embed
        w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
        self.w = w
        self.wte = ops.Embedding(e, w["wte"])
        self.wpe = ops.Embedding(e, w["wpe"])
        self.ln_f = ops.LayerNorm(e, w["ln_f_g"], w["ln_f_b"])
        self.lm_head = ops.Linear(e, config.vocab_size, w["wte"], None)  # main.zig:312
        self.h = []
        for l in range(config.n_layer):
            g = lambda n: w[f"h{l}.{n}"]  # noqa: E731
            blk = dict(
                ln_1=ops.LayerNorm(e, g("ln_1_g"), g("ln_1_b")),
                attn=ops.CausalSelfAttention(config.n_heads, e, ops.Linear(e, 3 * e, g("c_attn_w"), g("c_attn_b")),
                                             ops.Linear(e, e, g("c_proj_w"), g("c_proj_b"))),
                ln_2=ops.LayerNorm(e, g("ln_2_g"), g("ln_2_b")),
                c_fc=ops.Linear(e, 4 * e, g("c_fc_w"), g("c_fc_b")),
                c_proj=ops.Linear(4 * e, e, g("mlp_proj_w"), g("mlp_proj_b")),
                k_cache=np.zeros(config.context_size * e, np.float32),  # main.zig:298-299
                v_cache=np.zeros(config.context_size * e, np.float32),
            )
            self.h.append(blk)
        self.state = State(config)

    def _block_forward(self, blk, seq_len, inputs, st):  # main.zig:119-146
        e = self.config.n_embed
        st._h[:] = inputs
        blk["ln_1"].forward(st._h)
        blk["attn"].forward(seq_len, st._h, blk["k_cache"][: seq_len * e], blk["v_cache"][: seq_len * e], st.o,
                            st._qkv, st._q, st._k[: seq_len * e], st._v[: seq_len * e], st._attn[:seq_len])
        st._h[:] = st.o + inputs
        st.x[:] = st._h
        blk["ln_2"].forward(st._h)
        blk["c_fc"].forward(st._h, st._4xh)  # MLP.forward, main.zig:78-82
        ops.gelu(st._4xh)
        blk["c_proj"].forward(st._4xh, st.o)
        st.o += st.x
        st.x[:] = st.o

    def forward(self, seq_len, token, compute_logits=True):  # main.zig:178-195
        st = self.state
        self.wpe.forward(np.array([seq_len - 1], np.uint64), st.pos_emb)
        self.wte.forward(np.array([token], np.uint64), st.x)
        st.x += st.pos_emb
        for blk in self.h:
            self._block_forward(blk, seq_len, st.x.copy(), st)
        self.ln_f.forward(st.x)
        if compute_logits:
            self.lm_head.forward(st.x, st.logits)
            return st.logits.copy()
        return None
