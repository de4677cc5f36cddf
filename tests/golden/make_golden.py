#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own PyTorch oracles.  Runs ONLY in the build
container (needs /root/reference); the GPU box and the tests read just the committed outputs.

Two families of fixtures:

1. ops.npz — /root/reference/generate_test_data.py is executed unmodified (runpy) in a scratch
   directory.  It is the generator behind the 8 tests of src/tests.zig.  Because the script draws
   its inputs from torch's global RNG and would otherwise force us to commit ~19 MB of random
   weight matrices, the RNG entry points it uses (torch.randn, nn.Linear/nn.Embedding parameter
   initialisation) are redirected to the portable counter PRNG of zig_gpt2_amd/synth.py while it
   runs.  Every INPUT tensor is therefore reproducible from (generator, seed, shape) — recorded in
   ops_manifest.json and re-checked here against the bytes the script wrote — and only the
   script's OUTPUT tensors (computed by PyTorch) are stored.

2. gpt_<config>.npz — the GPT definition in /root/reference/generate_nano_gpt.py:24-152 (text
   before the script tail that needs real weights + tiktoken) is exec'd with a stub `tiktoken`
   module, instantiated for small configs, loaded with synth.make_weights(...) and run with
   full-sequence causal forwards (== incremental KV-cache decode).  The fed sequence follows the
   reference decode loop src/main.zig:322-342 with greedy argmax in place of the sampler.  Stored:
   fed tokens, greedy ids and last-position logits per generation step.

Usage:  python tests/golden/make_golden.py            (writes next to this file)
"""
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from zig_gpt2_amd import synth  # noqa: E402


# --------------------------------------------------------------------------------------- ops.npz
class PatchedRNG:
    """Redirects the RNG entry points generate_test_data.py touches to synth's counter PRNG."""

    def __init__(self):
        self.records = {}  # data_ptr -> manifest entry
        self.counter = 0
        self._orig = {}

    def _next_seed(self):
        self.counter += 1
        return 10_000 + self.counter

    def _normal(self, shape, std=1.0):
        seed = self._next_seed()
        n = int(np.prod(shape))
        t = torch.from_numpy(synth.fill_normal(seed, n, 0.0, std).reshape(shape).copy())
        self.records[t.data_ptr()] = {"gen": "normal", "seed": seed, "shape": list(shape), "mean": 0.0, "std": std}
        return t

    def _uniform(self, shape, bound):
        seed = self._next_seed()
        n = int(np.prod(shape))
        t = torch.from_numpy(synth.fill_uniform(seed, n, -bound, bound).reshape(shape).copy())
        self.records[t.data_ptr()] = {"gen": "uniform", "seed": seed, "shape": list(shape), "lo": -bound, "hi": bound}
        return t

    def __enter__(self):
        rng = self
        self._orig = {
            "randn": torch.randn,
            "lin": torch.nn.Linear.reset_parameters,
            "emb": torch.nn.Embedding.reset_parameters,
        }

        def randn(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
                shape = tuple(shape[0])
            return rng._normal(tuple(shape))

        def lin_reset(mod):
            bound = float(np.float32(1.0 / np.sqrt(mod.in_features)))
            mod.weight.data = rng._uniform(tuple(mod.weight.shape), bound)
            if mod.bias is not None:
                mod.bias.data = rng._uniform(tuple(mod.bias.shape), bound)

        def emb_reset(mod):
            mod.weight.data = rng._normal(tuple(mod.weight.shape))

        torch.randn = randn
        torch.nn.Linear.reset_parameters = lin_reset
        torch.nn.Embedding.reset_parameters = emb_reset
        return self

    def __exit__(self, *a):
        torch.randn = self._orig["randn"]
        torch.nn.Linear.reset_parameters = self._orig["lin"]
        torch.nn.Embedding.reset_parameters = self._orig["emb"]


def regen(entry):
    n = int(np.prod(entry["shape"]))
    if entry["gen"] == "normal":
        a = synth.fill_normal(entry["seed"], n, entry["mean"], entry["std"])
    elif entry["gen"] == "uniform":
        a = synth.fill_uniform(entry["seed"], n, entry["lo"], entry["hi"])
    elif entry["gen"] == "const":
        a = np.full(n, entry["value"], np.float32)
    else:
        raise ValueError(entry)
    return a.reshape(entry["shape"])


def make_ops():
    torch.manual_seed(20261002)  # only torch.randint (embedding ids) still uses torch's RNG
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "models", "test"))
        os.chdir(tmp)
        try:
            with PatchedRNG() as rng:
                g = runpy.run_path(os.path.join(REF, "generate_test_data.py"))
        finally:
            os.chdir(cwd)
        files = {}
        for name in os.listdir(os.path.join(tmp, "models", "test")):
            with open(os.path.join(tmp, "models", "test", name), "rb") as f:
                files[name] = f.read()

    name_to_tensor = g["name_to_tensor"]
    manifest, stored = {}, {}
    for name, t in name_to_tensor.items():
        raw = files[name]
        entry = rng.records.get(t.data_ptr()) if t.is_contiguous() or name.endswith(("weight", "bias", "inputs")) else None
        if entry is not None and tuple(entry["shape"]) == tuple(t.shape):
            a = regen(entry)
            assert a.tobytes() == raw, f"regenerated input {name} differs from what the reference wrote"
            manifest[name] = dict(entry, role="input")
        elif name in ("layer_norm_weight", "layer_norm_bias"):
            v = 1.0 if name.endswith("weight") else 0.0  # nn.LayerNorm defaults (generate_test_data.py:68)
            entry = {"gen": "const", "value": v, "shape": list(t.shape)}
            assert regen(entry).tobytes() == raw
            manifest[name] = dict(entry, role="input")
        else:
            dt = np.int64 if t.dtype == torch.int64 else np.float32
            a = np.frombuffer(raw, dtype=dt).reshape(tuple(t.shape)).copy()
            stored[name] = a
            manifest[name] = {"gen": "stored", "shape": list(t.shape), "dtype": str(a.dtype),
                              "role": "input" if name.endswith("inputs") else "output"}
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **stored)
    with open(os.path.join(HERE, "ops_manifest.json"), "w") as f:
        json.dump({"source": "reference generate_test_data.py via tests/golden/make_golden.py",
                   "torch": torch.__version__, "tensors": manifest}, f, indent=1, sort_keys=True)
    kb = sum(a.nbytes for a in stored.values()) / 1024
    print(f"ops.npz: {len(stored)} stored tensors ({kb:.0f} KiB raw), {len(manifest) - len(stored)} regenerable inputs")


# ------------------------------------------------------------------------------------ gpt_*.npz
def load_reference_gpt_module():
    src = open(os.path.join(REF, "generate_nano_gpt.py")).read()
    head = src.split("gpt_config = GPTConfig()")[0]  # lines 1-209: definitions only
    sys.modules.setdefault("tiktoken", types.ModuleType("tiktoken"))
    ns = {"__name__": "reference_nano_gpt"}
    exec(compile(head, os.path.join(REF, "generate_nano_gpt.py"), "exec"), ns)
    return ns


def build_reference_gpt(ns, cfg, weights):
    rc = ns["GPTConfig"](vocab_size=cfg.vocab_size, block_size=cfg.context_size, n_layer=cfg.n_layer,
                         n_head=cfg.n_heads, n_embd=cfg.n_embed)
    gpt = ns["GPT"](rc).eval()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    tr = gpt.transformer
    tr.wte.weight.data = T(weights["wte"])
    tr.wpe.weight.data = T(weights["wpe"])
    tr.ln_f.weight.data, tr.ln_f.bias.data = T(weights["ln_f_g"]), T(weights["ln_f_b"])
    for l, blk in enumerate(tr.h):
        w = lambda n: T(weights[f"h{l}.{n}"])  # noqa: E731
        blk.ln_1.weight.data, blk.ln_1.bias.data = w("ln_1_g"), w("ln_1_b")
        blk.attn.c_attn.weight.data, blk.attn.c_attn.bias.data = w("c_attn_w"), w("c_attn_b")
        blk.attn.c_proj.weight.data, blk.attn.c_proj.bias.data = w("c_proj_w"), w("c_proj_b")
        blk.ln_2.weight.data, blk.ln_2.bias.data = w("ln_2_g"), w("ln_2_b")
        blk.mlp.c_fc.weight.data, blk.mlp.c_fc.bias.data = w("c_fc_w"), w("c_fc_b")
        blk.mlp.c_proj.weight.data, blk.mlp.c_proj.bias.data = w("mlp_proj_w"), w("mlp_proj_b")
    assert gpt.lm_head.weight is tr.wte.weight  # generate_nano_gpt.py:126,208
    return gpt


GPT_CASES = [
    # name, config key, weight seed, prompt seed, prompt len, n_steps, logits column stride
    ("tiny", "tiny", 1, 11, 3, 64, 1),
    ("tiny3", "tiny3", 2, 12, 1, 48, 1),
    ("nano-char", "nano-char", 3, 13, 4, 256, 1),
    ("124M", "124M", 0, 14, 2, 40, 53),
    # long prompts: pin the whole-prompt prefill pass (zg_gpt_prefill inside generate) to the reference GPT
    ("tiny-p24", "tiny", 4, 15, 24, 64, 1),
    ("tiny3-p40", "tiny3", 5, 16, 40, 48, 1),
    ("124M-p48", "124M", 0, 17, 48, 56, 53),
]


@torch.no_grad()
def make_gpt(ns, name, key, wseed, pseed, n_prompt, n_steps, stride):
    cfg = synth.CONFIGS[key]
    weights = synth.make_weights(cfg, seed=wseed, bf16=True)
    gpt = build_reference_gpt(ns, cfg, weights)
    prompt = synth.rand_tokens(pseed, n_prompt, cfg.vocab_size).astype(np.int64)
    fed, out_tokens, logits = [], [], []
    token = None
    for s in range(n_steps):  # src/main.zig:330-341, greedy
        if s < n_prompt:
            token = int(prompt[s])
            fed.append(token)
            out_tokens.append(token)
            continue
        fed.append(token)  # sample(s+1, token): previous token at position s (main.zig:337)
        lg = gpt(torch.tensor(fed, dtype=torch.long).view(1, -1))[0, -1]
        token = int(torch.argmax(lg))
        out_tokens.append(token)
        logits.append(lg.numpy().copy())
    logits = np.stack(logits)
    top = np.sort(logits, axis=1)[:, -2:]
    cols = np.arange(0, cfg.vocab_size, stride)
    np.savez_compressed(
        os.path.join(HERE, f"gpt_{name}.npz"),
        config=np.array([cfg.vocab_size, cfg.context_size, cfg.n_layer, cfg.n_heads, cfg.n_embed], np.int64),
        weight_seed=np.int64(wseed), prompt=prompt, fed=np.array(fed, np.int64),
        out_tokens=np.array(out_tokens, np.int64), logit_cols=cols.astype(np.int64),
        logits=logits[:, cols].astype(np.float32), top1=top[:, 1].astype(np.float32),
        top2=top[:, 0].astype(np.float32),
    )
    print(f"gpt_{name}.npz: {n_steps} steps, min top-2 logit gap {float((top[:, 1] - top[:, 0]).min()):.3e}")


def main():
    only = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else None  # e.g. "tiny-p24,124M-p48": just these GPT cases
    if only is None:
        make_ops()
    ns = load_reference_gpt_module()
    for case in GPT_CASES:
        if only is None or case[0] in only:
            make_gpt(ns, *case)


if __name__ == "__main__":
    main()
