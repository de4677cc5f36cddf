"""generate (src/main.zig:322-342) as the reference runs it — every token behind the prompt drawn by GPT.sample (main.zig:198-207)
— with the loop on the device (zg_gpt_generate_sample): the sampler is a node of the captured decode step and the next step's
embed kernel feeds its draw.  Checked three ways: (1) token for token equal to the host loop over zg_gpt_sample with the same seed
(the per-token entry point the other tests hold to the oracle); (2) against the oracle's GPT.sample fed the same uniforms — the
library's counter PRNG restated here — position by position, teacher-forced on the device's draws (a differing draw is excused
only where u lands within 1e-6 of a boundary of the oracle's running sum, as tests/sweeps/sample.py does); (3) at 124M the device
loop must not be slower than the host loop it replaces."""
import time

import numpy as np
import pytest

import oracle
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu
M64 = (1 << 64) - 1


def uniform(seed, seq_len, b):
    """zg_gpt_sample's uniform for uniforms == NULL (csrc/api_gpt.hip): splitmix64 finaliser of (seed, seq_len, b), 24 random bits."""
    z = (seed * 0x9E3779B97F4A7C15 + seq_len * 0xD1B54A32D192ED03 + b + 1) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    z ^= z >> 31
    return np.float32((z >> 40) & 0xFFFFFF) * np.float32(5.9604644775390625e-08)


def host_loop(m, prompts, n_steps, temp, seed):
    """generate's token logic (prompt tokens, the last prompt token fed twice, then the draws) around the per-token sampler."""
    B = len(prompts)
    out = np.zeros((B, n_steps), np.uint64)
    draws = [0] * B
    min_np = min(len(p) for p in prompts)
    for s in range(n_steps):
        toks = [int(p[s]) if s < len(p) else (int(p[-1]) if s == len(p) else int(draws[b])) for b, p in enumerate(prompts)]
        if s >= min_np:
            draws = m.sample(s + 1, toks, temp, seed=seed)
        else:
            m.forward(s + 1, toks, compute_logits=False)
        for b, p in enumerate(prompts):
            out[b, s] = toks[b] if s < len(p) else draws[b]
    return out


@pytest.mark.parametrize("name,batch,graph", [("tiny", 1, True), ("tiny3", 3, True), ("tiny", 8, True), ("nano-char", 2, False)])
def test_device_loop_equals_host_loop_over_the_per_token_sampler(zg, name, batch, graph):
    cfg = synth.CONFIGS[name]
    w = synth.make_weights(cfg, seed=61, bf16=True)
    prompts = [synth.rand_tokens(610 + b, 1 + b % 3, cfg.vocab_size) for b in range(batch)]  # below the whole-prompt threshold: one kernel path
    n_steps = min(cfg.context_size, 70)
    for temp, seed in ((0.8, 5), (1.7, 123456789)):
        m = zgpt.GPT(cfg, batch=batch, use_graph=graph, sampled_generate=graph and batch == 1)
        m.load_weights(w)
        got = m.generate_sample(prompts, n_steps, temp, seed=seed)
        again = m.generate_sample(prompts, n_steps, temp, seed=seed)   # reproducible: same seed, same tokens
        other = m.generate_sample(prompts, n_steps, temp, seed=seed + 1)
        want = host_loop(m, prompts, n_steps, temp, seed)
        greedy = m.generate(prompts, n_steps)                           # the greedy loop still works on the same handle afterwards
        m.close()
        assert np.array_equal(got, again)
        assert np.array_equal(got, want), np.argwhere(got != want)[:4]
        assert not np.array_equal(got, other) and not np.array_equal(got, greedy)
        for b, p in enumerate(prompts):
            assert np.array_equal(got[b, : len(p)], p)


def test_sampled_generation_behind_a_whole_prompt_pass(zg):
    """A prompt long enough for the whole-prompt pass (zg_gpt_prefill inside generate): the draws behind it equal a host loop that
    prefills the same prompt and then calls zg_gpt_sample per position."""
    cfg = synth.CONFIGS["tiny3"]
    w = synth.make_weights(cfg, seed=63, bf16=True)
    prompts = [synth.rand_tokens(630 + b, 7, cfg.vocab_size) for b in range(2)]
    n_steps, temp, seed = 40, 0.8, 31
    m = zgpt.GPT(cfg, batch=2)
    m.load_weights(w)
    got = m.generate_sample(prompts, n_steps, temp, seed=seed)
    m.prefill(np.stack(prompts), compute_logits=False)
    want = np.zeros_like(got)
    want[:, :7] = np.stack(prompts)
    toks = [int(p[-1]) for p in prompts]  # main.zig:337: the last prompt token is fed again
    for s in range(7, n_steps):
        toks = [int(t) for t in m.sample(s + 1, toks, temp, seed=seed)]
        want[:, s] = toks
    m.close()
    assert np.array_equal(got, want), np.argwhere(got != want)[:4]


def test_device_loop_against_the_oracle_sampler(zg):
    cfg = synth.CONFIGS["tiny3"]
    w = synth.make_weights(cfg, seed=62, bf16=True)
    prompt = synth.rand_tokens(620, 2, cfg.vocab_size)
    n_steps, temp, seed = cfg.context_size, np.float32(0.8), 77
    m = zgpt.GPT(cfg)
    m.load_weights(w)
    got = m.generate_sample([prompt], n_steps, float(temp), seed=seed)[0]
    m.close()
    ref = oracle.GPT(cfg, w)
    near = 0
    for s in range(n_steps):  # teacher-forced on the device's own tokens: what is fed at position s (main.zig:331-338)
        tok = int(prompt[s]) if s < len(prompt) else (int(prompt[-1]) if s == len(prompt) else int(got[s - 1]))
        if s < len(prompt):
            ref.forward(s + 1, tok, False)
            assert got[s] == prompt[s]
            continue
        u = uniform(seed, s + 1, 0)
        exp_tok, exp_probs = ref.sample(s + 1, tok, temp, float(u))
        if int(got[s]) != exp_tok:
            cdf = np.cumsum(exp_probs.astype(np.float64))
            assert np.abs(cdf - float(u) * cdf[-1]).min() < 1e-6, (s, int(got[s]), exp_tok)
            near += 1
    assert near <= 2, near


def test_sampled_generation_at_124m_is_a_device_loop(zg):
    cfg = synth.CONFIGS["124M"]
    w = synth.make_weights(cfg, seed=0, bf16=True)
    prompt = [synth.rand_tokens(1000, 1, cfg.vocab_size)]
    m = zgpt.GPT(cfg)
    m.load_weights(w)
    n = 192
    m.generate_sample(prompt, n, 0.8, seed=1)  # (captures the sampled graphs)
    t0 = time.perf_counter()
    got = m.generate_sample(prompt, n, 0.8, seed=1)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = host_loop(m, prompt, n, 0.8, 1)
    t_host = time.perf_counter() - t0
    m.close()
    assert np.array_equal(got, want), np.argwhere(got != want)[:4]
    assert (got < cfg.vocab_size).all() and len(set(got[0].tolist())) > 20  # draws, not a constant
    print(f"124M sampled generation, {n} steps: device loop {n / t_dev:.0f} tok/s, host loop over zg_gpt_sample {n / t_host:.0f} tok/s")
    assert t_dev < 1.15 * t_host  # (4.6 k against 4.0 k tok/s when measured; a margin for a noisy box)
