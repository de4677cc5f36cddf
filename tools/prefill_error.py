import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle
from golden_io import assert_model_close
from zig_gpt2_amd import _lib, gpt as zgpt, synth
_lib.check(_lib.load().zg_init(0))
for name, n in [("tiny", 48), ("nano-char", 200), ("124M", 80)]:
    cfg = synth.CONFIGS[name]
    w = synth.make_weights(cfg, seed=71, bf16=True)
    toks = synth.rand_tokens(5, n + 1, cfg.vocab_size)
    ref = oracle.GPT(cfg, w)
    lg_ref = ref.forced_logits(toks, n - 1)
    m = zgpt.GPT(cfg); m.load_weights(w)
    lg_p = m.prefill([toks[:n]]); nx_p = m.forward(n + 1, [toks[n]])
    for s in range(n - 1): m.forward(s + 1, [toks[s]], compute_logits=False)
    lg_d = m.forward(n, [toks[n - 1]]); nx_d = m.forward(n + 1, [toks[n]])
    f = lambda e, a: assert_model_close(e, a, rtol=1.0)
    print(name, n, "prefill vs oracle %.2e | decode vs oracle %.2e | next: prefill-path %.2e decode-path %.2e | prefill vs decode %.2e" % (
        f(lg_ref[0], lg_p[0]), f(lg_ref[0], lg_d[0]), f(lg_ref[1], nx_p[0]), f(lg_ref[1], nx_d[0]), f(lg_d[0], lg_p[0])))
    m.close()
