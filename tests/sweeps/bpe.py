#!/usr/bin/env python3
"""Random byte strings (all 256 byte values, long runs, NULs, words beyond the reference's 20-byte buffer, partial vocabularies)
through the C-ABI tokenizer against the Python restatement of src/bpe.zig: same ids or the same refusal, no crash.  CPU only.
python tests/sweeps/bpe.py [first_seed] [count]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from oracle import bpe_oracle
from zig_gpt2_amd import _lib, bpe
import test_bpe_cpu as T

first, count = (int(v) for v in (sys.argv[1:3] + ["0", "300"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(4000 + seed)
    vocab, table = T.make_vocab(seed)
    if rng.integers(0, 3) == 0:  # drop part of the vocabulary: unknown words end early (bpe.zig:81)
        keys = list(vocab)
        for k in rng.choice(len(keys), len(keys) // 3, replace=False): vocab.pop(keys[k], None)
        vocab = {k: i for i, k in enumerate(vocab)}
    ref, enc = bpe_oracle.Encoder(vocab, table), bpe.Encoder(vocab, table)
    n = int(rng.integers(0, 400))
    kind = int(rng.integers(0, 4))
    if kind == 0: text = bytes(rng.integers(0, 256, n).astype(np.uint8))
    elif kind == 1: text = bytes(rng.choice(np.frombuffer(b"ab  \n\n'st9,.", np.uint8), n))
    elif kind == 2: text = bytes(rng.integers(97, 123, n).astype(np.uint8))  # long words
    else: text = bytes(rng.choice(np.array([0, 32, 39, 65, 200, 255], np.uint8), n))
    try:
        want = ref.encode(text); werr = None
    except Exception as e:
        want, werr = None, e
    try:
        got = enc.encode(text).tolist(); gerr = None
    except _lib.ZgError as e:
        got, gerr = None, e
    if (werr is None) != (gerr is None) or (werr is None and got != want):
        bad.append((seed, text[:40], werr, gerr)); continue
    if werr is None:
        try:
            a, b = ref.decode(want), enc.decode(got)
            if a != b: bad.append((seed, "decode", a[:30], b[:30]))
        except Exception as e:
            bad.append((seed, "decode raised", e))
    enc.close()
print(f"{count} texts from seed {first}: {len(bad)} mismatches {bad[:5]}")
sys.exit(1 if bad else 0)
