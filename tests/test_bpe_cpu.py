"""Tokenizer (src/bpe.zig): host code, no GPU needed.

The reference ships no tokenizer test and no vocabulary offline, so parity is stated as:
  * the hand-written word splitter == the C library's regexec on the reference's exact pattern
    (what bpe.zig executes), on seeded random ASCII / Latin-1 text and on the reference's known quirks;
  * the C-ABI Encoder (zg_bpe_*) == the Python restatement of bpe.zig:60-118 on a synthetic vocabulary
    (all 256 byte stand-ins + seeded multi-character tokens), token for token;
  * decode(encode(text)) == text whenever every byte has a single-character token.
"""
import numpy as np
import pytest

from oracle import bpe_oracle
from zig_gpt2_amd import _lib, bpe
from zig_gpt2_amd.synth import fill_uniform


def random_text(seed, n, alphabet):
    u = fill_uniform(seed, n, 0.0, 1.0)
    return bytes(alphabet[int(x * len(alphabet)) % len(alphabet)] for x in u)


ASCII = bytes(range(32, 127)) + b"\t\n" + b"   '''" + b"aeiou stn" * 3
LATIN = bytes(range(1, 256))


def make_vocab(seed, n_multi=400):
    table = bpe.unicode_to_bytes()
    chars = sorted(table, key=lambda c: table[c])  # index = byte value order is irrelevant, ids below
    vocab = {c: i for i, c in enumerate(chars)}
    by_byte = {b: c for c, b in table.items()}
    # multi-character tokens cut from seeded text, the way a real vocabulary holds word pieces (" the", "ing")
    text = random_text(seed, 6000, b"abcdefghij  e t a o n.,'0123456789")
    k = 0
    u = fill_uniform(seed + 1, n_multi, 0.0, 1.0)
    for j in range(n_multi):
        ln = 2 + int(u[j] * 4)
        piece = text[k:k + ln]
        k += ln
        tok = "".join(by_byte[b] for b in piece)
        if tok not in vocab:
            vocab[tok] = len(vocab)
    return vocab, table


def test_unicode_table_is_a_bijection_of_bytes():
    t = bpe.unicode_to_bytes()
    assert len(t) == 256 and sorted(t.values()) == list(range(256))
    assert t["!"] == 33 and t["A"] == 65 and t[chr(0xFF)] == 0xFF
    assert t[chr(256)] == 0 and t[chr(256 + 32)] == 32  # space -> U+0120, the familiar GPT-2 'Ġ'
    assert all(1 <= len(c.encode("utf-8")) <= 2 for c in t)


@pytest.mark.parametrize("alphabet,seed", [(ASCII, 1), (ASCII, 2), (LATIN, 3)])
def test_hand_splitter_equals_libc_regexec(alphabet, seed):
    import locale

    locale.setlocale(locale.LC_ALL, "C")  # the restatement states C-locale classes
    text = random_text(seed, 4000, alphabet)
    off, n_words = 0, 0
    while off < len(text):
        a = bpe_oracle.next_word(text, off)
        b = bpe_oracle.next_word_posix(text, off)
        assert a == b, (off, text[off:off + 24], a, b)
        off = a[1]
        n_words += 1
    assert n_words > 500


def test_reference_quirks():
    vocab, table = make_vocab(11)
    enc = bpe_oracle.Encoder(vocab, table)
    words = lambda t: [t[a:b] for a, b in iter_words(t)]

    def iter_words(t):
        off = 0
        while off < len(t):
            a, b = bpe_oracle.next_word(t, off)
            yield a, b
            off = b

    assert words(b"it's 42nd!") == [b"it", b"'s", b" 42", b"nd", b"!"]
    assert words(b"a  b") == [b"a", b"  ", b"b"]        # bpe.zig:38: multiple spaces are their own word
    assert words(b"x\n\ny") == [b"x", b"\n\n", b"y"]
    assert words(b"'re're''d") == [b"'re", b"'re", b"''", b"d"]
    assert enc.decode(enc.encode(b"hello  world")) == b"hello  world"


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_c_abi_encoder_matches_restatement(seed):
    vocab, table = make_vocab(seed)
    ref = bpe_oracle.Encoder(vocab, table)
    enc = bpe.Encoder(vocab, table)
    for j, alphabet in enumerate([b"abcdefghij  e t a o n.,'0123456789", ASCII, LATIN]):
        # words are capped at the reference's 20-byte buffer: keep runs short by sprinkling class changes
        text = random_text(seed * 10 + j, 3000, alphabet)
        text = b" ".join(text[i:i + 7] for i in range(0, len(text), 7))
        want = ref.encode(text)
        got = enc.encode(text)
        assert got.tolist() == want
        assert enc.decode(got) == ref.decode(want) == text
    enc.close()


def test_greedy_prefix_drops_the_rest_of_an_unknown_word():
    table = bpe.unicode_to_bytes()
    by_byte = {b: c for c, b in table.items()}
    vocab = {by_byte[ord("a")]: 0, by_byte[ord("a")] + by_byte[ord("b")]: 1}  # no token for "c"
    ref, enc = bpe_oracle.Encoder(vocab, table), bpe.Encoder(vocab, table)
    for text in (b"abab", b"abca", b"cab"):
        assert enc.encode(text).tolist() == ref.encode(text)
    assert ref.encode(b"abca") == [1]  # "abca": "ab", then no prefix of "ca" exists -> word ends (bpe.zig:81)
    enc.close()


def test_errors():
    vocab, table = make_vocab(31)
    enc = bpe.Encoder(vocab, table)
    with pytest.raises(_lib.ZgError):
        enc.encode(b"a" * 21)  # beyond the reference's 20-byte word buffer (bpe.zig:73)
    with pytest.raises(_lib.ZgError):
        enc.decode([10**9])
    assert enc.encode(b"").size == 0 and enc.decode([]) == b""
    enc.close()
