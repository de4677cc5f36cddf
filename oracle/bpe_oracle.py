"""Test infrastructure only (see oracle/__init__.py): restatement of the reference tokenizer src/bpe.zig.

Two word splitters: `next_word_posix` calls the C library's regcomp/regexec with the reference's exact
pattern (bpe.zig:33-40) — what the Zig code itself executes — and `next_word` restates POSIX
leftmost-longest matching for that pattern by hand with C-locale classes.  Parity: unpinned by reference
fixtures (the reference has no tokenizer test or vocabulary file offline); the hand restatement is pinned
against libc on random inputs (tests/test_bpe_cpu.py).
"""
import ctypes as C
import ctypes.util

PATTERN = (b"'s|'t|'re|'ve|'m|'ll|'d"
           b"|[[:space:]]?[[:alpha:]]+"
           b"|[[:space:]]?[[:digit:]]+"
           b"|[[:space:]]?[^[:space:][:alpha:][:digit:]]+"
           b"|[[:space:]]+")  # bpe.zig:33-39
REG_EXTENDED = 1
SPACE = b" \t\n\v\f\r"


def _cls(c):
    if c in SPACE:
        return "s"
    if 65 <= c <= 90 or 97 <= c <= 122:
        return "a"
    if 48 <= c <= 57:
        return "d"
    return "o"


def match_at(p):
    """Length of the longest alternative matching at the start of p (0 = none)."""
    best = 0
    for alt in (b"'s", b"'t", b"'re", b"'ve", b"'m", b"'ll", b"'d"):
        if p.startswith(alt):
            best = max(best, len(alt))
    for k in "ado":
        for lead in (0, 1):
            if lead and not (p and _cls(p[0]) == "s"):
                continue
            i = lead
            while i < len(p) and _cls(p[i]) == k:
                i += 1
            if i > lead:
                best = max(best, i)
    i = 0
    while i < len(p) and _cls(p[i]) == "s":
        i += 1
    return max(best, i)


def next_word(text, offset):
    for so in range(offset, len(text)):
        n = match_at(text[so:])
        if n:
            return so, so + n
    return None


class _RegMatch(C.Structure):
    _fields_ = [("rm_so", C.c_int), ("rm_eo", C.c_int)]


_libc = None
_regex = None


def next_word_posix(text, offset):
    """bpe.zig:66-70: regexec on the rest of the input (NUL-terminated: inputs must not contain NUL)."""
    global _libc, _regex
    if _regex is None:
        _libc = C.CDLL(ctypes.util.find_library("c"))
        _regex = C.create_string_buffer(256)  # regex_t is 64 bytes on glibc x86-64
        assert _libc.regcomp(_regex, PATTERN, REG_EXTENDED) == 0
    m = (_RegMatch * 1)()
    if _libc.regexec(_regex, text[offset:] + b"\0", 1, m, 0) != 0:
        return None
    return offset + m[0].rm_so, offset + m[0].rm_eo


class Encoder:
    WORD_CAP = 20  # bpe.zig:73

    def __init__(self, token_to_idx, unicode_to_byte, splitter=next_word):
        self.token_to_idx = {k.encode("utf-8"): int(v) for k, v in token_to_idx.items()}
        self.idx_to_token = {v: k for k, v in self.token_to_idx.items()}
        self.unicode_to_byte = {k.encode("utf-8"): int(v) for k, v in unicode_to_byte.items()}
        self.byte_to_unicode = {v: k for k, v in self.unicode_to_byte.items()}
        self.splitter = splitter

    def encode(self, text):
        out, offset = [], 0
        while offset < len(text):
            m = self.splitter(text, offset)
            if m is None:
                break
            so, eo = m
            word = b"".join(self.byte_to_unicode[b] for b in text[so:eo])
            if len(word) > self.WORD_CAP:
                raise OverflowError("word buffer")
            t_so, t_eo = 0, len(word)
            while t_so < t_eo:  # bpe.zig:79-91
                if word[t_so:t_eo] in self.token_to_idx:
                    out.append(self.token_to_idx[word[t_so:t_eo]])
                    t_so, t_eo = t_eo, len(word)
                else:
                    t_eo -= 1
            offset = eo
        return out

    def decode(self, ids):
        out = bytearray()
        for i in ids:
            tok = self.idx_to_token[int(i)]
            k = 0
            while k < len(tok):  # bpe.zig:104-115
                if tok[k:k + 1] in self.unicode_to_byte:
                    out.append(self.unicode_to_byte[tok[k:k + 1]])
                    k += 1
                else:
                    out.append(self.unicode_to_byte[tok[k:k + 2]])
                    k += 2
        return bytes(out)
