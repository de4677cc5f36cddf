"""Python mirror of the reference tokenizer (src/bpe.zig) over the C ABI (zg_bpe_*).

`Encoder(token_to_idx, unicode_to_byte)` takes the two dicts the reference parses from
models/<size>/encoder.json and byte_encoder.json (src/main.zig:316-320); `unicode_to_bytes()` restates the
static table download_weights.py:69-87 dumps.  encode/decode keep bpe.zig's behaviour (no merge table).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def unicode_to_bytes():
    """unicode char -> byte: printable Latin-1 bytes map to themselves, the other 68 bytes to U+0100.."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, n = {}, 0
    for b in keep:
        table[chr(b)] = b
    for b in range(256):
        if b not in keep:
            table[chr(256 + n)] = b
            n += 1
    return table


class Encoder:
    def __init__(self, token_to_idx, unicode_to_byte):
        self._L = _lib.load()
        toks = [t.encode("utf-8") for t in token_to_idx]
        ids = np.asarray([int(token_to_idx[t]) for t in token_to_idx], np.uint64)
        unis = [u.encode("utf-8") for u in unicode_to_byte]
        byts = np.asarray([int(unicode_to_byte[u]) for u in unicode_to_byte], np.uint8)
        tok_arr = (C.c_char_p * len(toks))(*toks)
        uni_arr = (C.c_char_p * len(unis))(*unis)
        h = C.c_void_p()
        check(self._L.zg_bpe_create(C.byref(h), tok_arr, ids.ctypes.data, len(toks), uni_arr, byts.ctypes.data, len(unis)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._L.zg_bpe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, text):
        """bpe.zig:60-97; text: bytes or str (UTF-8).  Returns the token ids (numpy uint64)."""
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        out = np.zeros(max(2 * len(data), 1), np.uint64)  # every token consumes at least one byte
        n = C.c_size_t()
        check(self._L.zg_bpe_encode(self.h, data, len(data), out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value].copy()

    def decode(self, ids):
        """bpe.zig:99-118; returns bytes."""
        ids = np.ascontiguousarray(ids, np.uint64)
        cap = 64 * max(ids.size, 1)
        buf = C.create_string_buffer(cap)
        n = C.c_size_t()
        check(self._L.zg_bpe_decode(self.h, ids.ctypes.data, ids.size, buf, cap, C.byref(n)))
        return buf.raw[: n.value]
