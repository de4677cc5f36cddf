// bpe.hip — host-side byte-pair tokenizer of the reference (src/bpe.zig), SURVEY §8(f)-4.  Pure host C++
// (it lives in a .hip file only so that the one-rule Makefile builds it into libzgpt2_hip.so); the generate
// loop calls it before the first and after every decode step (src/main.zig:339, :363).
//
// What bpe.zig does, restated:
//   * Encoder.init (bpe.zig:13-49): token -> id and unicode-char -> byte maps from two JSON objects, the
//     inverse maps, and ONE POSIX extended regex  's|'t|'re|'ve|'m|'ll|'d | ␣?alpha+ | ␣?digit+ |
//     ␣?other+ | space+   ([[:space:]] etc.).
//   * encode (bpe.zig:60-97): regexec on the rest of the text gives the next word (leftmost match, POSIX
//     longest alternative; anything before the match is skipped, bpe.zig:94); every byte of the word is
//     replaced by its unicode stand-in (1-2 bytes of UTF-8) into a 20-byte buffer; then greedy
//     longest-prefix lookup in the vocabulary — there is NO merge table, a known deviation from GPT-2 BPE.
//     A prefix of length 0 ends the word: the rest of it is dropped (loop condition bpe.zig:81).
//   * decode (bpe.zig:99-118): token string -> unicode chars (1 byte if that byte alone is a key, else 2)
//     -> bytes.
// Kept deviations: no merges; "multiple spaces between tokens are not handled correctly" (bpe.zig:38).
// The regex is matched by hand with C-locale classes (space = " \t\n\v\f\r", alpha = A-Za-z, digit = 0-9) so
// that the result does not depend on the process locale; tests pin it against libc regexec in the C locale.
// Words longer than the reference's 20-byte buffer are an error here (a safety panic there).
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "zg_common.h"

using namespace zg;

struct zg_bpe {
    std::unordered_map<std::string, size_t> token_to_idx;
    std::unordered_map<size_t, std::string> idx_to_token;
    std::unordered_map<std::string, unsigned char> unicode_to_byte;
    std::string byte_to_unicode[256];
    bool have_byte[256];
};

namespace {

constexpr size_t kWordCap = 20;  // bpe.zig:73

inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool is_alpha(unsigned char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool is_other(unsigned char c) { return !is_space(c) && !is_alpha(c) && !is_digit(c); }

// Longest match of the reference regex anchored at p (0 = none).  POSIX semantics: among the alternatives
// that match at the leftmost position the longest wins.
size_t match_at(const unsigned char* p, size_t n) {
    size_t best = 0;
    if (p[0] == '\'' && n >= 2) {
        static const char* const two[] = {"re", "ve", "ll"};
        if (p[1] == 's' || p[1] == 't' || p[1] == 'm' || p[1] == 'd') best = 2;
        if (n >= 3)
            for (const char* t : two)
                if (p[1] == (unsigned char)t[0] && p[2] == (unsigned char)t[1]) best = 3;
    }
    auto run = [&](size_t start, bool (*cls)(unsigned char)) {
        size_t i = start;
        while (i < n && cls(p[i])) ++i;
        return i - start;
    };
    bool (*const classes[3])(unsigned char) = {is_alpha, is_digit, is_other};
    for (auto cls : classes) {
        size_t r = run(0, cls);  // without the optional leading space
        if (r > best) best = r;
        if (is_space(p[0]) && n >= 2) {
            r = run(1, cls);
            if (r > 0 && r + 1 > best) best = r + 1;
        }
    }
    const size_t sp = run(0, is_space);
    if (sp > best) best = sp;
    return best;
}

}  // namespace

extern "C" {

int zg_bpe_create(zg_bpe** out, const char* const* tokens, const size_t* token_ids, size_t n_tokens,
                  const char* const* unicode_chars, const unsigned char* bytes, size_t n_bytes) {
    ZG_REQUIRE(out && tokens && token_ids && unicode_chars && bytes, ZG_ERR_ARG, "bpe_create: null argument");
    zg_bpe* e = new zg_bpe();
    memset(e->have_byte, 0, sizeof(e->have_byte));
    for (size_t i = 0; i < n_tokens; ++i) {
        e->token_to_idx[tokens[i]] = token_ids[i];
        e->idx_to_token[token_ids[i]] = tokens[i];
    }
    for (size_t i = 0; i < n_bytes; ++i) {
        const size_t len = strlen(unicode_chars[i]);
        if (len < 1 || len > 2) {  // decode reads keys of 1 or 2 bytes (bpe.zig:106-112)
            delete e;
            set_error("bpe_create: unicode key %zu has %zu bytes (1 or 2 expected)", i, len);
            return ZG_ERR_ARG;
        }
        e->unicode_to_byte[unicode_chars[i]] = bytes[i];
        e->byte_to_unicode[bytes[i]] = unicode_chars[i];
        e->have_byte[bytes[i]] = true;
    }
    *out = e;
    return ZG_OK;
}

int zg_bpe_destroy(zg_bpe* e) {
    delete e;
    return ZG_OK;
}

int zg_bpe_encode(zg_bpe* e, const char* text, size_t text_len, size_t* out, size_t out_cap, size_t* n_out) {
    ZG_REQUIRE(e && (text || text_len == 0) && n_out && (out || out_cap == 0), ZG_ERR_ARG, "bpe_encode: null argument");
    const unsigned char* in = reinterpret_cast<const unsigned char*>(text);
    size_t n_tok = 0, offset = 0;
    while (offset < text_len) {
        // next word: leftmost position with a match (bytes before it are skipped, bpe.zig:68-70, :94)
        size_t so = offset, len = 0;
        for (; so < text_len; ++so) {
            len = match_at(in + so, text_len - so);
            if (len) break;
        }
        if (!len) break;  // no further match
        char word[2 * kWordCap];
        size_t word_eo = 0;
        for (size_t i = so; i < so + len; ++i) {
            ZG_REQUIRE(e->have_byte[in[i]], ZG_ERR_ARG, "bpe_encode: byte 0x%02x has no unicode stand-in", in[i]);
            const std::string& u = e->byte_to_unicode[in[i]];
            ZG_REQUIRE(word_eo + u.size() <= kWordCap, ZG_ERR_SHAPE,
                       "bpe_encode: word at byte %zu exceeds the reference's %zu-byte word buffer (bpe.zig:73)", so, kWordCap);
            memcpy(word + word_eo, u.data(), u.size());
            word_eo += u.size();
        }
        size_t token_so = 0, token_eo = word_eo;
        while (token_so < token_eo) {  // greedy longest prefix, bpe.zig:79-91
            auto it = e->token_to_idx.find(std::string(word + token_so, token_eo - token_so));
            if (it != e->token_to_idx.end()) {
                ZG_REQUIRE(n_tok < out_cap, ZG_ERR_SHAPE, "bpe_encode: more than %zu tokens", out_cap);
                out[n_tok++] = it->second;
                token_so = token_eo;
                token_eo = word_eo;
            } else {
                --token_eo;
            }
        }
        offset = so + len;
    }
    *n_out = n_tok;
    return ZG_OK;
}

int zg_bpe_decode(zg_bpe* e, const size_t* ids, size_t n_ids, char* out, size_t out_cap, size_t* n_out) {
    ZG_REQUIRE(e && (ids || n_ids == 0) && n_out && (out || out_cap == 0), ZG_ERR_ARG, "bpe_decode: null argument");
    size_t n = 0;
    for (size_t t = 0; t < n_ids; ++t) {
        auto it = e->idx_to_token.find(ids[t]);
        ZG_REQUIRE(it != e->idx_to_token.end(), ZG_ERR_ARG, "bpe_decode: unknown token id %zu", ids[t]);
        const std::string& tok = it->second;
        size_t i = 0;
        while (i < tok.size()) {
            size_t w = 1;
            auto u = e->unicode_to_byte.find(tok.substr(i, 1));
            if (u == e->unicode_to_byte.end()) {
                w = 2;
                u = i + 2 <= tok.size() ? e->unicode_to_byte.find(tok.substr(i, 2)) : e->unicode_to_byte.end();
            }
            ZG_REQUIRE(u != e->unicode_to_byte.end(), ZG_ERR_ARG, "bpe_decode: token %zu has an unmapped character at byte %zu", ids[t], i);
            ZG_REQUIRE(n < out_cap, ZG_ERR_SHAPE, "bpe_decode: more than %zu bytes", out_cap);
            out[n++] = (char)u->second;
            i += w;
        }
    }
    *n_out = n;
    return ZG_OK;
}

}  // extern "C"
