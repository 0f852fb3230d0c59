// detok.cpp -- GPT-2 byte-level BPE DE-tokenizer: token ids -> UTF-8 text, host only.
//
// The reference stops at the language arg-max (Whisper.swift:37-39) and has no tokenizer; wm_transcribe_greedy returns
// ids, so a host that wants text needs the inverse of openai-whisper's tokenizer [3p] (GPT-2's byte-level BPE).
// Decoding needs no merges: a token's piece is a string over GPT-2's printable byte alphabet (bytes_to_unicode: the 188
// "nice" bytes map to themselves, the other 68 to U+0100 ...), the text is the concatenation of the pieces mapped back
// to bytes.  The vocabulary is the tokenizer's vocab.json ({"piece": id, ...}); neither the reference nor this image
// ships one, so the file is supplied by the host.  Ids without a piece (special tokens, timestamps: they live in
// added_tokens.json) are skipped or written as <|id|>.
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "wm_internal.h"

struct wm_vocab {
    std::vector<std::string> piece;  // id -> raw bytes of the piece (already mapped back from the byte alphabet)
    std::vector<char> has;
};

namespace {
constexpr long kMaxVocabId = 1L << 20;
// GPT-2 bytes_to_unicode(), inverted: code point -> byte
void byte_alphabet(std::unordered_map<uint32_t, unsigned char> &inv) {
    bool nice[256] = {false};
    for (int b = 33; b <= 126; ++b) nice[b] = true;
    for (int b = 161; b <= 172; ++b) nice[b] = true;
    for (int b = 174; b <= 255; ++b) nice[b] = true;
    int n = 0;
    for (int b = 0; b < 256; ++b) {
        if (nice[b]) inv[(uint32_t)b] = (unsigned char)b;
        else inv[256u + (uint32_t)n++] = (unsigned char)b;
    }
}

// minimal JSON string reader: s[i] is just past the opening quote; appends the code points
bool read_json_string(const std::string &s, size_t &i, std::vector<uint32_t> &cps) {
    auto hex4 = [&](size_t at, uint32_t &v) {
        if (at + 4 > s.size()) return false;
        v = 0;
        for (int k = 0; k < 4; ++k) {
            const char c = s[at + k];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return false;
        }
        return true;
    };
    while (i < s.size()) {
        const unsigned char c = (unsigned char)s[i];
        if (c == '"') { ++i; return true; }
        if (c == '\\') {
            if (i + 1 >= s.size()) return false;
            const char e = s[i + 1];
            i += 2;
            switch (e) {
                case '"': cps.push_back('"'); break;
                case '\\': cps.push_back('\\'); break;
                case '/': cps.push_back('/'); break;
                case 'b': cps.push_back('\b'); break;
                case 'f': cps.push_back('\f'); break;
                case 'n': cps.push_back('\n'); break;
                case 'r': cps.push_back('\r'); break;
                case 't': cps.push_back('\t'); break;
                case 'u': {
                    uint32_t v;
                    if (!hex4(i, v)) return false;
                    i += 4;
                    if (v >= 0xD800 && v <= 0xDBFF && i + 6 <= s.size() && s[i] == '\\' && s[i + 1] == 'u') {  // surrogate pair
                        uint32_t lo;
                        if (!hex4(i + 2, lo)) return false;
                        if (lo >= 0xDC00 && lo <= 0xDFFF) { v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00); i += 6; }
                    }
                    cps.push_back(v);
                    break;
                }
                default: return false;
            }
            continue;
        }
        // raw UTF-8
        uint32_t v;
        int len;
        if (c < 0x80) { v = c; len = 1; }
        else if ((c >> 5) == 6) { v = c & 0x1f; len = 2; }
        else if ((c >> 4) == 14) { v = c & 0x0f; len = 3; }
        else if ((c >> 3) == 30) { v = c & 0x07; len = 4; }
        else return false;
        if (i + len > s.size()) return false;
        for (int k = 1; k < len; ++k) {
            const unsigned char cc = (unsigned char)s[i + k];
            if ((cc & 0xC0) != 0x80) return false;   // not a continuation byte (10xxxxxx): malformed UTF-8 is rejected
            v = (v << 6) | (cc & 0x3f);
        }
        // overlong encodings and surrogates are malformed too
        if ((len == 2 && v < 0x80) || (len == 3 && v < 0x800) || (len == 4 && (v < 0x10000 || v > 0x10FFFF)) ||
            (v >= 0xD800 && v <= 0xDFFF))
            return false;
        cps.push_back(v);
        i += len;
    }
    return false;
}

void skip_ws(const std::string &s, size_t &i) {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\r' || s[i] == '\t')) ++i;
}

void put_utf8(std::string &o, uint32_t v) {
    if (v < 0x80) o += (char)v;
    else if (v < 0x800) { o += (char)(0xC0 | (v >> 6)); o += (char)(0x80 | (v & 0x3f)); }
    else if (v < 0x10000) { o += (char)(0xE0 | (v >> 12)); o += (char)(0x80 | ((v >> 6) & 0x3f)); o += (char)(0x80 | (v & 0x3f)); }
    else { o += (char)(0xF0 | (v >> 18)); o += (char)(0x80 | ((v >> 12) & 0x3f)); o += (char)(0x80 | ((v >> 6) & 0x3f)); o += (char)(0x80 | (v & 0x3f)); }
}
}  // namespace

extern "C" int wm_vocab_load(const char *vocab_json_path, wm_vocab **out) try {
    WM_REQUIRE(vocab_json_path && out, WM_ERR_INVALID, "vocab_load: null pointer");
    *out = nullptr;
    FILE *f = fopen(vocab_json_path, "rb");
    WM_REQUIRE(f, WM_ERR_IO, "cannot open '%s'", vocab_json_path);
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
    fclose(f);
    std::unordered_map<uint32_t, unsigned char> inv;
    byte_alphabet(inv);
    wm_vocab *v = new wm_vocab();
    size_t i = 0;
    bool ok = true;
    skip_ws(s, i);
    if (i >= s.size() || s[i] != '{') ok = false;
    ++i;
    skip_ws(s, i);
    if (ok && i < s.size() && s[i] == '}') { ++i; }
    else
        while (ok) {
            skip_ws(s, i);
            if (i >= s.size() || s[i] != '"') { ok = false; break; }
            ++i;
            std::vector<uint32_t> cps;
            if (!read_json_string(s, i, cps)) { ok = false; break; }
            skip_ws(s, i);
            if (i >= s.size() || s[i] != ':') { ok = false; break; }
            ++i;
            skip_ws(s, i);
            long id = 0;
            size_t digits = 0;
            while (i < s.size() && s[i] >= '0' && s[i] <= '9' && digits < 9) { id = id * 10 + (s[i] - '0'); ++i; ++digits; }
            // ids index a dense table: cap them at a realistic vocabulary size (Whisper: 51 866; GPT-class: < 2^20) so that
            // one bogus id in a malformed file cannot allocate hundreds of megabytes of empty strings
            if (digits == 0 || id >= kMaxVocabId) { ok = false; break; }
            std::string bytes;
            for (uint32_t cp : cps) {
                auto it = inv.find(cp);
                if (it != inv.end()) bytes += (char)it->second;
                else put_utf8(bytes, cp);  // outside the byte alphabet (added tokens kept verbatim)
            }
            if ((size_t)id >= v->piece.size()) { v->piece.resize(id + 1); v->has.resize(id + 1, 0); }
            v->piece[id] = bytes;
            v->has[id] = 1;
            skip_ws(s, i);
            if (i < s.size() && s[i] == ',') { ++i; continue; }
            if (i < s.size() && s[i] == '}') { ++i; break; }
            ok = false;
        }
    if (!ok) {
        delete v;
        wm_set_error("'%s': not a {\"piece\": id, ...} JSON object (stopped at byte %zu)", vocab_json_path, i);
        return WM_ERR_IO;
    }
    *out = v;
    return WM_OK;
} WM_API_CATCH

extern "C" void wm_vocab_free(wm_vocab *v) { delete v; }

extern "C" int wm_vocab_size(const wm_vocab *v) { return v ? (int)v->piece.size() : 0; }

extern "C" int wm_detokenize(const wm_vocab *v, const int32_t *ids, int n, int skip_special, char *buf, size_t cap,
                             size_t *needed) try {
    WM_REQUIRE(v && (ids || n == 0) && n >= 0 && (buf || cap == 0), WM_ERR_INVALID, "detokenize: bad arguments");
    std::string out;
    for (int i = 0; i < n; ++i) {
        const int32_t id = ids[i];
        if (id >= 0 && (size_t)id < v->piece.size() && v->has[id]) out += v->piece[id];
        else if (!skip_special) { char t[32]; snprintf(t, sizeof(t), "<|%d|>", id); out += t; }
    }
    if (needed) *needed = out.size() + 1;
    if (cap > 0) {
        const size_t m = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), m);
        buf[m] = 0;
    }
    WM_REQUIRE(cap == 0 || out.size() + 1 <= cap, WM_ERR_INVALID, "detokenize: buffer too small (%zu needed)", out.size() + 1);
    return WM_OK;
} WM_API_CATCH
