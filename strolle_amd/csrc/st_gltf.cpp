// st_gltf.cpp — scene ingest: glTF 2.0 (.gltf / .glb) and PNG, on top of the public C ABI.
//
// In the reference this work is not in strolle itself: Bevy's glTF loader produces Mesh / StandardMaterial / Image
// assets and bevy-strolle's stages turn them into Engine calls (bevy-strolle/src/stages/prepare.rs:20-122 meshes,
// :124-180 materials, :182-260 images; extract.rs:200-280 instances). This file is the same step for a C caller:
// parse the file, walk the default scene's node hierarchy and call st_image_insert_rgba8 / st_material_insert /
// st_mesh_insert / st_instance_insert. It uses nothing but the public entry points, so it cannot reach around the
// boundary. Host-only, no GPU work.
//
// Conventions (the ones tools/convert_assets.py + strolle_amd/scenes.py:_insert_gltf fixed in round 1, so that both
// routes fill the engine with the same bytes):
//   * one mesh + one instance per triangle-list primitive, numbered in depth-first node order (the reference iterates a
//     HashMap there, instances.rs:80 — an order has to be fixed); handle = first_handle + index;
//   * node transforms are composed in double precision, then rounded to the f32 Affine3A the engine takes;
//   * vertices are stored object-space, de-indexed, three per triangle (prepare.rs:92-118);
//   * materials follow prepare.rs:132-175: Opaque forces alpha 1, Mask(c) turns alpha into 0/1 and becomes Blend,
//     reflectance 0.5 and ior 1 are Bevy's StandardMaterial defaults, emissive alpha 1;
//   * KHR_lights_punctual point and spot lights become lights (see Loader::light); directional ones are skipped;
//   * missing normals become flat normals (bevy_gltf computes flat normals for such meshes); missing UVs / tangents
//     are zero (prepare.rs:104-110 `unwrap_or_default`); tangents are not generated.
// Textures: PNG (decoder below) and JPEG (st_jpeg.h). Not supported (reported as ST_ERR_UNSUPPORTED, never skipped
// silently): KTX2 / WebP textures, sparse accessors, Draco / meshopt compression. Point / line / strip / fan primitives are skipped and counted, as strolle only takes triangles.
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/strolle_hip.h"
#include "st_jpeg.h"

extern "C" int st_internal_fail(int status, const char* message);  // st_engine.cpp: records st_last_error()

namespace {

struct IngestError {
    int status;
    std::string message;
};
[[noreturn]] void bad(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    for (char* c = buf; *c; c++)  // messages quote bytes of the file: keep them printable ASCII
        if ((unsigned char)*c < 0x20 || (unsigned char)*c > 0x7E) *c = '?';
    throw IngestError{status, buf};
}
#define PARSE_FAIL(...) bad(ST_ERR_PARSE, __VA_ARGS__)

using Bytes = std::vector<uint8_t>;

// ------------------------------------------------------------------------------------------------ JSON (RFC 8259)
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool boolean = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const char* key) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& at(const char* key) const {
        const Json* j = find(key);
        if (!j) PARSE_FAIL("glTF: missing property \"%s\"", key);
        return *j;
    }
    const Json& at(size_t i, const char* what) const {
        if (kind != Arr || i >= arr.size()) PARSE_FAIL("glTF: %s index %zu out of range", what, i);
        return arr[i];
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
    double number(const char* what) const {
        if (kind != Num) PARSE_FAIL("glTF: %s is not a number", what);
        return num;
    }
    size_t index(const char* what) const {
        const double v = number(what);
        if (!(v >= 0.0) || v > 4294967295.0 || v != std::floor(v)) PARSE_FAIL("glTF: %s is not a non-negative integer", what);
        return (size_t)v;
    }
    double number_or(const char* key, double fallback) const {
        const Json* j = find(key);
        return j ? j->number(key) : fallback;
    }
    size_t index_or(const char* key, size_t fallback) const {
        const Json* j = find(key);
        return j ? j->index(key) : fallback;
    }
    std::string string_or(const char* key, const char* fallback) const {
        const Json* j = find(key);
        if (!j) return fallback;
        if (j->kind != Str) PARSE_FAIL("glTF: \"%s\" is not a string", key);
        return j->str;
    }
};

class JsonParser {
  public:
    JsonParser(const char* begin, const char* end) : p_(begin), end_(end), begin_(begin) {}
    Json parse_document() {
        Json j = value(0);
        skip_ws();
        if (p_ != end_) PARSE_FAIL("JSON: trailing characters at offset %zu", (size_t)(p_ - begin_));
        return j;
    }

  private:
    const char* p_;
    const char* end_;
    const char* begin_;
    void skip_ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++;
    }
    bool eat(const char* word) {
        const size_t n = strlen(word);
        if ((size_t)(end_ - p_) >= n && memcmp(p_, word, n) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }
    static void utf8(std::string& s, uint32_t c) {
        if (c < 0x80) s += (char)c;
        else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 63)); }
        else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 63)); s += (char)(0x80 | (c & 63)); }
        else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 63)); s += (char)(0x80 | ((c >> 6) & 63)); s += (char)(0x80 | (c & 63)); }
    }
    uint32_t hex4() {
        if (end_ - p_ < 4) PARSE_FAIL("JSON: truncated \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else PARSE_FAIL("JSON: bad \\u escape");
        }
        return v;
    }
    std::string string() {
        std::string s;
        p_++;  // opening quote
        for (;;) {
            if (p_ >= end_) PARSE_FAIL("JSON: unterminated string");
            const unsigned char c = (unsigned char)*p_++;
            if (c == '"') return s;
            if (c < 0x20) PARSE_FAIL("JSON: control character inside a string");
            if (c != '\\') { s += (char)c; continue; }
            if (p_ >= end_) PARSE_FAIL("JSON: unterminated escape");
            const char e = *p_++;
            switch (e) {
                case '"': s += '"'; break;
                case '\\': s += '\\'; break;
                case '/': s += '/'; break;
                case 'b': s += '\b'; break;
                case 'f': s += '\f'; break;
                case 'n': s += '\n'; break;
                case 'r': s += '\r'; break;
                case 't': s += '\t'; break;
                case 'u': {
                    uint32_t c0 = hex4();
                    if (c0 >= 0xD800 && c0 < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        const uint32_t c1 = hex4();
                        if (c1 >= 0xDC00 && c1 < 0xE000) c0 = 0x10000 + ((c0 - 0xD800) << 10) + (c1 - 0xDC00);
                        else { utf8(s, 0xFFFD); c0 = c1; }
                    }
                    utf8(s, c0);
                    break;
                }
                default: PARSE_FAIL("JSON: unknown escape \\%c", e);
            }
        }
    }
    Json value(int depth) {
        if (depth > 128) PARSE_FAIL("JSON: nesting deeper than 128 levels");
        skip_ws();
        if (p_ >= end_) PARSE_FAIL("JSON: unexpected end of input");
        Json j;
        const char c = *p_;
        if (c == '{') {
            j.kind = Json::Obj;
            p_++;
            skip_ws();
            if (p_ < end_ && *p_ == '}') { p_++; return j; }
            for (;;) {
                skip_ws();
                if (p_ >= end_ || *p_ != '"') PARSE_FAIL("JSON: expected a property name");
                std::string key = string();
                skip_ws();
                if (p_ >= end_ || *p_ != ':') PARSE_FAIL("JSON: expected ':' after \"%s\"", key.c_str());
                p_++;
                j.obj.emplace_back(std::move(key), value(depth + 1));
                skip_ws();
                if (p_ < end_ && *p_ == ',') { p_++; continue; }
                if (p_ < end_ && *p_ == '}') { p_++; return j; }
                PARSE_FAIL("JSON: expected ',' or '}'");
            }
        }
        if (c == '[') {
            j.kind = Json::Arr;
            p_++;
            skip_ws();
            if (p_ < end_ && *p_ == ']') { p_++; return j; }
            for (;;) {
                j.arr.push_back(value(depth + 1));
                skip_ws();
                if (p_ < end_ && *p_ == ',') { p_++; continue; }
                if (p_ < end_ && *p_ == ']') { p_++; return j; }
                PARSE_FAIL("JSON: expected ',' or ']'");
            }
        }
        if (c == '"') { j.kind = Json::Str; j.str = string(); return j; }
        if (eat("true")) { j.kind = Json::Bool; j.boolean = true; return j; }
        if (eat("false")) { j.kind = Json::Bool; return j; }
        if (eat("null")) return j;
        // number: from_chars is locale-independent and correctly rounded (the Python converter's float() is too)
        j.kind = Json::Num;
        const auto r = std::from_chars(p_, end_, j.num);
        if (r.ec != std::errc() || r.ptr == p_) PARSE_FAIL("JSON: unexpected character '%c'", c);
        p_ = r.ptr;
        return j;
    }
};

// ------------------------------------------------------------------------------------------------ files, base64
Bytes read_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) bad(ST_ERR_IO, "cannot open %s", path.c_str());
    Bytes out;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    const bool err = ferror(f) != 0;
    fclose(f);
    if (err) bad(ST_ERR_IO, "read error on %s", path.c_str());
    return out;
}

Bytes base64_decode(const char* p, size_t n) {
    Bytes out;
    out.reserve(n / 4 * 3);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; i++) {
        const char c = p[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=' || c == '\n' || c == '\r') continue;
        else PARSE_FAIL("data URI: character 0x%02x is not base64", (unsigned)(unsigned char)c);
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)(acc >> bits));
        }
    }
    return out;
}

std::string percent_decode(const std::string& s) {
    std::string out;
    for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == '%' && i + 2 < s.size() && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
            out += (char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16);
            i += 2;
        } else out += s[i];
    }
    return out;
}

// uri -> bytes: data: URIs inline, anything else a path relative to the .gltf file
Bytes resolve_uri(const std::string& uri, const std::string& base_dir, const char* what) {
    if (uri.compare(0, 5, "data:") == 0) {
        const size_t comma = uri.find(',');
        if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma)
            PARSE_FAIL("glTF: %s has a data URI that is not base64", what);
        return base64_decode(uri.data() + comma + 1, uri.size() - comma - 1);
    }
    if (uri.find("://") != std::string::npos) bad(ST_ERR_UNSUPPORTED, "glTF: %s refers to a remote resource (%s)", what, uri.c_str());
    const std::string rel = percent_decode(uri);
    return read_file(base_dir.empty() || rel[0] == '/' ? rel : base_dir + "/" + rel);
}

// ------------------------------------------------------------------------------------------------ inflate (RFC 1950/1951)
class BitReader {
  public:
    BitReader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    uint32_t bits(int need) {
        while (count_ < need) {
            if (pos_ >= n_) PARSE_FAIL("deflate: stream ends inside a block");
            buf_ |= (uint32_t)p_[pos_++] << count_;
            count_ += 8;
        }
        const uint32_t v = buf_ & ((1u << need) - 1u);
        buf_ >>= need;
        count_ -= need;
        return v;
    }
    void align() { buf_ = 0; count_ = 0; }
    const uint8_t* take(size_t n) {
        if (n > n_ - pos_) PARSE_FAIL("deflate: stored block runs past the end");
        const uint8_t* q = p_ + pos_;
        pos_ += n;
        return q;
    }
    size_t position() const { return pos_; }

  private:
    const uint8_t* p_;
    size_t n_, pos_ = 0;
    uint32_t buf_ = 0;
    int count_ = 0;
};

struct Huffman {
    uint16_t count[16];
    uint16_t symbol[288];
    // canonical code from code lengths; returns false when the set is over-subscribed
    bool build(const uint8_t* lengths, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        int left = 1;
        for (int len = 1; len < 16; len++) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= (int)br.bits(1);
            const int cnt = count[len];
            if (code - cnt < first) return symbol[index + (code - first)];
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        PARSE_FAIL("deflate: invalid Huffman code");
    }
};

void inflate_codes(BitReader& br, Bytes& out, const Huffman& lit, const Huffman& dist, size_t limit) {
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        const int sym = lit.decode(br);
        if (sym < 256) {
            if (out.size() >= limit) PARSE_FAIL("deflate: output larger than the %zu bytes expected", limit);
            out.push_back((uint8_t)sym);
        } else if (sym == 256) {
            return;
        } else {
            if (sym > 285) PARSE_FAIL("deflate: invalid length symbol");
            const size_t len = len_base[sym - 257] + br.bits(len_extra[sym - 257]);
            const int ds = dist.decode(br);
            if (ds > 29) PARSE_FAIL("deflate: invalid distance symbol");
            const size_t d = dist_base[ds] + br.bits(dist_extra[ds]);
            if (d > out.size()) PARSE_FAIL("deflate: distance reaches before the start of the output");
            if (out.size() + len > limit) PARSE_FAIL("deflate: output larger than the %zu bytes expected", limit);
            for (size_t i = 0; i < len; i++) out.push_back(out[out.size() - d]);
        }
    }
}

// zlib stream -> bytes; `limit` bounds the output (PNG knows its exact size up front)
Bytes zlib_inflate(const uint8_t* p, size_t n, size_t limit) {
    if (n < 6) PARSE_FAIL("zlib: stream too short");
    if ((p[0] & 15) != 8 || ((p[0] << 8) | p[1]) % 31 != 0 || (p[1] & 0x20)) PARSE_FAIL("zlib: bad header");
    BitReader br(p + 2, n - 2);
    Bytes out;
    out.reserve(limit < (64u << 20) ? limit : (64u << 20));
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (type == 0) {
            br.align();
            const uint8_t* h = br.take(4);
            const uint32_t len = h[0] | (h[1] << 8), nlen = h[2] | (h[3] << 8);
            if ((len ^ 0xFFFFu) != nlen) PARSE_FAIL("deflate: stored block length check failed");
            const uint8_t* q = br.take(len);
            if (out.size() + len > limit) PARSE_FAIL("deflate: output larger than the %zu bytes expected", limit);
            out.insert(out.end(), q, q + len);
        } else if (type == 1) {
            uint8_t l[288];
            for (int i = 0; i < 144; i++) l[i] = 8;
            for (int i = 144; i < 256; i++) l[i] = 9;
            for (int i = 256; i < 280; i++) l[i] = 7;
            for (int i = 280; i < 288; i++) l[i] = 8;
            Huffman lit, dist;
            lit.build(l, 288);
            uint8_t d[30];
            memset(d, 5, sizeof d);
            dist.build(d, 30);
            inflate_codes(br, out, lit, dist, limit);
        } else if (type == 2) {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            if (nlen > 286 || ndist > 30) PARSE_FAIL("deflate: too many codes");
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t l[320];
            memset(l, 0, sizeof l);
            for (int i = 0; i < ncode; i++) l[order[i]] = (uint8_t)br.bits(3);
            Huffman cl;
            if (!cl.build(l, 19)) PARSE_FAIL("deflate: bad code-length code");
            uint8_t lengths[320];
            int i = 0;
            while (i < nlen + ndist) {
                const int sym = cl.decode(br);
                if (sym < 16) { lengths[i++] = (uint8_t)sym; continue; }
                uint8_t prev = 0;
                int rep;
                if (sym == 16) {
                    if (i == 0) PARSE_FAIL("deflate: repeat with nothing to repeat");
                    prev = lengths[i - 1];
                    rep = 3 + (int)br.bits(2);
                } else if (sym == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (i + rep > nlen + ndist) PARSE_FAIL("deflate: code lengths overflow");
                while (rep--) lengths[i++] = prev;
            }
            if (lengths[256] == 0) PARSE_FAIL("deflate: no end-of-block code");
            Huffman lit, dist;
            if (!lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) PARSE_FAIL("deflate: over-subscribed code");
            inflate_codes(br, out, lit, dist, limit);
        } else PARSE_FAIL("deflate: reserved block type");
        if (last) break;
    }
    br.align();
    const uint8_t* a = br.take(4);
    uint32_t s1 = 1, s2 = 0;
    for (uint8_t b : out) { s1 = (s1 + b) % 65521u; s2 = (s2 + s1) % 65521u; }
    if ((((uint32_t)a[0] << 24) | (a[1] << 16) | (a[2] << 8) | a[3]) != ((s2 << 16) | s1)) PARSE_FAIL("zlib: Adler-32 mismatch");
    return out;
}

// ------------------------------------------------------------------------------------------------ PNG (ISO/IEC 15948)
struct Picture {
    uint32_t width = 0, height = 0;
    Bytes rgba;  // width*height*4, straight alpha, 8 bits per channel (16-bit files keep their high byte)
};

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

uint32_t crc32(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

constexpr uint32_t kMaxImageSide = 8192;  // the atlas is 8192 texels wide (images.rs:17-20); nothing larger can be used

Picture decode_png(const uint8_t* p, size_t n) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(p, sig, 8) != 0) PARSE_FAIL("PNG: bad signature");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    Bytes idat, plte, trns;
    bool have_trns = false, end = false;
    for (size_t off = 8; !end;) {
        if (n - off < 12) PARSE_FAIL("PNG: truncated chunk header");
        const uint32_t len = be32(p + off);
        if ((size_t)len > n - off - 12) PARSE_FAIL("PNG: chunk runs past the end of the file");
        const uint8_t* type = p + off + 4;
        const uint8_t* data = p + off + 8;
        if (crc32(type, 4 + (size_t)len) != be32(data + len)) PARSE_FAIL("PNG: CRC mismatch in %.4s", (const char*)type);
        if (memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) PARSE_FAIL("PNG: IHDR has %u bytes", len);
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || interlace > 1) PARSE_FAIL("PNG: unknown compression / filter / interlace method");
        } else if (memcmp(type, "PLTE", 4) == 0) plte.assign(data, data + len);
        else if (memcmp(type, "tRNS", 4) == 0) { trns.assign(data, data + len); have_trns = true; }
        else if (memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), data, data + len);
        else if (memcmp(type, "IEND", 4) == 0) end = true;
        else if (!(type[0] & 0x20)) PARSE_FAIL("PNG: unknown critical chunk %.4s", (const char*)type);
        off += 12 + (size_t)len;
    }
    if (ctype < 0) PARSE_FAIL("PNG: no IHDR");
    if (w == 0 || h == 0 || w > kMaxImageSide || h > kMaxImageSide) bad(ST_ERR_UNSUPPORTED, "PNG: %ux%u is outside 1..%u", w, h, kMaxImageSide);
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: PARSE_FAIL("PNG: colour type %d", ctype);
    }
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok) PARSE_FAIL("PNG: bit depth %d with colour type %d", depth, ctype);
    if (ctype == 3 && (plte.empty() || plte.size() % 3)) PARSE_FAIL("PNG: palette image without a valid PLTE");

    const int bits_pp = channels * depth;
    const size_t bpp = (size_t)(bits_pp + 7) / 8;  // filter unit
    auto row_bytes = [&](uint32_t pw) { return ((size_t)pw * bits_pp + 7) / 8; };
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass* passes = interlace ? adam7 : &whole;
    const int n_pass = interlace ? 7 : 1;
    size_t expect = 0;
    for (int i = 0; i < n_pass; i++) {
        const uint32_t pw = w > passes[i].x0 ? (w - passes[i].x0 + passes[i].dx - 1) / passes[i].dx : 0;
        const uint32_t ph = h > passes[i].y0 ? (h - passes[i].y0 + passes[i].dy - 1) / passes[i].dy : 0;
        if (pw && ph) expect += (size_t)ph * (1 + row_bytes(pw));
    }
    Bytes raw = zlib_inflate(idat.data(), idat.size(), expect);
    if (raw.size() != expect) PARSE_FAIL("PNG: image data has %zu bytes, %zu expected", raw.size(), expect);

    Picture pic;
    pic.width = w; pic.height = h;
    pic.rgba.assign((size_t)w * h * 4, 0);
    // a colour key (tRNS on grey / RGB images) compares full-depth samples
    uint32_t key[3] = {0, 0, 0};
    const bool keyed = have_trns && (ctype == 0 || ctype == 2) && trns.size() >= (size_t)(ctype == 0 ? 2 : 6);
    if (keyed)
        for (int c = 0; c < (ctype == 0 ? 1 : 3); c++) key[c] = ((uint32_t)trns[2 * c] << 8) | trns[2 * c + 1];

    size_t at = 0;
    Bytes prev_row, row;
    for (int i = 0; i < n_pass; i++) {
        const Pass& ps = passes[i];
        const uint32_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0;
        const uint32_t ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        if (!pw || !ph) continue;
        const size_t rb = row_bytes(pw);
        prev_row.assign(rb, 0);
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t filter = raw[at++];
            row.assign(raw.begin() + (ptrdiff_t)at, raw.begin() + (ptrdiff_t)(at + rb));
            at += rb;
            for (size_t k = 0; k < rb; k++) {
                const int a = k >= bpp ? row[k - bpp] : 0, b = prev_row[k], c = k >= bpp ? prev_row[k - bpp] : 0;
                int pred;
                switch (filter) {
                    case 0: pred = 0; break;
                    case 1: pred = a; break;
                    case 2: pred = b; break;
                    case 3: pred = (a + b) >> 1; break;
                    case 4: {
                        const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
                        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                        break;
                    }
                    default: PARSE_FAIL("PNG: filter type %d", filter);
                }
                row[k] = (uint8_t)(row[k] + pred);
            }
            // full-depth sample c of pixel x in this row
            auto sample = [&](uint32_t x, int c) -> uint32_t {
                const size_t s = (size_t)x * channels + c;
                if (depth == 8) return row[s];
                if (depth == 16) return ((uint32_t)row[2 * s] << 8) | row[2 * s + 1];
                const size_t bit = s * depth;
                return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
            };
            auto to8 = [&](uint32_t v) -> uint8_t {
                if (depth == 8) return (uint8_t)v;
                if (depth == 16) return (uint8_t)(v >> 8);
                return (uint8_t)(v * 255u / ((1u << depth) - 1u));
            };
            for (uint32_t x = 0; x < pw; x++) {
                uint8_t* o = &pic.rgba[(((size_t)ps.y0 + (size_t)y * ps.dy) * w + ps.x0 + (size_t)x * ps.dx) * 4];
                if (ctype == 3) {
                    const uint32_t idx = sample(x, 0);
                    if ((size_t)idx * 3 + 2 >= plte.size()) PARSE_FAIL("PNG: palette index %u out of range", idx);
                    o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2];
                    o[3] = have_trns && idx < trns.size() ? trns[idx] : 255;
                } else if (ctype == 0 || ctype == 4) {
                    const uint32_t g = sample(x, 0);
                    o[0] = o[1] = o[2] = to8(g);
                    o[3] = ctype == 4 ? to8(sample(x, 1)) : (keyed && g == key[0] ? 0 : 255);
                } else {
                    const uint32_t r = sample(x, 0), g = sample(x, 1), b = sample(x, 2);
                    o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
                    o[3] = ctype == 6 ? to8(sample(x, 3)) : (keyed && r == key[0] && g == key[1] && b == key[2] ? 0 : 255);
                }
            }
            prev_row.swap(row);
        }
    }
    return pic;
}

// PNG or JPEG, by signature (glTF 2.0 allows exactly these two, section 3.8.3)
Picture decode_image(const uint8_t* p, size_t n, const char* what) {
    if (n >= 3 && p[0] == 0xFF && p[1] == 0xD8) {
        try {
            st_jpeg::Image j = st_jpeg::Decoder(p, n, kMaxImageSide).decode();
            Picture pic;
            pic.width = j.width; pic.height = j.height; pic.rgba.swap(j.rgba);
            return pic;
        } catch (const st_jpeg::Error& e) {
            bad(e.unsupported ? ST_ERR_UNSUPPORTED : ST_ERR_PARSE, "%s: %s", what, e.message);
        }
    }
    if (n >= 8 && p[0] == 0x89 && p[1] == 'P') return decode_png(p, n);
    bad(ST_ERR_UNSUPPORTED, "%s: neither a PNG nor a JPEG (KTX2 / WebP / DDS textures are not read)", what);
}

// ------------------------------------------------------------------------------------------------ glTF
struct Mat4d {
    double m[4][4];  // row-major, m[row][col]
    static Mat4d identity() {
        Mat4d r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0 : 0.0;
        return r;
    }
};
Mat4d mul(const Mat4d& a, const Mat4d& b) {
    Mat4d r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; k++) s = s + a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}

struct Document {
    Json root;
    std::vector<Bytes> buffers;
    std::string base_dir;
};

Mat4d node_matrix(const Json& node) {
    Mat4d r = Mat4d::identity();
    if (const Json* mj = node.find("matrix")) {
        if (mj->size() != 16) PARSE_FAIL("glTF: node matrix needs 16 numbers");
        for (int c = 0; c < 4; c++)
            for (int rr = 0; rr < 4; rr++) r.m[rr][c] = mj->arr[(size_t)c * 4 + rr].number("matrix element");  // column-major
        return r;
    }
    double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    auto read = [&](const char* key, double* out, size_t n) {
        if (const Json* j = node.find(key)) {
            if (j->size() != n) PARSE_FAIL("glTF: node %s needs %zu numbers", key, n);
            for (size_t i = 0; i < n; i++) out[i] = j->arr[i].number(key);
        }
    };
    read("translation", t, 3);
    read("rotation", q, 4);
    read("scale", s, 3);
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double rot[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                              {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                              {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) r.m[i][j] = rot[i][j] * s[j];
        r.m[i][3] = t[i];
    }
    return r;
}

struct AccessorView {
    const uint8_t* base = nullptr;  // nullptr: the accessor has no bufferView and reads as zeros
    size_t count = 0, stride = 0;
    int component = 0, ncomp = 0;
    bool normalized = false;
};

int component_size(int component) {
    switch (component) {
        case 5120: case 5121: return 1;
        case 5122: case 5123: return 2;
        case 5125: case 5126: return 4;
        default: PARSE_FAIL("glTF: accessor componentType %d", component);
    }
}

AccessorView accessor(const Document& doc, size_t index) {
    const Json& acc = doc.root.at("accessors").at(index, "accessor");
    if (acc.find("sparse")) bad(ST_ERR_UNSUPPORTED, "glTF: sparse accessors are not supported");
    AccessorView v;
    v.count = acc.at("count").index("accessor.count");
    v.component = (int)acc.at("componentType").index("accessor.componentType");
    const std::string type = acc.string_or("type", "");
    v.ncomp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
    if (!v.ncomp) PARSE_FAIL("glTF: accessor type \"%s\"", type.c_str());
    if (const Json* nj = acc.find("normalized")) v.normalized = nj->kind == Json::Bool && nj->boolean;
    const size_t elem = (size_t)component_size(v.component) * v.ncomp;
    const Json* bvj = acc.find("bufferView");
    if (!bvj) {  // reads as zeros (spec 3.6.2.2); nothing in the file bounds its size, so bound it here
        if (v.count > (1u << 24)) bad(ST_ERR_UNSUPPORTED, "glTF: accessor %zu has no bufferView and %zu elements", index, v.count);
        return v;
    }
    const Json& bv = doc.root.at("bufferViews").at(bvj->index("accessor.bufferView"), "bufferView");
    const size_t buffer = bv.at("buffer").index("bufferView.buffer");
    if (buffer >= doc.buffers.size()) PARSE_FAIL("glTF: bufferView refers to buffer %zu", buffer);
    const Bytes& data = doc.buffers[buffer];
    const size_t view_off = bv.index_or("byteOffset", 0), view_len = bv.at("byteLength").index("bufferView.byteLength");
    if (view_off > data.size() || view_len > data.size() - view_off) PARSE_FAIL("glTF: bufferView runs past the end of buffer %zu", buffer);
    const size_t acc_off = acc.index_or("byteOffset", 0);
    v.stride = bv.index_or("byteStride", 0);
    if (v.stride == 0) v.stride = elem;
    if (v.count && (acc_off > view_len || (v.count - 1) * v.stride + elem > view_len - acc_off)) PARSE_FAIL("glTF: accessor %zu runs past its bufferView", index);
    v.base = data.data() + view_off + acc_off;
    return v;
}

// element i, component c as float: float as is, normalized integers per the glTF spec (3.6.2.3), others cast
float accessor_float(const AccessorView& v, size_t i, int c) {
    if (!v.base) return 0.0f;
    const uint8_t* p = v.base + i * v.stride + (size_t)c * component_size(v.component);
    switch (v.component) {
        case 5126: { float f; memcpy(&f, p, 4); return f; }
        case 5121: return v.normalized ? (float)*p / 255.0f : (float)*p;
        case 5123: { uint16_t u; memcpy(&u, p, 2); return v.normalized ? (float)u / 65535.0f : (float)u; }
        case 5120: { const int8_t s = (int8_t)*p; return v.normalized ? fmaxf((float)s / 127.0f, -1.0f) : (float)s; }
        case 5122: { int16_t s; memcpy(&s, p, 2); return v.normalized ? fmaxf((float)s / 32767.0f, -1.0f) : (float)s; }
        default: { uint32_t u; memcpy(&u, p, 4); return (float)u; }
    }
}
uint32_t accessor_index(const AccessorView& v, size_t i) {
    if (!v.base) return 0;
    const uint8_t* p = v.base + i * v.stride;
    switch (v.component) {
        case 5121: return *p;
        case 5123: { uint16_t u; memcpy(&u, p, 2); return u; }
        case 5125: { uint32_t u; memcpy(&u, p, 4); return u; }
        default: PARSE_FAIL("glTF: index accessor with componentType %d", v.component);
    }
}

Document open_document(const uint8_t* bytes, size_t size, const std::string& base_dir) {
    Document doc;
    doc.base_dir = base_dir;
    Bytes glb_bin;
    bool have_bin = false;
    if (size >= 12 && memcmp(bytes, "glTF", 4) == 0) {
        uint32_t version, length;
        memcpy(&version, bytes + 4, 4);
        memcpy(&length, bytes + 8, 4);
        if (version != 2) bad(ST_ERR_UNSUPPORTED, "GLB: container version %u", version);
        if (length > size) PARSE_FAIL("GLB: header says %u bytes, file has %zu", length, size);
        size_t off = 12;
        bool have_json = false;
        while (off + 8 <= length) {
            uint32_t clen, ctype;
            memcpy(&clen, bytes + off, 4);
            memcpy(&ctype, bytes + off + 4, 4);
            if ((size_t)clen > length - off - 8) PARSE_FAIL("GLB: chunk runs past the end");
            const uint8_t* c = bytes + off + 8;
            if (ctype == 0x4E4F534Au && !have_json) {
                doc.root = JsonParser((const char*)c, (const char*)c + clen).parse_document();
                have_json = true;
            } else if (ctype == 0x004E4942u && !have_bin) {
                glb_bin.assign(c, c + clen);
                have_bin = true;
            }
            off += 8 + (size_t)clen;
        }
        if (!have_json) PARSE_FAIL("GLB: no JSON chunk");
    } else {
        doc.root = JsonParser((const char*)bytes, (const char*)bytes + size).parse_document();
    }
    if (doc.root.kind != Json::Obj) PARSE_FAIL("glTF: the document is not a JSON object");
    if (const Json* asset = doc.root.find("asset")) {
        const std::string v = asset->string_or("version", "2.0");
        if (v.compare(0, 2, "2.") != 0) bad(ST_ERR_UNSUPPORTED, "glTF: asset version %s", v.c_str());
    }
    if (const Json* req = doc.root.find("extensionsRequired"))
        for (auto& e : req->arr)
            if (e.kind == Json::Str && e.str != "KHR_materials_unlit" && e.str != "KHR_texture_transform" && e.str != "KHR_materials_emissive_strength")
                bad(ST_ERR_UNSUPPORTED, "glTF: required extension %s is not supported", e.str.c_str());
    if (const Json* bufs = doc.root.find("buffers")) {
        for (size_t i = 0; i < bufs->size(); i++) {
            const Json& b = bufs->arr[i];
            Bytes data;
            if (const Json* uri = b.find("uri")) {
                if (uri->kind != Json::Str) PARSE_FAIL("glTF: buffer uri is not a string");
                data = resolve_uri(uri->str, base_dir, "a buffer");
            } else {
                if (i != 0 || !have_bin) PARSE_FAIL("glTF: buffer %zu has no uri and there is no GLB binary chunk", i);
                data = std::move(glb_bin);
            }
            const size_t need = b.at("byteLength").index("buffer.byteLength");
            if (data.size() < need) PARSE_FAIL("glTF: buffer %zu holds %zu bytes, %zu declared", i, data.size(), need);
            doc.buffers.push_back(std::move(data));
        }
    }
    return doc;
}

Bytes image_bytes(const Document& doc, const Json& image, size_t index) {
    if (const Json* bvj = image.find("bufferView")) {
        const Json& bv = doc.root.at("bufferViews").at(bvj->index("image.bufferView"), "bufferView");
        const size_t buffer = bv.at("buffer").index("bufferView.buffer");
        if (buffer >= doc.buffers.size()) PARSE_FAIL("glTF: image %zu refers to buffer %zu", index, buffer);
        const Bytes& data = doc.buffers[buffer];
        const size_t off = bv.index_or("byteOffset", 0), len = bv.at("byteLength").index("bufferView.byteLength");
        if (off > data.size() || len > data.size() - off) PARSE_FAIL("glTF: image %zu runs past its buffer", index);
        return Bytes(data.begin() + (ptrdiff_t)off, data.begin() + (ptrdiff_t)(off + len));
    }
    const Json* uri = image.find("uri");
    if (!uri || uri->kind != Json::Str) PARSE_FAIL("glTF: image %zu has neither a bufferView nor a uri", index);
    return resolve_uri(uri->str, doc.base_dir, "an image");
}

// midpoint subdivision, 4 triangles per level, in the order scenes.py:_subdivide emits them (all corner-0 triangles, then
// corner-1, corner-2, then the centre ones): the surface, normals and texture mapping stay, only the triangle count grows
void subdivide(std::vector<StMeshTriangle>& tris, uint32_t levels) {
    for (uint32_t l = 0; l < levels; l++) {
        const size_t n = tris.size();
        std::vector<StMeshTriangle> out(n * 4);
        for (size_t i = 0; i < n; i++) {
            const StMeshTriangle& t = tris[i];
            StMeshTriangle m;  // midpoints 01, 12, 20 stored at [0], [1], [2]
            auto mid = [](const float* a, const float* b, float* o, int k) {
                for (int c = 0; c < k; c++) o[c] = (a[c] + b[c]) * 0.5f;
            };
            for (int e = 0; e < 3; e++) {
                const int a = e, b = (e + 1) % 3;
                mid(t.positions[a], t.positions[b], m.positions[e], 3);
                mid(t.normals[a], t.normals[b], m.normals[e], 3);
                mid(t.uvs[a], t.uvs[b], m.uvs[e], 2);
                mid(t.tangents[a], t.tangents[b], m.tangents[e], 4);
            }
            auto put = [](StMeshTriangle& d, int slot, const StMeshTriangle& s, int from) {
                memcpy(d.positions[slot], s.positions[from], sizeof d.positions[slot]);
                memcpy(d.normals[slot], s.normals[from], sizeof d.normals[slot]);
                memcpy(d.uvs[slot], s.uvs[from], sizeof d.uvs[slot]);
                memcpy(d.tangents[slot], s.tangents[from], sizeof d.tangents[slot]);
            };
            StMeshTriangle& a = out[i];          // v0, m01, m20
            put(a, 0, t, 0); put(a, 1, m, 0); put(a, 2, m, 2);
            StMeshTriangle& b = out[n + i];      // m01, v1, m12
            put(b, 0, m, 0); put(b, 1, t, 1); put(b, 2, m, 1);
            StMeshTriangle& c = out[2 * n + i];  // m20, m12, v2
            put(c, 0, m, 2); put(c, 1, m, 1); put(c, 2, t, 2);
            StMeshTriangle& d = out[3 * n + i];  // m01, m12, m20
            put(d, 0, m, 0); put(d, 1, m, 1); put(d, 2, m, 2);
        }
        tris.swap(out);
    }
}

struct Loader {
    StEngine* engine;
    const Document& doc;
    StGltfOptions opt;
    StGltfSummary sum{};
    size_t default_material = (size_t)-1;  // index of the material made for primitives that name none

    void check(int status, const char* what) {
        if (status != ST_OK) bad(status, "%s: %s", what, st_last_error());
    }

    void images() {
        const Json* imgs = doc.root.find("images");
        if (!imgs) return;
        // linear only when every use of the image is a data texture (normal / metallic-roughness / occlusion)
        std::vector<int> colour_use(imgs->size(), 0), data_use(imgs->size(), 0);
        auto texture_source = [&](const Json* info) -> size_t {
            if (!info) return (size_t)-1;
            const Json& tex = doc.root.at("textures").at(info->at("index").index("texture index"), "texture");
            const Json* src = tex.find("source");
            return src ? src->index("texture.source") : (size_t)-1;
        };
        if (const Json* mats = doc.root.find("materials"))
            for (auto& m : mats->arr) {
                const Json* pbr = m.find("pbrMetallicRoughness");
                const size_t uses[5] = {texture_source(pbr ? pbr->find("baseColorTexture") : nullptr), texture_source(m.find("emissiveTexture")),
                                        texture_source(pbr ? pbr->find("metallicRoughnessTexture") : nullptr), texture_source(m.find("normalTexture")),
                                        texture_source(m.find("occlusionTexture"))};
                for (int k = 0; k < 5; k++)
                    if (uses[k] < imgs->size()) (k < 2 ? colour_use : data_use)[uses[k]]++;
            }
        for (size_t i = 0; i < imgs->size(); i++) {
            const Bytes raw = image_bytes(doc, imgs->arr[i], i);
            char what[32];
            snprintf(what, sizeof what, "glTF image %zu", i);
            const Picture pic = decode_image(raw.data(), raw.size(), what);
            const int srgb = (data_use[i] && !colour_use[i]) ? 0 : 1;
            const int rc = st_image_insert_rgba8(engine, opt.first_image_handle + i, pic.width, pic.height, pic.rgba.data(), srgb);
            if (rc == ST_ERR_ATLAS_FULL) { sum.images_dropped++; continue; }  // the reference warns and drops (images.rs:71-79)
            check(rc, "st_image_insert_rgba8");
            sum.images++;
        }
    }

    StHandle texture_handle(const Json* info) {
        if (!info) return 0;
        const Json& tex = doc.root.at("textures").at(info->at("index").index("texture index"), "texture");
        const Json* src = tex.find("source");
        return src ? opt.first_image_handle + src->index("texture.source") : 0;
    }

    void insert_material(size_t index, const Json* m) {
        static const Json empty;
        const Json* pbr = m ? m->find("pbrMetallicRoughness") : nullptr;
        if (!pbr) pbr = &empty;
        StMaterial out;
        memset(&out, 0, sizeof out);
        double base[4] = {1, 1, 1, 1}, emissive[3] = {0, 0, 0};
        if (const Json* f = pbr->find("baseColorFactor")) {
            if (f->size() != 4) PARSE_FAIL("glTF: baseColorFactor needs 4 numbers");
            for (int i = 0; i < 4; i++) base[i] = f->arr[(size_t)i].number("baseColorFactor");
        }
        if (const Json* f = m ? m->find("emissiveFactor") : nullptr) {
            if (f->size() != 3) PARSE_FAIL("glTF: emissiveFactor needs 3 numbers");
            for (int i = 0; i < 3; i++) emissive[i] = f->arr[(size_t)i].number("emissiveFactor");
        }
        const std::string mode = m ? m->string_or("alphaMode", "OPAQUE") : "OPAQUE";
        // prepare.rs:132-154
        if (mode == "OPAQUE") base[3] = 1.0;
        else if (mode == "MASK") base[3] = (float)base[3] >= (float)(m->number_or("alphaCutoff", 0.5)) ? 1.0 : 0.0;
        else if (mode != "BLEND") PARSE_FAIL("glTF: alphaMode \"%s\"", mode.c_str());
        out.alpha_mode = mode == "OPAQUE" ? 0u : 1u;
        for (int i = 0; i < 4; i++) out.base_color[i] = (float)base[i];
        for (int i = 0; i < 3; i++) out.emissive[i] = (float)emissive[i];
        out.emissive[3] = 1.0f;
        out.metallic = (float)pbr->number_or("metallicFactor", 1.0);
        out.perceptual_roughness = (float)pbr->number_or("roughnessFactor", 1.0);  // bevy_gltf: perceptual_roughness = roughness_factor
        out.reflectance = 0.5f;                                                     // StandardMaterial::default()
        out.ior = 1.0f;                                                             // prepare.rs:149 (thickness == 0)
        if (opt.override_mask & ST_GLTF_OVERRIDE_REFLECTANCE) out.reflectance = opt.reflectance;
        if (opt.override_mask & ST_GLTF_OVERRIDE_PERCEPTUAL_ROUGHNESS) out.perceptual_roughness = opt.perceptual_roughness;
        out.base_color_texture = texture_handle(pbr->find("baseColorTexture"));
        out.emissive_texture = texture_handle(m ? m->find("emissiveTexture") : nullptr);
        out.metallic_roughness_texture = texture_handle(pbr->find("metallicRoughnessTexture"));
        out.normal_map_texture = texture_handle(m ? m->find("normalTexture") : nullptr);
        check(st_material_insert(engine, opt.first_handle + index, &out), "st_material_insert");
        sum.materials++;
    }

    void materials() {
        const Json* mats = doc.root.find("materials");
        const size_t n = mats ? mats->size() : 0;
        for (size_t i = 0; i < n; i++) insert_material(i, &mats->arr[i]);
        // primitives without a material get glTF's default material (spec 3.9.6), made once, after the file's own
        bool needed = false;
        if (const Json* meshes = doc.root.find("meshes"))
            for (auto& mesh : meshes->arr)
                if (const Json* prims = mesh.find("primitives"))
                    for (auto& p : prims->arr) needed = needed || !p.find("material");
        if (needed) {
            default_material = n;
            insert_material(n, nullptr);
        }
    }

    void primitive(const Json& prim, const Mat4d& world) {
        if (prim.index_or("mode", 4) != 4) { sum.primitives_skipped++; return; }
        if (const Json* ext = prim.find("extensions"))
            if (ext->find("KHR_draco_mesh_compression")) bad(ST_ERR_UNSUPPORTED, "glTF: Draco-compressed primitives are not supported");
        const Json& attrs = prim.at("attributes");
        const AccessorView pos = accessor(doc, attrs.at("POSITION").index("POSITION"));
        if (pos.ncomp != 3) PARSE_FAIL("glTF: POSITION must be VEC3");
        AccessorView nrm, uv, tan;
        const Json* nj = attrs.find("NORMAL");
        const Json* uj = attrs.find("TEXCOORD_0");
        const Json* tj = attrs.find("TANGENT");
        if (nj) { nrm = accessor(doc, nj->index("NORMAL")); if (nrm.ncomp != 3 || nrm.count < pos.count) PARSE_FAIL("glTF: NORMAL accessor does not match POSITION"); }
        if (uj) { uv = accessor(doc, uj->index("TEXCOORD_0")); if (uv.ncomp != 2 || uv.count < pos.count) PARSE_FAIL("glTF: TEXCOORD_0 accessor does not match POSITION"); }
        if (tj) { tan = accessor(doc, tj->index("TANGENT")); if (tan.ncomp != 4 || tan.count < pos.count) PARSE_FAIL("glTF: TANGENT accessor does not match POSITION"); }
        AccessorView idx;
        const Json* ij = prim.find("indices");
        if (ij) { idx = accessor(doc, ij->index("indices")); if (idx.ncomp != 1) PARSE_FAIL("glTF: indices must be SCALAR"); }
        const size_t n_idx = (ij ? idx.count : pos.count) / 3 * 3;
        std::vector<StMeshTriangle> tris(n_idx / 3);
        for (size_t t = 0; t < tris.size(); t++) {
            StMeshTriangle& tri = tris[t];
            memset(&tri, 0, sizeof tri);
            for (int k = 0; k < 3; k++) {
                const size_t v = ij ? accessor_index(idx, t * 3 + (size_t)k) : t * 3 + (size_t)k;
                if (v >= pos.count) PARSE_FAIL("glTF: vertex index %zu out of range (%zu vertices)", v, pos.count);
                for (int c = 0; c < 3; c++) tri.positions[k][c] = accessor_float(pos, v, c);
                if (nj) for (int c = 0; c < 3; c++) tri.normals[k][c] = accessor_float(nrm, v, c);
                if (uj) for (int c = 0; c < 2; c++) tri.uvs[k][c] = accessor_float(uv, v, c);
                if (tj) for (int c = 0; c < 4; c++) tri.tangents[k][c] = accessor_float(tan, v, c);
            }
            if (!nj) {  // flat normal: normalize(cross(p1 - p0, p2 - p0)), f32
                float e1[3], e2[3];
                for (int c = 0; c < 3; c++) { e1[c] = tri.positions[1][c] - tri.positions[0][c]; e2[c] = tri.positions[2][c] - tri.positions[0][c]; }
                float nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const float len = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
                if (len > 0.0f) for (int c = 0; c < 3; c++) nn[c] = nn[c] / len;
                for (int k = 0; k < 3; k++) memcpy(tri.normals[k], nn, sizeof nn);
            }
        }
        if (tris.empty()) { sum.primitives_skipped++; return; }  // an empty mesh would assert in the reference (triangles.rs:50-53)
        subdivide(tris, opt.subdivide);
        const StHandle handle = opt.first_handle + sum.meshes;
        check(st_mesh_insert(engine, handle, tris.data(), tris.size()), "st_mesh_insert");
        float xform[12];
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) xform[c * 3 + r] = (float)world.m[r][c];
        const Json* mj = prim.find("material");
        const size_t material = mj ? mj->index("primitive.material") : default_material;
        if (mj && material >= doc.root.at("materials").size()) PARSE_FAIL("glTF: primitive refers to material %zu", material);
        check(st_instance_insert(engine, handle, handle, opt.first_handle + material, xform), "st_instance_insert");
        sum.meshes++;
        sum.triangles += (uint32_t)tris.size();
    }

    void node(size_t index, const Mat4d& parent, int depth) {
        if (depth > 256) PARSE_FAIL("glTF: node hierarchy deeper than 256 (a cycle?)");
        const Json& n = doc.root.at("nodes").at(index, "node");
        const Mat4d world = mul(parent, node_matrix(n));
        if (const Json* mj = n.find("mesh")) {
            const Json& mesh = doc.root.at("meshes").at(mj->index("node.mesh"), "mesh");
            for (auto& prim : mesh.at("primitives").arr) primitive(prim, world);
        }
        if (const Json* ext = n.find("extensions"))
            if (const Json* lp = ext->find("KHR_lights_punctual"))
                if (const Json* li = lp->find("light")) light(li->index("node light"), world);
        if (const Json* kids = n.find("children"))
            for (auto& k : kids->arr) node(k.index("node child"), world, depth + 1);
    }

    // KHR_lights_punctual -> Light (light.rs:6-22) the way it would arrive through Bevy: bevy_gltf turns a point or spot light's
    // candela into lumens (x 4 pi) and leaves PointLight::radius at its default 0; bevy-strolle's extract stage turns lumens
    // back into colour x lumens / (4 pi), takes the position from the node's world transform and, for spots, the direction
    // -(rotation x Z) and the outer cone angle (bevy-strolle/src/stages/extract.rs:283-327), dropping lights fainter than
    // 0.0001. A range the file does not give is Bevy's default 20. Directional lights have no counterpart: strolle's sun is
    // a resource of its own (st_sun_update), so they are counted and skipped.
    void light(size_t index, const Mat4d& world) {
        const Json* ext = doc.root.find("extensions");
        const Json* lp = ext ? ext->find("KHR_lights_punctual") : nullptr;
        if (!lp) PARSE_FAIL("glTF: a node names a light but the document has no KHR_lights_punctual");
        const Json& l = lp->at("lights").at(index, "light");
        const std::string type = l.string_or("type", "");
        if (type == "directional") { sum.lights_skipped++; return; }
        if (type != "point" && type != "spot") PARSE_FAIL("glTF: light type \"%s\"", type.c_str());
        double color[3] = {1, 1, 1};
        if (const Json* c = l.find("color")) {
            if (c->size() != 3) PARSE_FAIL("glTF: light colour needs 3 numbers");
            for (int i = 0; i < 3; i++) color[i] = c->arr[(size_t)i].number("light colour");
        }
        const float candela = (float)l.number_or("intensity", 1.0);
        if (candela < 0.0001f) { sum.lights_skipped++; return; }  // extract.rs:287-290
        StLight out;
        memset(&out, 0, sizeof out);
        out.kind = type == "spot" ? ST_LIGHT_SPOT : ST_LIGHT_POINT;
        for (int i = 0; i < 3; i++) { out.position[i] = (float)world.m[i][3]; out.color[i] = (float)color[i] * candela; }
        out.radius = opt.light_radius;
        out.range = (float)l.number_or("range", 20.0);
        if (out.kind == ST_LIGHT_SPOT) {
            double d[3] = {-world.m[0][2], -world.m[1][2], -world.m[2][2]};  // the node's -Z axis in world space
            const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (!(len > 0.0)) PARSE_FAIL("glTF: spot light on a node with a degenerate transform");
            for (int i = 0; i < 3; i++) out.direction[i] = (float)(d[i] / len);
            const Json* spot = l.find("spot");
            out.angle = (float)(spot ? spot->number_or("outerConeAngle", 0.7853981633974483) : 0.7853981633974483);
        }
        check(st_light_insert(engine, opt.first_handle + sum.lights, &out), "st_light_insert");
        sum.lights++;
    }

    void run() {
        images();
        materials();
        const Json* scenes = doc.root.find("scenes");
        if (!scenes || scenes->size() == 0) return;
        const Json& scene = scenes->at(doc.root.index_or("scene", 0), "scene");
        if (const Json* roots = scene.find("nodes"))
            for (auto& r : roots->arr) node(r.index("scene node"), Mat4d::identity(), 0);
    }
};

int load(StEngine* e, const uint8_t* bytes, size_t size, const std::string& base_dir, const StGltfOptions* options, StGltfSummary* summary) {
    StGltfOptions opt;
    memset(&opt, 0, sizeof opt);
    opt.first_handle = 1;
    opt.first_image_handle = 1000;
    if (options) opt = *options;
    if (opt.subdivide > 6) return st_internal_fail(ST_ERR_INVALID_ARGUMENT, "subdivide > 6 (4096 triangles per triangle)");
    try {
        const Document doc = open_document(bytes, size, base_dir);
        Loader loader{e, doc, opt};
        loader.run();
        if (summary) *summary = loader.sum;
        return ST_OK;
    } catch (const IngestError& err) {
        return st_internal_fail(err.status, err.message.c_str());
    } catch (const std::bad_alloc&) {
        return st_internal_fail(ST_ERR_PARSE, "out of memory while reading the scene file");
    }
}

}  // namespace

extern "C" {

int st_scene_load_gltf(StEngine* e, const char* path, const StGltfOptions* options, StGltfSummary* summary) {
    if (!e || !path) return st_internal_fail(ST_ERR_INVALID_ARGUMENT, "null argument");
    try {
        const Bytes file = read_file(path);
        std::string dir(path);
        const size_t slash = dir.find_last_of('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash == 0 ? 1 : slash);
        return load(e, file.data(), file.size(), dir, options, summary);
    } catch (const IngestError& err) {
        return st_internal_fail(err.status, err.message.c_str());
    }
}

int st_scene_load_gltf_memory(StEngine* e, const void* bytes, size_t size, const char* base_dir, const StGltfOptions* options, StGltfSummary* summary) {
    if (!e || !bytes || !size) return st_internal_fail(ST_ERR_INVALID_ARGUMENT, "null argument");
    return load(e, (const uint8_t*)bytes, size, base_dir ? base_dir : "", options, summary);
}

static int decode_to(const void* bytes, size_t size, uint8_t* out_rgba, size_t capacity, uint32_t* width, uint32_t* height, bool png_only) {
    if (!bytes || !width || !height) return st_internal_fail(ST_ERR_INVALID_ARGUMENT, "null argument");
    try {
        const Picture pic = png_only ? decode_png((const uint8_t*)bytes, size) : decode_image((const uint8_t*)bytes, size, "image");
        *width = pic.width;
        *height = pic.height;
        if (!out_rgba) return ST_OK;  // size query
        if (capacity < pic.rgba.size()) return st_internal_fail(ST_ERR_INVALID_ARGUMENT, "output buffer too small");
        memcpy(out_rgba, pic.rgba.data(), pic.rgba.size());
        return ST_OK;
    } catch (const IngestError& err) {
        return st_internal_fail(err.status, err.message.c_str());
    } catch (const std::bad_alloc&) {
        return st_internal_fail(ST_ERR_PARSE, "out of memory while decoding the image");
    }
}
int st_decode_png(const void* bytes, size_t size, uint8_t* out_rgba, size_t capacity, uint32_t* width, uint32_t* height) {
    return decode_to(bytes, size, out_rgba, capacity, width, height, true);
}
int st_decode_image(const void* bytes, size_t size, uint8_t* out_rgba, size_t capacity, uint32_t* width, uint32_t* height) {
    return decode_to(bytes, size, out_rgba, capacity, width, height, false);
}

}  // extern "C"
