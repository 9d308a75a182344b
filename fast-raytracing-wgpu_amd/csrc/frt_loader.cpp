// frt_loader.cpp — see frt_loader.hpp. Restates src/scene/loader.rs:9-181 (load_gltf), src/scene/builder.rs:191-314
// (add_gltf_materials / _meshes / _instances) and src/scene/scenes.rs:246-322 (create_gltf_scene). The crates the reference leans on
// are not in /root/reference; what is restated of them is their documented behaviour:
//   gltf 1.4.1   gltf::import: JSON / GLB container, buffers (file, data: URI, BIN chunk), accessor reads incl. byteStride and the
//                normalised-integer -> f32 conversion of tex-coords (`into_f32`), image decode to R8G8B8 / R8G8B8A8
//   image 0.25.9 PNG decode; DynamicImage::resize_exact(.., Lanczos3) = vertical then horizontal pass in f32
//                baseline JPEG decode (the crate's decoder, zune-jpeg, differs from any other decoder by +-1 in places: the
//                IDCT and the chroma upsampling are not bit-specified by the standard)
// Arithmetic-coded JPEGs, grey / 16-bit / interlaced PNGs are not decoded: such an image becomes the white 1024 x 1024 texture the
// reference substitutes for formats it does not handle (loader.rs:35-44), with a warning.
#include "frt_loader.hpp"
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <fstream>
#include <sstream>

namespace frt {

static const uint32_t kTexW = 1024, kTexH = 1024;   // src/scene/mod.rs:12-13

// ================================================================================================ JSON
namespace {
struct JVal {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    bool b = false; double n = 0.0; std::string s;
    std::vector<JVal> a; std::vector<std::pair<std::string, JVal>> o;
    const JVal* get(const char* key) const {
        if (type != Obj) return nullptr;
        for (auto& kv : o) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_num() const { return type == Num; }
    double num(const char* key, double def) const { const JVal* v = get(key); return (v && v->type == Num) ? v->n : def; }
    long long integer(const char* key, long long def) const { const JVal* v = get(key); return (v && v->type == Num) ? (long long)v->n : def; }
    std::string str(const char* key, const std::string& def = "") const { const JVal* v = get(key); return (v && v->type == Str) ? v->s : def; }
    const std::vector<JVal>& arr(const char* key) const { static const std::vector<JVal> e; const JVal* v = get(key); return (v && v->type == Arr) ? v->a : e; }
};

struct JParser {
    const char* p; const char* e; std::string err; int depth = 0;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    static void utf8(std::string& s, uint32_t c) {
        if (c < 0x80) s += (char)c;
        else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
        else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 0x3F)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t& v) {
        if (e - p < 4) return fail("json: short \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++; v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0'); else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10); else return fail("json: bad \\u escape");
        }
        return true;
    }
    bool string(std::string& out) {
        if (p >= e || *p != '"') return fail("json: expected string");
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) return fail("json: dangling escape");
                char c = *p++;
                switch (c) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    uint32_t c1; if (!hex4(c1)) return false;
                    if (c1 >= 0xD800 && c1 < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        p += 2; uint32_t c2; if (!hex4(c2)) return false;
                        c1 = 0x10000 + ((c1 - 0xD800) << 10) + (c2 - 0xDC00);
                    }
                    utf8(out, c1); break;
                }
                default: return fail("json: unknown escape");
                }
            } else out += *p++;
        }
        if (p >= e) return fail("json: unterminated string");
        ++p;
        return true;
    }
    bool value(JVal& v) {
        if (++depth > 128) return fail("json: nesting too deep");
        ws();
        if (p >= e) return fail("json: unexpected end");
        bool ok = true;
        if (*p == '{') {
            v.type = JVal::Obj; ++p; ws();
            if (p < e && *p == '}') ++p;
            else for (;;) {
                ws(); std::string k;
                if (!string(k)) { ok = false; break; }
                ws(); if (p >= e || *p != ':') { ok = fail("json: expected ':'"); break; }
                ++p; v.o.emplace_back(k, JVal());
                if (!value(v.o.back().second)) { ok = false; break; }
                ws(); if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; break; }
                ok = fail("json: expected ',' or '}'"); break;
            }
        } else if (*p == '[') {
            v.type = JVal::Arr; ++p; ws();
            if (p < e && *p == ']') ++p;
            else for (;;) {
                v.a.emplace_back();
                if (!value(v.a.back())) { ok = false; break; }
                ws(); if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; break; }
                ok = fail("json: expected ',' or ']'"); break;
            }
        } else if (*p == '"') { v.type = JVal::Str; ok = string(v.s); }
        else if (e - p >= 4 && !strncmp(p, "true", 4)) { v.type = JVal::Bool; v.b = true; p += 4; }
        else if (e - p >= 5 && !strncmp(p, "false", 5)) { v.type = JVal::Bool; v.b = false; p += 5; }
        else if (e - p >= 4 && !strncmp(p, "null", 4)) { v.type = JVal::Null; p += 4; }
        else {
            const char* q = p;
            while (q < e && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9'))) ++q;
            if (q == p) ok = fail("json: unexpected character");
            else { std::string t(p, q); char* end = nullptr; v.n = strtod(t.c_str(), &end); v.type = JVal::Num; if (!end || *end) ok = fail("json: bad number"); p = q; }
        }
        --depth;
        return ok;
    }
};

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end); std::streamoff n = f.tellg(); f.seekg(0);
    if (n < 0) return false;
    out.resize((size_t)n);
    if (n) f.read(reinterpret_cast<char*>(out.data()), n);
    return (bool)f || n == 0;
}
std::string dir_of(const std::string& path) { size_t k = path.find_last_of("/\\"); return k == std::string::npos ? std::string() : path.substr(0, k + 1); }

bool base64_decode(const char* p, size_t n, std::vector<uint8_t>& out) {
    uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        char c = p[i]; int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62; else if (c == '/' || c == '_') v = 63; else if (c == '=' || c == '\n' || c == '\r') continue; else return false;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}
std::string percent_decode(const std::string& s) {
    std::string o;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '%' && i + 2 < s.size() + 0 && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
            o += (char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16); i += 2;
        } else o += s[i];
    }
    return o;
}
// uri: "data:<mime>;base64,<payload>" or a path relative to the .gltf file
bool load_uri(const std::string& uri, const std::string& base_dir, std::vector<uint8_t>& out, std::string& err) {
    if (uri.compare(0, 5, "data:") == 0) {
        size_t k = uri.find(',');
        if (k == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > k) { err = "data URI is not base64"; return false; }
        if (!base64_decode(uri.data() + k + 1, uri.size() - k - 1, out)) { err = "bad base64 payload"; return false; }
        return true;
    }
    std::string path = base_dir + percent_decode(uri);
    if (!read_file(path, out)) { err = "cannot read " + path; return false; }
    return true;
}
} // namespace

// ================================================================================================ PNG
namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
}

bool decode_png(const uint8_t* d, size_t n, std::vector<uint8_t>& px, uint32_t& w, uint32_t& h, uint32_t& channels, std::string& why) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(d, sig, 8)) { why = "not a PNG"; return false; }
    size_t pos = 8;
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, done = false;
    while (!done && pos + 12 <= n) {
        uint32_t len = be32(d + pos); const uint8_t* tag = d + pos + 4; const uint8_t* body = d + pos + 8;
        if ((size_t)len > n - pos - 12) { why = "truncated PNG chunk"; return false; }
        if (!memcmp(tag, "IHDR", 4)) {
            if (len < 13) { why = "short IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; have_ihdr = true;
        } else if (!memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
        else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(tag, "IEND", 4)) done = true;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || w > 16384 || h > 16384 || (uint64_t)w * h > (1ull << 26)) { why = "bad or oversized PNG header"; return false; }
    if (interlace) { why = "interlaced PNG (not decoded in this build)"; return false; }
    uint32_t src_ch;
    if (ctype == 2 && depth == 8) src_ch = 3;
    else if (ctype == 6 && depth == 8) src_ch = 4;
    else if (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) src_ch = 1;
    else { why = "PNG colour type " + std::to_string(ctype) + " / depth " + std::to_string(depth) + " decodes to a format other than R8G8B8 / R8G8B8A8"; return false; }
    const size_t row_bits = (size_t)w * src_ch * depth, row_bytes = (row_bits + 7) / 8, bpp = (src_ch * depth + 7) / 8;
    std::vector<uint8_t> raw((row_bytes + 1) * (size_t)h);
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) { why = "zlib init"; return false; }
    zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size(); zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
    int zr = inflate(&zs, Z_FINISH);
    size_t got = raw.size() - zs.avail_out;
    inflateEnd(&zs);
    if ((zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR) || got != raw.size()) { why = "PNG data stream is corrupt or short"; return false; }
    std::vector<uint8_t> prev(row_bytes, 0), cur(row_bytes);
    const bool palette = ctype == 3;
    const bool key = ctype == 2 && trns.size() >= 6;
    channels = palette ? (trns.empty() ? 3u : 4u) : (ctype == 6 || key ? 4u : 3u);
    if (palette && plte.size() < 3) { why = "palette PNG without PLTE"; return false; }
    px.assign((size_t)w * h * channels, 255);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* r = raw.data() + (size_t)y * (row_bytes + 1);
        const uint8_t ft = r[0]; ++r;
        if (ft > 4) { why = "bad PNG filter"; return false; }
        for (size_t i = 0; i < row_bytes; ++i) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = r[i];
            int v = ft == 0 ? x : ft == 1 ? x + a : ft == 2 ? x + b : ft == 3 ? x + ((a + b) >> 1) : x + paeth(a, b, c);
            cur[i] = (uint8_t)v;
        }
        uint8_t* o = px.data() + (size_t)y * w * channels;
        if (palette) {
            for (uint32_t x = 0; x < w; ++x) {
                size_t bit = (size_t)x * depth; uint32_t idx = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                if ((size_t)idx * 3 + 2 >= plte.size()) idx = 0;
                o[x * channels + 0] = plte[idx * 3]; o[x * channels + 1] = plte[idx * 3 + 1]; o[x * channels + 2] = plte[idx * 3 + 2];
                if (channels == 4) o[x * 4 + 3] = idx < trns.size() ? trns[idx] : 255;
            }
        } else if (key) {
            for (uint32_t x = 0; x < w; ++x) {
                const uint8_t* s = &cur[(size_t)x * 3];
                bool t = s[0] == trns[1] && s[1] == trns[3] && s[2] == trns[5] && trns[0] == 0 && trns[2] == 0 && trns[4] == 0;
                o[x * 4] = s[0]; o[x * 4 + 1] = s[1]; o[x * 4 + 2] = s[2]; o[x * 4 + 3] = t ? 0 : 255;
            }
        } else memcpy(o, cur.data(), (size_t)w * channels);
        prev.swap(cur);
    }
    return true;
}

// ================================================================================================ JPEG (baseline + progressive Huffman)
// Sequential Huffman JPEG, 8-bit, three components (what gltf's importer hands the reference as R8G8B8): ITU-T T.81 decoding,
// float separable IDCT, libjpeg-style triangle ("fancy") chroma upsampling for 2x1 / 2x2 subsampling, JFIF YCbCr -> RGB.
// Progressive Huffman files (spectral selection + successive approximation) are decoded too. Arithmetic / 12-bit / lossless files
// and 1- or 4-component images are reported as unsupported (the caller falls back to the white texture); single-component (grey) JPEGs are exactly the case the reference itself rejects (Format::R8).
namespace {
struct JHuff {
    uint8_t vals[256]; int maxcode[18], valptr[17], mincode[17]; bool present = false;
    uint16_t fast[512];       // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
    bool build(const uint8_t counts[16], const uint8_t* symbols, int n) {
        present = false;
        memcpy(vals, symbols, (size_t)n);
        int code = 0, k = 0;
        memset(fast, 0, sizeof fast);
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k; mincode[len] = code;
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
                if (code >= (1 << len)) return false;            // over-subscribed table
                if (len <= 9) for (int f = 0; f < (1 << (9 - len)); ++f) fast[(code << (9 - len)) | f] = (uint16_t)((len << 8) | vals[k]);
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
        return true;
    }
};
struct JBits {
    const uint8_t* p; const uint8_t* e; uint32_t acc = 0; int n = 0; bool marker = false;
    void fill() {
        while (n <= 24) {
            uint32_t b = 0;
            if (!marker && p < e) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < e && p[1] == 0x00) p += 2;          // stuffed zero
                    else { marker = true; b = 0; }                   // a marker: feed zeros until the caller handles it
                } else ++p;
            }
            acc |= b << (24 - n); n += 8;
        }
    }
    int peek(int k) { fill(); return (int)(acc >> (32 - k)); }
    void skip(int k) { acc <<= k; n -= k; }
    int get(int k) { if (k == 0) return 0; int v = peek(k); skip(k); return v; }
    void reset() { acc = 0; n = 0; marker = false; }
};
int jdecode(JBits& b, const JHuff& h) {
    int look = b.peek(9);
    uint16_t f = h.fast[look];
    if (f) { b.skip(f >> 8); return f & 0xFF; }
    int code = b.peek(16);
    for (int len = 10; len <= 16; ++len) {
        int c = code >> (16 - len);
        if (h.maxcode[len] >= 0 && c <= h.maxcode[len] && c >= h.mincode[len]) { b.skip(len); return h.vals[h.valptr[len] + c - h.mincode[len]]; }
    }
    return -1;
}
int jextend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
struct JComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; std::vector<int16_t> coef; };   // coef: progressive only

void jidct(const float in[64], uint8_t* out, size_t pitch) {
    static float c[8][8]; static bool init = false;
    if (!init) { for (int x = 0; x < 8; ++x) for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? 0.35355339059327379f : 0.5f) * cosf((2 * x + 1) * u * 3.14159265358979323846f / 16.0f); init = true; }
    float tmp[64];
    for (int v = 0; v < 8; ++v) for (int x = 0; x < 8; ++x) { float s = 0; for (int u = 0; u < 8; ++u) s += c[x][u] * in[v * 8 + u]; tmp[v * 8 + x] = s; }
    for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
        float s = 0; for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
        s = floorf(s + 128.5f);
        out[(size_t)y * pitch + x] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
    }
}
// plane (pw x ph) -> full resolution (fw x fh, fw = pw * fx, fh = ph * fy)
void jupsample(const std::vector<uint8_t>& in, int pw, int ph, int fx, int fy, std::vector<uint8_t>& out) {
    const int fw = pw * fx, fh = ph * fy;
    out.resize((size_t)fw * fh);
    if (fx == 1 && fy == 1) { out = in; return; }
    if (fx == 2 && fy == 1) {
        for (int y = 0; y < ph; ++y) {
            const uint8_t* r = &in[(size_t)y * pw]; uint8_t* o = &out[(size_t)y * fw];
            for (int x = 0; x < pw; ++x) {
                int c = r[x], l = r[x > 0 ? x - 1 : 0], n = r[x + 1 < pw ? x + 1 : pw - 1];
                o[2 * x] = (uint8_t)(x == 0 ? c : (3 * c + l + 1) >> 2);
                o[2 * x + 1] = (uint8_t)(x + 1 == pw ? c : (3 * c + n + 2) >> 2);
            }
        }
        return;
    }
    if (fx == 2 && fy == 2) {
        std::vector<int> sum((size_t)pw);
        for (int oy = 0; oy < fh; ++oy) {
            int y0 = oy >> 1, y1 = (oy & 1) ? y0 + 1 : y0 - 1;
            y1 = y1 < 0 ? 0 : (y1 >= ph ? ph - 1 : y1);
            const uint8_t* a = &in[(size_t)y0 * pw]; const uint8_t* b = &in[(size_t)y1 * pw];
            for (int x = 0; x < pw; ++x) sum[(size_t)x] = 3 * a[x] + b[x];
            uint8_t* o = &out[(size_t)oy * fw];
            for (int x = 0; x < pw; ++x) {
                int t = sum[(size_t)x], l = sum[(size_t)(x > 0 ? x - 1 : 0)], n = sum[(size_t)(x + 1 < pw ? x + 1 : pw - 1)];
                o[2 * x] = (uint8_t)(x == 0 ? (t * 4 + 8) >> 4 : (3 * t + l + 8) >> 4);
                o[2 * x + 1] = (uint8_t)(x + 1 == pw ? (t * 4 + 7) >> 4 : (3 * t + n + 7) >> 4);
            }
        }
        return;
    }
    for (int y = 0; y < fh; ++y) for (int x = 0; x < fw; ++x) out[(size_t)y * fw + x] = in[(size_t)(y / fy) * pw + x / fx];   // other ratios: replicate
}
} // namespace

bool decode_jpeg(const uint8_t* d, size_t n, std::vector<uint8_t>& rgb, uint32_t& W, uint32_t& H, std::string& why) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { why = "not a JPEG"; return false; }
    uint16_t qt[4][64]; bool have_qt[4] = {false, false, false, false};
    JHuff hdc[4], hac[4];
    std::vector<JComp> comps;
    int hmax = 1, vmax = 1, restart = 0, mcus_x = 0, mcus_y = 0;
    bool have_sof = false, adobe = false, progressive = false; int adobe_transform = -1;
    size_t pos = 2;
    bool decoded_any = false;
    while (pos + 4 <= n) {
        if (d[pos] != 0xFF) { ++pos; continue; }
        uint8_t m = d[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        pos += 2;
        if (m == 0xD9) break;                                            // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;             // TEM, stray RSTn
        if (pos + 2 > n) break;
        size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
        if (len < 2 || pos + len > n) { why = "truncated JPEG segment"; return false; }
        const uint8_t* s = d + pos + 2; size_t sl = len - 2;
        if (m == 0xDB) {                                                  // DQT
            size_t k = 0;
            while (k < sl) {
                int pq = s[k] >> 4, tq = s[k] & 15; ++k;
                if (tq > 3 || k + (pq ? 128u : 64u) > sl) { why = "bad DQT"; return false; }
                for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[k] << 8) | s[k + 1]) : s[k]; k += pq ? 2 : 1; }
                have_qt[tq] = true;
            }
        } else if (m == 0xC4) {                                           // DHT
            size_t k = 0;
            while (k + 17 <= sl) {
                int tc = s[k] >> 4, th = s[k] & 15; ++k;
                int total = 0; for (int i = 0; i < 16; ++i) total += s[k + i];
                if (th > 3 || tc > 1 || total > 256 || k + 16 + (size_t)total > sl) { why = "bad DHT"; return false; }
                if (!(tc ? hac[th] : hdc[th]).build(s + k, s + k + 16, total)) { why = "bad DHT"; return false; }
                k += 16 + (size_t)total;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {                 // SOF0 / SOF1: sequential Huffman; SOF2: progressive Huffman
            progressive = m == 0xC2;
            if (sl < 6 || s[0] != 8) { why = "JPEG precision other than 8 bits"; return false; }
            H = ((uint32_t)s[1] << 8) | s[2]; W = ((uint32_t)s[3] << 8) | s[4];
            int nf = s[5];
            if (nf != 3) { why = std::to_string(nf) + "-component JPEG decodes to a format other than R8G8B8"; return false; }
            if (W == 0 || H == 0 || W > 16384 || H > 16384 || (uint64_t)W * H > (1ull << 26) || sl < 6 + 3 * (size_t)nf) { why = "bad or oversized JPEG frame header"; return false; }
            comps.assign((size_t)nf, JComp());
            for (int i = 0; i < nf; ++i) {
                comps[(size_t)i].id = s[6 + 3 * i]; comps[(size_t)i].h = s[7 + 3 * i] >> 4; comps[(size_t)i].v = s[7 + 3 * i] & 15; comps[(size_t)i].tq = s[8 + 3 * i] & 3;
                if (comps[(size_t)i].h < 1 || comps[(size_t)i].h > 4 || comps[(size_t)i].v < 1 || comps[(size_t)i].v > 4) { why = "bad JPEG sampling factors"; return false; }
                hmax = std::max(hmax, comps[(size_t)i].h); vmax = std::max(vmax, comps[(size_t)i].v);
            }
            mcus_x = (int)((W + 8u * (uint32_t)hmax - 1u) / (8u * (uint32_t)hmax)); mcus_y = (int)((H + 8u * (uint32_t)vmax - 1u) / (8u * (uint32_t)vmax));
            for (JComp& c : comps) {
                if (hmax % c.h || vmax % c.v) { why = "fractional JPEG sampling ratio"; return false; }
                c.bw = mcus_x * c.h; c.bh = mcus_y * c.v; c.plane.assign((size_t)c.bw * 8 * (size_t)c.bh * 8, 128);
                if (progressive) c.coef.assign((size_t)c.bw * (size_t)c.bh * 64, 0);
            }
            have_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            why = "JPEG process other than baseline / progressive Huffman (not decoded in this build)"; return false;
        } else if (m == 0xDD) { if (sl >= 2) restart = (s[0] << 8) | s[1]; }
        else if (m == 0xEE && sl >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; }
        else if (m == 0xDA) {                                             // SOS + entropy-coded data
            if (!have_sof || sl < 1) { why = "JPEG scan before frame header"; return false; }
            int ns = s[0];
            if (ns < 1 || ns > 3 || sl < 1 + 2 * (size_t)ns + 3) { why = "bad JPEG scan header"; return false; }
            std::vector<JComp*> sc;
            for (int i = 0; i < ns; ++i) {
                JComp* c = nullptr;
                for (JComp& k : comps) if (k.id == s[1 + 2 * i]) c = &k;
                if (!c) { why = "JPEG scan names an unknown component"; return false; }
                c->td = s[2 + 2 * i] >> 4; c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !have_qt[c->tq]) { why = "JPEG scan uses a missing table"; return false; }
                c->pred = 0; sc.push_back(c);
            }
            const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
            if (progressive) {
                if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) { why = "bad progressive scan parameters"; return false; }
            }
            for (JComp* c : sc) {
                const bool need_dc = !progressive || (Ss == 0 && Ah == 0), need_ac = !progressive || Ss > 0;
                if ((need_dc && !hdc[c->td].present) || (need_ac && !hac[c->ta].present)) { why = "JPEG scan uses a missing table"; return false; }
            }
            JBits b; b.p = d + pos + len; b.e = d + n;
            int eobrun = 0;
            // one block of a progressive scan (ITU-T T.81 G.1.2; the refinement pass as in libjpeg's jdphuff.c)
            auto pblock = [&](JComp& c, int bx, int by) -> bool {
                int16_t* co = &c.coef[((size_t)by * (size_t)c.bw + (size_t)bx) * 64];
                if (Ss == 0) {
                    if (Ah == 0) {
                        int t = jdecode(b, hdc[c.td]);
                        if (t < 0 || t > 11) return false;
                        c.pred += jextend(b.get(t), t);
                        co[0] = (int16_t)(c.pred * (1 << Al));
                    } else if (b.get(1)) co[0] = (int16_t)(co[0] | (1 << Al));
                    return true;
                }
                if (Ah == 0) {
                    if (eobrun > 0) { --eobrun; return true; }
                    for (int k = Ss; k <= Se;) {
                        int rs = jdecode(b, hac[c.ta]);
                        if (rs < 0) return false;
                        int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += b.get(r); break; }
                            k += 16;
                        } else {
                            k += r;
                            if (k > Se) return false;
                            co[kZigzag[k]] = (int16_t)(jextend(b.get(sz), sz) * (1 << Al));
                            ++k;
                        }
                    }
                    return true;
                }
                const int p1 = 1 << Al, m1 = -(1 << Al);
                int k = Ss;
                if (eobrun == 0) {
                    for (; k <= Se; ++k) {
                        int rs = jdecode(b, hac[c.ta]);
                        if (rs < 0) return false;
                        int r = rs >> 4, sz = rs & 15, val = 0;
                        if (sz == 0) {
                            if (r < 15) { eobrun = 1 << r; if (r) eobrun += b.get(r); break; }
                        } else val = b.get(1) ? p1 : m1;     // sz must be 1 here
                        while (k <= Se) {
                            int16_t& cv = co[kZigzag[k]];
                            if (cv != 0) { if (b.get(1) && (cv & p1) == 0) cv = (int16_t)(cv + (cv >= 0 ? p1 : m1)); }
                            else if (--r < 0) break;
                            ++k;
                        }
                        if (val && k <= Se) co[kZigzag[k]] = (int16_t)val;
                    }
                }
                if (eobrun > 0) {
                    for (; k <= Se; ++k) {
                        int16_t& cv = co[kZigzag[k]];
                        if (cv != 0 && b.get(1) && (cv & p1) == 0) cv = (int16_t)(cv + (cv >= 0 ? p1 : m1));
                    }
                    --eobrun;
                }
                return true;
            };
            auto block = [&](JComp& c, int bx, int by) -> bool {
                float coef[64]; for (float& f : coef) f = 0.0f;
                int t = jdecode(b, hdc[c.td]);
                if (t < 0 || t > 11) return false;
                c.pred += jextend(b.get(t), t);
                coef[0] = (float)(c.pred * (int)qt[c.tq][0]);
                for (int k = 1; k < 64;) {
                    int rs = jdecode(b, hac[c.ta]);
                    if (rs < 0) return false;
                    int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                    k += r;
                    if (k > 63) return false;
                    int nat = kZigzag[k];
                    coef[nat] = (float)(jextend(b.get(sz), sz) * (int)qt[c.tq][nat]);
                    ++k;
                }
                jidct(coef, &c.plane[((size_t)by * 8) * ((size_t)c.bw * 8) + (size_t)bx * 8], (size_t)c.bw * 8);
                return true;
            };
            int units_x, units_y;
            if (ns == 1) {   // non-interleaved: the component's own block grid, clipped to the image
                JComp& c = *sc[0];
                units_x = (int)(((W * (uint32_t)c.h + (uint32_t)hmax - 1u) / (uint32_t)hmax + 7u) / 8u);
                units_y = (int)(((H * (uint32_t)c.v + (uint32_t)vmax - 1u) / (uint32_t)vmax + 7u) / 8u);
            } else { units_x = mcus_x; units_y = mcus_y; }
            int count = 0, rst = 0;
            for (int uy = 0; uy < units_y; ++uy) for (int ux = 0; ux < units_x; ++ux) {
                if (restart && count == restart) {   // RSTn: byte-align, consume the marker, reset predictors
                    b.reset();
                    while (b.p + 1 < b.e && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) ++b.p;
                    if (b.p + 1 < b.e) b.p += 2;
                    (void)rst; ++rst;
                    for (JComp* c : sc) c->pred = 0;
                    eobrun = 0;
                    count = 0;
                }
                auto one = [&](JComp& c, int bx, int by) { return progressive ? pblock(c, bx, by) : block(c, bx, by); };
                if (ns == 1) { if (!one(*sc[0], ux, uy)) { why = "corrupt JPEG entropy data"; return false; } }
                else for (JComp* c : sc) for (int v = 0; v < c->v; ++v) for (int h = 0; h < c->h; ++h)
                    if (!one(*c, ux * c->h + h, uy * c->v + v)) { why = "corrupt JPEG entropy data"; return false; }
                ++count;
            }
            decoded_any = true;
            // continue after the entropy-coded segment: the reader stopped at (or before) the next marker
            pos = (size_t)(b.p - d);
            while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] != 0x00 && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7) && d[pos + 1] != 0xFF)) ++pos;
            continue;
        }
        pos += len;
    }
    if (!have_sof || !decoded_any) { why = "JPEG without image data"; return false; }
    if (progressive) {
        for (JComp& c : comps) {
            if (!have_qt[c.tq]) { why = "JPEG component without quantisation table"; return false; }
            for (int by = 0; by < c.bh; ++by) for (int bx = 0; bx < c.bw; ++bx) {
                const int16_t* co = &c.coef[((size_t)by * (size_t)c.bw + (size_t)bx) * 64];
                float f[64];
                for (int k = 0; k < 64; ++k) f[k] = (float)((int)co[k] * (int)qt[c.tq][k]);
                jidct(f, &c.plane[((size_t)by * 8) * ((size_t)c.bw * 8) + (size_t)bx * 8], (size_t)c.bw * 8);
            }
        }
    }
    std::vector<uint8_t> full[3];
    const int fw = mcus_x * hmax * 8;
    for (int i = 0; i < 3; ++i) jupsample(comps[(size_t)i].plane, comps[(size_t)i].bw * 8, comps[(size_t)i].bh * 8, hmax / comps[(size_t)i].h, vmax / comps[(size_t)i].v, full[i]);
    rgb.resize((size_t)W * H * 3);
    const bool is_rgb = adobe && adobe_transform == 0;
    for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) {
        size_t k = (size_t)y * (size_t)fw + x;
        float Y = full[0][k], cb = (float)full[1][k] - 128.0f, cr = (float)full[2][k] - 128.0f;
        float r = is_rgb ? Y : Y + 1.402f * cr, g = is_rgb ? (float)full[1][k] : Y - 0.344136f * cb - 0.714136f * cr, bl = is_rgb ? (float)full[2][k] : Y + 1.772f * cb;
        auto q = [](float v) { v = floorf(v + 0.5f); return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        uint8_t* o = &rgb[((size_t)y * W + x) * 3];
        o[0] = q(r); o[1] = q(g); o[2] = q(bl);
    }
    return true;
}

// ================================================================================================ Lanczos3 resize
namespace {
float sinc_(float t) { float a = t * 3.14159265358979323846f; return t == 0.0f ? 1.0f : sinf(a) / a; }   // image: imageops/sample.rs sinc
float lanczos3_(float x) { return fabsf(x) < 3.0f ? sinc_(x) * sinc_(x / 3.0f) : 0.0f; }             // lanczos(x, 3.0)
long long clampll(long long v, long long lo, long long hi) { return v < lo ? lo : (v > hi ? hi : v); }
// One axis of image's resize: `n_in` samples -> `n_out`, weights normalised per output sample.
struct AxisTap { uint32_t left; std::vector<float> w; };
std::vector<AxisTap> axis_taps(uint32_t n_in, uint32_t n_out) {
    std::vector<AxisTap> taps(n_out);
    const float ratio = (float)n_in / (float)n_out;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 3.0f * sratio;
    for (uint32_t o = 0; o < n_out; ++o) {
        float in = ((float)o + 0.5f) * ratio;
        long long left = clampll((long long)floorf(in - src_support), 0, (long long)n_in - 1);
        long long right = clampll((long long)ceilf(in + src_support), left + 1, (long long)n_in);
        in = in - 0.5f;
        AxisTap& t = taps[o]; t.left = (uint32_t)left;
        float sum = 0.0f;
        for (long long i = left; i < right; ++i) { float w = lanczos3_(((float)i - in) / sratio); t.w.push_back(w); sum += w; }
        for (float& w : t.w) w /= sum;
    }
    return taps;
}
}

void resize_lanczos3_rgba8(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh) {
    if (sw == dw && sh == dh) { memcpy(dst, src, (size_t)sw * sh * 4); return; }   // image: same size -> plain copy
    // vertical pass -> f32 intermediate of sw x dh (unclamped), then horizontal pass -> clamp, round half away, u8
    std::vector<float> tmp((size_t)sw * dh * 4);
    std::vector<AxisTap> vt = axis_taps(sh, dh);
    for (uint32_t oy = 0; oy < dh; ++oy) {
        const AxisTap& t = vt[oy];
        for (uint32_t x = 0; x < sw; ++x) {
            float acc[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < t.w.size(); ++i) {
                const uint8_t* p = src + ((size_t)(t.left + i) * sw + x) * 4;
                for (int c = 0; c < 4; ++c) acc[c] += (float)p[c] * t.w[i];
            }
            memcpy(&tmp[((size_t)oy * sw + x) * 4], acc, sizeof acc);
        }
    }
    std::vector<AxisTap> ht = axis_taps(sw, dw);
    for (uint32_t ox = 0; ox < dw; ++ox) {
        const AxisTap& t = ht[ox];
        for (uint32_t y = 0; y < dh; ++y) {
            float acc[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < t.w.size(); ++i) {
                const float* p = &tmp[((size_t)y * sw + t.left + i) * 4];
                for (int c = 0; c < 4; ++c) acc[c] += p[c] * t.w[i];
            }
            uint8_t* o = dst + ((size_t)y * dw + ox) * 4;
            for (int c = 0; c < 4; ++c) { float v = acc[c] < 0.0f ? 0.0f : (acc[c] > 255.0f ? 255.0f : acc[c]); o[c] = (uint8_t)roundf(v); }
        }
    }
}

// ================================================================================================ glTF
namespace {
struct Doc {
    JVal root; std::string base_dir;
    std::vector<std::vector<uint8_t>> buffers;
};
struct Acc { const uint8_t* base = nullptr; size_t count = 0, stride = 0; int comp = 0, ncomp = 0; bool normalized = false; };

int type_components(const std::string& t) {
    if (t == "SCALAR") return 1; if (t == "VEC2") return 2; if (t == "VEC3") return 3; if (t == "VEC4") return 4;
    if (t == "MAT2") return 4; if (t == "MAT3") return 9; if (t == "MAT4") return 16; return 0;
}
int comp_size(int c) { return (c == 5120 || c == 5121) ? 1 : (c == 5122 || c == 5123) ? 2 : (c == 5125 || c == 5126) ? 4 : 0; }

bool get_accessor(const Doc& d, long long index, Acc& a, std::vector<uint8_t>& zeros, std::string& err) {
    const auto& accs = d.root.arr("accessors");
    if (index < 0 || (size_t)index >= accs.size()) { err = "accessor index out of range"; return false; }
    const JVal& j = accs[(size_t)index];
    if (j.get("sparse")) { err = "sparse accessors are not supported"; return false; }
    a.comp = (int)j.integer("componentType", 0); a.ncomp = type_components(j.str("type")); a.count = (size_t)j.integer("count", 0);
    const JVal* nz = j.get("normalized"); a.normalized = nz && nz->type == JVal::Bool && nz->b;
    const size_t esz = (size_t)comp_size(a.comp) * (size_t)a.ncomp;
    if (esz == 0) { err = "accessor with unknown component or element type"; return false; }
    long long bv = j.integer("bufferView", -1);
    if (bv < 0) {   // glTF: an accessor without a bufferView is all zeros
        if (a.count > (1u << 28)) { err = "accessor without bufferView is too large"; return false; }
        zeros.assign(esz * a.count, 0); a.base = zeros.data(); a.stride = esz; return true;
    }
    const auto& views = d.root.arr("bufferViews");
    if ((size_t)bv >= views.size()) { err = "bufferView index out of range"; return false; }
    const JVal& v = views[(size_t)bv];
    long long buf = v.integer("buffer", -1);
    if (buf < 0 || (size_t)buf >= d.buffers.size()) { err = "buffer index out of range"; return false; }
    const std::vector<uint8_t>& B = d.buffers[(size_t)buf];
    size_t voff = (size_t)v.integer("byteOffset", 0), vlen = (size_t)v.integer("byteLength", 0), aoff = (size_t)j.integer("byteOffset", 0);
    size_t stride = (size_t)v.integer("byteStride", 0); if (stride == 0) stride = esz;
    if (voff > B.size() || vlen > B.size() - voff) { err = "bufferView exceeds its buffer"; return false; }
    if (a.count && (aoff > vlen || a.count > vlen || stride > vlen || (a.count - 1) * stride + esz > vlen - aoff)) { err = "accessor exceeds its bufferView"; return false; }
    a.base = B.data() + voff + aoff; a.stride = stride;
    return true;
}
template <typename T> T rd(const uint8_t* p) { T v; memcpy(&v, p, sizeof v); return v; }

bool read_floats(const Acc& a, int want_comp, bool allow_normalized_ints, std::vector<float>& out, std::string& err, const char* what) {
    if (a.ncomp != want_comp) { err = std::string(what) + ": unexpected element type"; return false; }
    out.resize(a.count * (size_t)want_comp);
    for (size_t i = 0; i < a.count; ++i) {
        const uint8_t* p = a.base + i * a.stride;
        for (int c = 0; c < want_comp; ++c) {
            float v;
            if (a.comp == 5126) v = rd<float>(p + 4 * c);
            else if (allow_normalized_ints && a.comp == 5121) v = (float)p[c] / 255.0f;                   // gltf: u8 -> f32, x / 255
            else if (allow_normalized_ints && a.comp == 5123) v = (float)rd<uint16_t>(p + 2 * c) / 65535.0f;   // u16 -> f32, x / 65535
            else { err = std::string(what) + ": unsupported component type"; return false; }
            out[i * want_comp + c] = v;
        }
    }
    return true;
}

bool decode_image(const Doc& d, const JVal& img, std::vector<uint8_t>& rgba, uint32_t& w, uint32_t& h, std::string& why) {
    std::vector<uint8_t> bytes;
    std::string uri = img.str("uri");
    if (!uri.empty()) { if (!load_uri(uri, d.base_dir, bytes, why)) return false; }
    else {
        long long bv = img.integer("bufferView", -1);
        const auto& views = d.root.arr("bufferViews");
        if (bv < 0 || (size_t)bv >= views.size()) { why = "image without uri or bufferView"; return false; }
        const JVal& v = views[(size_t)bv];
        long long buf = v.integer("buffer", -1);
        if (buf < 0 || (size_t)buf >= d.buffers.size()) { why = "image bufferView names an unknown buffer"; return false; }
        const auto& B = d.buffers[(size_t)buf];
        size_t off = (size_t)v.integer("byteOffset", 0), len = (size_t)v.integer("byteLength", 0);
        if (off > B.size() || len > B.size() - off) { why = "image bufferView exceeds its buffer"; return false; }
        bytes.assign(B.begin() + (long)off, B.begin() + (long)(off + len));
    }
    std::vector<uint8_t> px; uint32_t ch = 0;
    if (bytes.size() >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) { if (!decode_jpeg(bytes.data(), bytes.size(), px, w, h, why)) return false; ch = 3; }
    else if (!decode_png(bytes.data(), bytes.size(), px, w, h, ch, why)) return false;
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0, n = (size_t)w * h; i < n; ++i) {   // DynamicImage::to_rgba8: RGB -> alpha 255
        rgba[i * 4] = px[i * ch]; rgba[i * 4 + 1] = px[i * ch + 1]; rgba[i * 4 + 2] = px[i * ch + 2]; rgba[i * 4 + 3] = ch == 4 ? px[i * 4 + 3] : 255;
    }
    return true;
}

uint32_t pack16(uint32_t cur, uint32_t val, bool high) {   // material.rs:77-84
    uint32_t v = val & 0xFFFFu;
    return high ? ((cur & 0x0000FFFFu) | (v << 16)) : ((cur & 0xFFFF0000u) | v);
}
// texture index -> image index (texture.source().index(), loader.rs:75-92)
bool texture_source(const Doc& d, const JVal* info, uint32_t& image_index, std::vector<std::string>* warnings) {
    if (!info || info->type != JVal::Obj) return false;
    long long ti = info->integer("index", -1);
    const auto& tex = d.root.arr("textures");
    if (ti < 0 || (size_t)ti >= tex.size()) return false;
    long long src = tex[(size_t)ti].integer("source", -1);
    if (src < 0) return false;
    // the gltf crate rejects a document whose texture names an image that does not exist; here the slot is left empty (0xFFFF) with a warning
    if ((size_t)src >= d.root.arr("images").size()) {
        if (warnings) warnings->push_back("texture " + std::to_string(ti) + " refers to image " + std::to_string(src) + ", which does not exist: texture slot left empty");
        return false;
    }
    image_index = (uint32_t)src;
    return true;
}
} // namespace

bool load_gltf(const std::string& path, LoadedModel& out, std::string& err) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { err = "cannot read " + path; return false; }
    Doc d; d.base_dir = dir_of(path);
    std::vector<uint8_t> glb_bin; bool have_bin = false;
    const char* js = nullptr; size_t jn = 0;
    if (file.size() >= 12 && rd<uint32_t>(file.data()) == 0x46546C67u) {   // "glTF"
        uint32_t version = rd<uint32_t>(file.data() + 4), total = rd<uint32_t>(file.data() + 8);
        if (version != 2) { err = "GLB version " + std::to_string(version) + " is not supported"; return false; }
        if (total > file.size()) { err = "GLB is truncated"; return false; }
        size_t pos = 12;
        while (pos + 8 <= total) {
            uint32_t clen = rd<uint32_t>(file.data() + pos), ctype = rd<uint32_t>(file.data() + pos + 4);
            if ((size_t)clen > total - pos - 8) { err = "GLB chunk exceeds the file"; return false; }
            if (ctype == 0x4E4F534Au && !js) { js = reinterpret_cast<const char*>(file.data() + pos + 8); jn = clen; }
            else if (ctype == 0x004E4942u && !have_bin) { glb_bin.assign(file.begin() + (long)(pos + 8), file.begin() + (long)(pos + 8 + clen)); have_bin = true; }
            pos += 8 + (size_t)clen; pos = (pos + 3) & ~(size_t)3;
        }
        if (!js) { err = "GLB without a JSON chunk"; return false; }
    } else { js = reinterpret_cast<const char*>(file.data()); jn = file.size(); }
    {
        JParser p{js, js + jn, {}, 0};
        if (jn >= 3 && (uint8_t)js[0] == 0xEF && (uint8_t)js[1] == 0xBB && (uint8_t)js[2] == 0xBF) p.p += 3;
        if (!p.value(d.root) || d.root.type != JVal::Obj) { err = p.err.empty() ? "glTF JSON is not an object" : p.err; return false; }
    }
    // buffers
    const auto& jbufs = d.root.arr("buffers");
    for (size_t i = 0; i < jbufs.size(); ++i) {
        std::vector<uint8_t> b; std::string uri = jbufs[i].str("uri");
        if (uri.empty()) {
            if (i != 0 || !have_bin) { err = "buffer " + std::to_string(i) + " has no uri and there is no BIN chunk"; return false; }
            b = glb_bin;
        } else { std::string why; if (!load_uri(uri, d.base_dir, b, why)) { err = "buffer " + std::to_string(i) + ": " + why; return false; } }
        size_t want = (size_t)jbufs[i].integer("byteLength", 0);
        if (b.size() < want) { err = "buffer " + std::to_string(i) + " is shorter than its byteLength"; return false; }
        d.buffers.push_back(std::move(b));
    }
    out = LoadedModel();
    // 0. images (loader.rs:19-56)
    for (const JVal& img : d.root.arr("images")) {
        std::vector<uint8_t> rgba; uint32_t w = 0, h = 0; std::string why;
        std::vector<uint8_t> tex((size_t)kTexW * kTexH * 4, 255);
        if (decode_image(d, img, rgba, w, h, why)) resize_lanczos3_rgba8(rgba.data(), w, h, tex.data(), kTexW, kTexH);
        else out.warnings.push_back("image " + std::to_string(out.images.size()) + ": " + why + " -> white texture");
        out.images.push_back(std::move(tex));
    }
    // 1. materials (loader.rs:58-99)
    for (const JVal& jm : d.root.arr("materials")) {
        static const JVal empty_obj = [] { JVal v; v.type = JVal::Obj; return v; }();
        const JVal* pbrp = jm.get("pbrMetallicRoughness");
        const JVal& pbr = (pbrp && pbrp->type == JVal::Obj) ? *pbrp : empty_obj;
        float bc[4] = {1, 1, 1, 1};
        const auto& jbc = pbr.arr("baseColorFactor");
        for (size_t k = 0; k < 4 && k < jbc.size(); ++k) if (jbc[k].is_num()) bc[k] = (float)jbc[k].n;
        float metallic = (float)pbr.num("metallicFactor", 1.0), roughness = (float)pbr.num("roughnessFactor", 1.0);
        frt_material m = MaterialBuilder(bc[0], bc[1], bc[2], bc[3]).metallic(metallic).roughness(roughness);   // .metallic(x) sets metallic = 1 (sic, material.rs:54-58)
        m.tex_info_0 = pack16(pack16(m.tex_info_0, 0xFFFFFFFFu, false), 0xFFFFFFFFu, true);
        m.tex_info_1 = pack16(pack16(m.tex_info_1, 0xFFFFFFFFu, false), 0xFFFFFFFFu, true);
        m.tex_info_2 = pack16(m.tex_info_2, 0xFFFFFFFFu, false);
        uint32_t ii;
        if (texture_source(d, pbr.get("baseColorTexture"), ii, &out.warnings)) m.tex_info_0 = pack16(m.tex_info_0, ii, false);
        if (texture_source(d, jm.get("normalTexture"), ii, &out.warnings)) m.tex_info_0 = pack16(m.tex_info_0, ii, true);
        if (texture_source(d, jm.get("occlusionTexture"), ii, &out.warnings)) m.tex_info_1 = pack16(m.tex_info_1, ii, false);
        if (texture_source(d, jm.get("emissiveTexture"), ii, &out.warnings)) m.tex_info_1 = pack16(m.tex_info_1, ii, true);
        if (texture_source(d, pbr.get("metallicRoughnessTexture"), ii, &out.warnings)) m.tex_info_2 = pack16(m.tex_info_2, ii, false);
        const auto& je = jm.arr("emissiveFactor");
        for (size_t k = 0; k < 3; ++k) m.emissive_factor[k] = (k < je.size() && je[k].is_num()) ? (float)je[k].n : 0.0f;
        out.materials.push_back(m);
    }
    if (out.materials.empty()) out.materials.push_back(MaterialBuilder(1, 1, 1, 1));   // loader.rs:101-104
    // 2. meshes (loader.rs:106-178)
    for (const JVal& mesh : d.root.arr("meshes")) {
        for (const JVal& prim : mesh.arr("primitives")) {
            long long mode = prim.integer("mode", 4);
            if (mode != 4) { err = "primitive mode " + std::to_string(mode) + ": only TRIANGLES (4) is supported"; return false; }
            const JVal* attrs = prim.get("attributes");
            if (!attrs || attrs->type != JVal::Obj) { err = "primitive without attributes"; return false; }
            Geometry g; Acc a; std::vector<uint8_t> zeros; std::vector<float> f;
            long long ai = attrs->integer("POSITION", -1);
            if (ai < 0) { err = "primitive without POSITION"; return false; }
            if (!get_accessor(d, ai, a, zeros, err)) return false;
            if (a.comp != 5126) { err = "POSITION must be float"; return false; }
            if (!read_floats(a, 3, false, f, err, "POSITION")) return false;
            const size_t nv = a.count;
            if (nv == 0) { err = "primitive with zero vertices"; return false; }
            g.positions.resize(nv * 4);
            for (size_t i = 0; i < nv; ++i) { g.positions[i * 4] = f[i * 3]; g.positions[i * 4 + 1] = f[i * 3 + 1]; g.positions[i * 4 + 2] = f[i * 3 + 2]; g.positions[i * 4 + 3] = 1.0f; }
            std::vector<float> normals, uvs, tangents;
            if ((ai = attrs->integer("NORMAL", -1)) >= 0) {
                if (!get_accessor(d, ai, a, zeros, err) || !read_floats(a, 3, false, normals, err, "NORMAL")) return false;
            } else { normals.resize(nv * 3); for (size_t i = 0; i < nv; ++i) { normals[i * 3] = 0; normals[i * 3 + 1] = 1; normals[i * 3 + 2] = 0; } }   // loader.rs:127-129
            if (normals.size() != nv * 3) { err = "NORMAL count differs from POSITION count"; return false; }
            if ((ai = attrs->integer("TEXCOORD_0", -1)) >= 0) {
                if (!get_accessor(d, ai, a, zeros, err) || !read_floats(a, 2, true, uvs, err, "TEXCOORD_0")) return false;
            }
            if ((ai = attrs->integer("TANGENT", -1)) >= 0) {
                if (!get_accessor(d, ai, a, zeros, err) || !read_floats(a, 4, false, tangents, err, "TANGENT")) return false;
            }
            g.attributes.resize(nv);
            for (size_t i = 0; i < nv; ++i) {
                frt_vertex_attr va;
                geometry::encode_octahedral_normal(&normals[i * 3], va.normal);
                bool hu = (i + 1) * 2 <= uvs.size(), ht = (i + 1) * 4 <= tangents.size();
                va.uv[0] = hu ? uvs[i * 2] : 0.0f; va.uv[1] = hu ? uvs[i * 2 + 1] : 0.0f;
                va.tangent[0] = ht ? tangents[i * 4] : 1.0f; va.tangent[1] = ht ? tangents[i * 4 + 1] : 0.0f;
                va.tangent[2] = ht ? tangents[i * 4 + 2] : 0.0f; va.tangent[3] = ht ? tangents[i * 4 + 3] : 1.0f;
                g.attributes[i] = va;
            }
            long long ii = prim.integer("indices", -1);
            if (ii >= 0) {
                if (!get_accessor(d, ii, a, zeros, err)) return false;
                if (a.ncomp != 1 || (a.comp != 5121 && a.comp != 5123 && a.comp != 5125)) { err = "indices must be unsigned scalar"; return false; }
                g.indices.resize(a.count);
                for (size_t i = 0; i < a.count; ++i) {
                    const uint8_t* p = a.base + i * a.stride;
                    g.indices[i] = a.comp == 5121 ? (uint32_t)p[0] : a.comp == 5123 ? (uint32_t)rd<uint16_t>(p) : rd<uint32_t>(p);
                }
            } else { g.indices.resize(nv); for (size_t i = 0; i < nv; ++i) g.indices[i] = (uint32_t)i; }   // loader.rs:160-163
            if (g.indices.empty() || g.indices.size() % 3 != 0) { err = "index count is not a positive multiple of 3"; return false; }
            for (uint32_t v : g.indices) if (v >= nv) { err = "index out of range"; return false; }
            out.geometries.push_back(std::move(g));
            long long mi = prim.integer("material", 0);
            out.material_indices.push_back(mi < 0 ? 0u : (uint32_t)mi);   // primitive.material().index().unwrap_or(0)
        }
    }
    return true;
}

// ================================================================================================ OBJ (extension)
bool load_obj(const std::string& path, LoadedModel& out, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "cannot read " + path; return false; }
    std::vector<float> P, T, N;
    struct Corner { long long v, t, n; };
    std::vector<Corner> corners;   // 3 per triangle
    std::string line; size_t lineno = 0; bool warned_mtl = false;
    out = LoadedModel();
    auto resolve = [](long long i, size_t n) -> long long { return i > 0 ? i - 1 : (i < 0 ? (long long)n + i : -1); };
    while (std::getline(f, line)) {
        ++lineno;
        std::istringstream ss(line); std::string tag; ss >> tag;
        if (tag == "v") { float x = 0, y = 0, z = 0; ss >> x >> y >> z; P.insert(P.end(), {x, y, z}); }
        else if (tag == "vt") { float u = 0, v = 0; ss >> u >> v; T.insert(T.end(), {u, v}); }
        else if (tag == "vn") { float x = 0, y = 0, z = 0; ss >> x >> y >> z; N.insert(N.end(), {x, y, z}); }
        else if (tag == "f") {
            std::vector<Corner> poly; std::string tok;
            while (ss >> tok) {
                Corner c{-1, -1, -1}; long long vals[3] = {0, 0, 0}; int k = 0; size_t s = 0;
                for (size_t i = 0; i <= tok.size() && k < 3; ++i) if (i == tok.size() || tok[i] == '/') { if (i > s) vals[k] = atoll(tok.substr(s, i - s).c_str()); ++k; s = i + 1; }
                c.v = resolve(vals[0], P.size() / 3); c.t = resolve(vals[1], T.size() / 2); c.n = resolve(vals[2], N.size() / 3);
                if (c.v < 0 || (size_t)c.v >= P.size() / 3) { err = "OBJ line " + std::to_string(lineno) + ": vertex index out of range"; return false; }
                if (c.t >= (long long)(T.size() / 2)) c.t = -1;
                if (c.n >= (long long)(N.size() / 3)) c.n = -1;
                poly.push_back(c);
            }
            for (size_t i = 2; i < poly.size(); ++i) { corners.push_back(poly[0]); corners.push_back(poly[i - 1]); corners.push_back(poly[i]); }
        } else if ((tag == "usemtl" || tag == "mtllib") && !warned_mtl) { out.warnings.push_back("OBJ materials are ignored: one default material"); warned_mtl = true; }
    }
    if (corners.empty()) { err = "OBJ without faces"; return false; }
    // smooth normals for corners that name none: area-weighted sum of the face normals around each position
    std::vector<float> SN;
    bool need_sn = false;
    for (const Corner& c : corners) if (c.n < 0) { need_sn = true; break; }
    if (need_sn) {
        SN.assign(P.size(), 0.0f);
        for (size_t t = 0; t + 2 < corners.size(); t += 3) {
            const float* a = &P[(size_t)corners[t].v * 3]; const float* b = &P[(size_t)corners[t + 1].v * 3]; const float* c = &P[(size_t)corners[t + 2].v * 3];
            float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) SN[(size_t)corners[t + k].v * 3 + j] += n[j];
        }
    }
    Geometry g;
    std::map<std::tuple<long long, long long, long long>, uint32_t> seen;
    for (const Corner& c : corners) {
        auto key = std::make_tuple(c.v, c.t, c.n);
        auto it = seen.find(key);
        if (it == seen.end()) {
            uint32_t id = (uint32_t)g.attributes.size();
            seen.emplace(key, id);
            const float* p = &P[(size_t)c.v * 3];
            g.positions.insert(g.positions.end(), {p[0], p[1], p[2], 1.0f});
            float n[3] = {0, 1, 0};
            const float* src = c.n >= 0 ? &N[(size_t)c.n * 3] : &SN[(size_t)c.v * 3];
            float len = sqrtf(src[0] * src[0] + src[1] * src[1] + src[2] * src[2]);
            if (len > 0.0f) { n[0] = src[0] / len; n[1] = src[1] / len; n[2] = src[2] / len; }
            frt_vertex_attr va;
            geometry::encode_octahedral_normal(n, va.normal);
            va.uv[0] = c.t >= 0 ? T[(size_t)c.t * 2] : 0.0f; va.uv[1] = c.t >= 0 ? 1.0f - T[(size_t)c.t * 2 + 1] : 0.0f;   // OBJ v runs upward
            va.tangent[0] = 1.0f; va.tangent[1] = 0.0f; va.tangent[2] = 0.0f; va.tangent[3] = 1.0f;
            g.attributes.push_back(va);
            g.indices.push_back(id);
        } else g.indices.push_back(it->second);
    }
    out.geometries.push_back(std::move(g));
    out.materials.push_back(MaterialBuilder(1, 1, 1, 1));
    out.material_indices.push_back(0);
    return true;
}

bool load_model(const std::string& path, LoadedModel& out, std::string& err) {
    size_t k = path.find_last_of('.');
    std::string ext = k == std::string::npos ? "" : path.substr(k + 1);
    for (char& c : ext) c = (char)tolower((unsigned char)c);
    if (ext == "obj") return load_obj(path, out, err);
    return load_gltf(path, out, err);     // .gltf, .glb, .vrm (scenes.rs:354) …
}

// ================================================================================================ builder.rs:191-314
std::vector<uint32_t> add_gltf_materials(SceneBuilder& b, const LoadedModel& m) {
    const size_t ni = m.images.size();
    std::vector<int64_t> color_map(ni, -1), data_map(ni, -1);
    std::vector<uint32_t> ids;
    auto color = [&](uint32_t img) -> uint32_t { if (color_map[img] < 0) color_map[img] = b.add_color_texture(m.images[img].data()); return (uint32_t)color_map[img]; };
    auto data = [&](uint32_t img) -> uint32_t { if (data_map[img] < 0) data_map[img] = b.add_data_texture(m.images[img].data()); return (uint32_t)data_map[img]; };
    for (frt_material mat : m.materials) {
        // order as in the reference: base colour (colour array), normal, occlusion (data array), emissive (colour), metallic-roughness (data)
        uint32_t t = mat.tex_info_0 & 0xFFFFu;
        if (t != 0xFFFFu && t < ni) mat.tex_info_0 = pack16(mat.tex_info_0, color(t), false);
        t = mat.tex_info_0 >> 16;
        if (t != 0xFFFFu && t < ni) mat.tex_info_0 = pack16(mat.tex_info_0, data(t), true);
        t = mat.tex_info_1 & 0xFFFFu;
        if (t != 0xFFFFu && t < ni) mat.tex_info_1 = pack16(mat.tex_info_1, data(t), false);
        t = mat.tex_info_1 >> 16;
        if (t != 0xFFFFu && t < ni) mat.tex_info_1 = pack16(mat.tex_info_1, color(t), true);
        t = mat.tex_info_2 & 0xFFFFu;
        if (t != 0xFFFFu && t < ni) mat.tex_info_2 = pack16(mat.tex_info_2, data(t), false);
        ids.push_back(b.add_material(mat));
    }
    return ids;
}
std::vector<uint32_t> add_gltf_meshes(SceneBuilder& b, const LoadedModel& m) {
    std::vector<uint32_t> ids;
    for (const Geometry& g : m.geometries) ids.push_back(b.add_mesh(g));
    return ids;
}
void add_gltf_instances(SceneBuilder& b, const std::vector<uint32_t>& mesh_ids, const std::vector<uint32_t>& mat_ids,
                        const std::vector<uint32_t>& material_indices, const Mat4& transform) {
    for (size_t i = 0; i < mesh_ids.size(); ++i) {
        uint32_t mat_index = i < material_indices.size() ? material_indices[i] : 0u;    // builder.rs:294-300
        uint32_t mat_id = mat_index < mat_ids.size() ? mat_ids[mat_index] : 0u;         // :302-307
        b.add_instance(mesh_ids[i], mat_id, transform);
    }
}

namespace scenes {
bool create_gltf_scene(SceneBuilder& b, const std::string& path, const Mat4& model_transform, const Mat4& light_transform, std::string& err) {
    LoadedModel m;
    if (!load_model(path, m, err)) return false;
    uint32_t plane_id = b.add_mesh(geometry::create_plane());
    uint32_t light_mesh_id = b.add_mesh(geometry::create_plane());
    uint32_t mat_floor = b.add_material(MaterialBuilder(0.73f, 0.73f, 0.73f, 1.0f).roughness(0.99f));
    b.add_instance(plane_id, mat_floor, mat4_mul(mat4_translation(0.0f, -1.0f, 0.0f), mat4_scale(10.0f, 10.0f, 10.0f)));
    const float white[3] = {1.0f, 1.0f, 1.0f};
    b.register_quad_light(light_mesh_id, light_transform, white, 15.0f);
    std::vector<uint32_t> mat_ids = add_gltf_materials(b, m);
    std::vector<uint32_t> mesh_ids = add_gltf_meshes(b, m);
    add_gltf_instances(b, mesh_ids, mat_ids, m.material_indices, model_transform);
    return true;
}
}

} // namespace frt

// ================================================================================================ C ABI (include/frt.h)
using namespace frt;
namespace frt { int set_error(int code, const std::string& msg); }

extern "C" {

frt_model* frt_model_load(const char* path) {
    if (!path) { set_error(FRT_ERR_INVALID_ARG, "model_load: null path"); return nullptr; }
    frt_model* m = new frt_model();
    std::string err;
    if (!load_model(path, m->m, err)) { set_error(FRT_ERR_INVALID_ARG, std::string("model_load: ") + err); delete m; return nullptr; }
    return m;
}
void frt_model_destroy(frt_model* m) { delete m; }

int frt_model_counts(const frt_model* m, uint32_t out[4]) {
    if (!m || !out) return set_error(FRT_ERR_INVALID_ARG, "model_counts: null");
    out[0] = (uint32_t)m->m.geometries.size(); out[1] = (uint32_t)m->m.materials.size(); out[2] = (uint32_t)m->m.images.size(); out[3] = (uint32_t)m->m.warnings.size();
    return FRT_OK;
}
int frt_model_geometry_counts(const frt_model* m, uint32_t geo, uint32_t* nverts, uint32_t* nidx, uint32_t* material_index) {
    if (!m || geo >= m->m.geometries.size()) return set_error(FRT_ERR_INVALID_ARG, "model_geometry_counts: bad arguments");
    if (nverts) *nverts = (uint32_t)m->m.geometries[geo].attributes.size();
    if (nidx) *nidx = (uint32_t)m->m.geometries[geo].indices.size();
    if (material_index) *material_index = m->m.material_indices[geo];
    return FRT_OK;
}
int frt_model_geometry_get(const frt_model* m, uint32_t geo, float* pos4, frt_vertex_attr* attrs, uint32_t* idx) {
    if (!m || geo >= m->m.geometries.size()) return set_error(FRT_ERR_INVALID_ARG, "model_geometry_get: bad arguments");
    const Geometry& g = m->m.geometries[geo];
    if (pos4) memcpy(pos4, g.positions.data(), g.positions.size() * sizeof(float));
    if (attrs) memcpy(attrs, g.attributes.data(), g.attributes.size() * sizeof(frt_vertex_attr));
    if (idx) memcpy(idx, g.indices.data(), g.indices.size() * sizeof(uint32_t));
    return FRT_OK;
}
int frt_model_material_get(const frt_model* m, uint32_t i, frt_material* out) {
    if (!m || !out || i >= m->m.materials.size()) return set_error(FRT_ERR_INVALID_ARG, "model_material_get: bad arguments");
    *out = m->m.materials[i];
    return FRT_OK;
}
int frt_model_material_set(frt_model* m, uint32_t i, const frt_material* in) {
    if (!m || !in || i >= m->m.materials.size()) return set_error(FRT_ERR_INVALID_ARG, "model_material_set: bad arguments");
    m->m.materials[i] = *in;
    return FRT_OK;
}
int frt_model_image_get(const frt_model* m, uint32_t i, uint8_t* rgba8) {
    if (!m || !rgba8 || i >= m->m.images.size()) return set_error(FRT_ERR_INVALID_ARG, "model_image_get: bad arguments");
    memcpy(rgba8, m->m.images[i].data(), m->m.images[i].size());
    return FRT_OK;
}
const char* frt_model_warning(const frt_model* m, uint32_t i) {
    if (!m || i >= m->m.warnings.size()) return nullptr;
    return m->m.warnings[i].c_str();
}

int frt_scene_add_gltf_materials(frt_scene* s, const frt_model* m, uint32_t* mat_ids) {
    if (!s || !m) return set_error(FRT_ERR_INVALID_ARG, "add_gltf_materials: null");
    if (s->b.materials.size() + m->m.materials.size() > 0xFFFFu) return set_error(FRT_ERR_LIMIT, "more than 65535 materials");
    if (s->b.color_textures.size() + m->m.images.size() >= 0xFFFFu || s->b.data_textures.size() + m->m.images.size() >= 0xFFFFu)
        return set_error(FRT_ERR_LIMIT, "too many texture layers");
    std::vector<uint32_t> ids = add_gltf_materials(s->b, m->m);
    s->b.built = false;
    if (mat_ids) memcpy(mat_ids, ids.data(), ids.size() * sizeof(uint32_t));
    return (int)ids.size();
}
int frt_scene_add_gltf_meshes(frt_scene* s, const frt_model* m, uint32_t* mesh_ids) {
    if (!s || !m) return set_error(FRT_ERR_INVALID_ARG, "add_gltf_meshes: null");
    std::vector<uint32_t> ids = add_gltf_meshes(s->b, m->m);
    s->b.built = false;
    if (mesh_ids) memcpy(mesh_ids, ids.data(), ids.size() * sizeof(uint32_t));
    return (int)ids.size();
}
int frt_scene_add_gltf_instances(frt_scene* s, const frt_model* m, const uint32_t* mesh_ids, uint32_t n_mesh, const uint32_t* mat_ids, uint32_t n_mat, const float transform[16]) {
    if (!s || !m || !transform || (n_mesh && !mesh_ids) || (n_mat && !mat_ids)) return set_error(FRT_ERR_INVALID_ARG, "add_gltf_instances: null");
    for (uint32_t i = 0; i < n_mesh; ++i) if (mesh_ids[i] >= s->b.mesh_infos.size()) return set_error(FRT_ERR_INVALID_ARG, "add_gltf_instances: unknown mesh id");
    for (uint32_t i = 0; i < n_mat; ++i) if (mat_ids[i] >= s->b.materials.size()) return set_error(FRT_ERR_INVALID_ARG, "add_gltf_instances: unknown material id");
    if (s->b.materials.empty()) return set_error(FRT_ERR_STATE, "add_gltf_instances: the scene has no material 0 to fall back to");
    Mat4 t; memcpy(t.m, transform, sizeof t.m);
    add_gltf_instances(s->b, std::vector<uint32_t>(mesh_ids, mesh_ids + n_mesh), std::vector<uint32_t>(mat_ids, mat_ids + n_mat), m->m.material_indices, t);
    s->b.built = false;
    return FRT_OK;
}
frt_scene* frt_scene_create_gltf_scene(const char* path, const float model_transform[16], const float light_transform[16]) {
    if (!path || !model_transform || !light_transform) { set_error(FRT_ERR_INVALID_ARG, "create_gltf_scene: null"); return nullptr; }
    frt_scene* s = new frt_scene();
    Mat4 mt, lt; memcpy(mt.m, model_transform, sizeof mt.m); memcpy(lt.m, light_transform, sizeof lt.m);
    std::string err;
    // scenes.rs:311-314 logs a load failure and goes on to build an EMPTY scene (every ray misses). This builder refuses scenes
    // without triangles, so the failure is reported instead: NULL + the loader's message.
    if (!scenes::create_gltf_scene(s->b, path, mt, lt, err)) { set_error(FRT_ERR_INVALID_ARG, std::string("create_gltf_scene: ") + err); delete s; return nullptr; }
    s->b.build();
    if (!s->b.built) { set_error(FRT_ERR_LIMIT, "create_gltf_scene: build: " + s->b.error); delete s; return nullptr; }
    return s;
}

} // extern "C"
