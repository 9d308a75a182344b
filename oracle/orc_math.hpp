// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Scalar CPU restatement of the numeric environment of the reference shaders
// (WGSL builtins used by src/shaders/{gbuffer,restir,restir_spatial,post}.wgsl).
// Parity status: PARITY UNPINNED — the reference ships no tests / golden vectors
// and cannot be built here (no rustc, no Vulkan ray-query device); WGSL leaves the
// precision of sin/cos/pow/exp/normalize implementation-defined. This file therefore
// FIXES one definition for each builtin ("numeric contract"):
// IEEE-754 binary32, round-to-nearest-even, no FMA contraction, operations in the
// order written below. The HIP product implements the same contract independently
// (fast-raytracing-wgpu_amd/csrc/frt_math.hpp); nothing in the product includes this file.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

// TEXT MODE (tests only): evaluate the WGSL as written, with the C library standing in for the GPU's builtins — true division wherever
// the shader divides, pow(x, y) = exp2(y * log2(x)) (NaN for a negative base, like WGSL), normalize = v * inverseSqrt(dot(v, v)), libm
// sin / cos / exp / log2 — instead of the numeric contract below (reciprocal-multiply, pow by repeated multiplication, polynomials),
// which was chosen so that CPU and GPU agree bit for bit and, in places, because it is cheaper on the GPU. A path tracer is chaotic per
// pixel, so the two modes cannot be compared pixel by pixel; tests/test_oracle_text_mode.py compares the accumulated images
// statistically and so bounds what the contract costs relative to a literal reading of the reference. Set before rendering, never during.
inline bool& text_mode() { static bool on = false; return on; }

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct ivec2 { int x, y; };

inline vec2 V2(float x, float y) { return {x, y}; }
inline vec3 V3(float x, float y, float z) { return {x, y, z}; }
inline vec3 V3(float s) { return {s, s, s}; }
inline vec4 V4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline vec4 V4(vec3 v, float w) { return {v.x, v.y, v.z, w}; }
inline vec3 xyz(vec4 v) { return {v.x, v.y, v.z}; }

inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
inline vec2 operator/(vec2 a, vec2 b) { return {a.x / b.x, a.y / b.y}; }

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
// vector / scalar: one IEEE reciprocal, then three multiplies (contract; WGSL allows 2.5 ulp for division)
inline vec3 operator/(vec3 a, float s) {
    if (text_mode()) return {a.x / s, a.y / s, a.z / s};
    float r = 1.0f / s; return {a.x * r, a.y * r, a.z * r};
}
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(float s, vec3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }

inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

// --- scalar builtins ---
// min/max: IEEE-754 minNum/maxNum (a NaN operand is dropped), which is what GPU min/max instructions do
// and what keeps the reference's NaN weights (GGX D = inf on the roughness-0.01 box) from spreading through
// clamp(): clamp(NaN, 0, 20) = 0. The sign of a zero result is unspecified; outputs are compared numerically.
inline float fmin_(float a, float b) { return fminf(a, b); }
inline float fmax_(float a, float b) { return fmaxf(a, b); }
inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
inline float sign_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float fract_(float x) { return x - floorf(x); }
inline float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float smoothstep_(float e0, float e1, float x) {
    float t = clamp_((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
inline float inversesqrt_(float x) { return 1.0f / sqrtf(x); }

inline vec3 max3(vec3 a, vec3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline vec3 clamp3(vec3 v, vec3 lo, vec3 hi) {
    return {clamp_(v.x, lo.x, hi.x), clamp_(v.y, lo.y, hi.y), clamp_(v.z, lo.z, hi.z)};
}
inline vec3 mix3(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix3v(vec3 a, vec3 b, vec3 t) { return a * (1.0f - t) + b * t; }

inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot2(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(vec3 v) { return sqrtf(dot(v, v)); }
inline float length2(vec2 v) { return sqrtf(dot2(v, v)); }
// normalize: v * (1 / length(v)) — one IEEE division, then three multiplies (contract; WGSL leaves it open)
inline vec3 normalize(vec3 v) {
    if (text_mode()) return v * inversesqrt_(dot(v, v));
    float r = 1.0f / length(v); return v * r;
}
inline float distance(vec3 a, vec3 b) { return length(a - b); }
inline vec3 reflect(vec3 i, vec3 n) { return i - n * (2.0f * dot(n, i)); }
inline vec3 refract(vec3 i, vec3 n, float eta) {
    float ndi = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f) return V3(0.0f);
    return i * eta - n * (eta * ndi + sqrtf(k));
}

// column-major 4x4, as in glam / WGSL mat4x4f
struct mat4 { vec4 c[4]; };
inline vec4 mul(const mat4& m, vec4 v) {
    return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w;
}
inline mat4 mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int j = 0; j < 4; ++j) r.c[j] = mul(a, b.c[j]);
    return r;
}

// --- elementary functions: fixed algorithms (contract) ---
inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// sin/cos: Cody-Waite reduction by pi/2 (3 constants), Cephes sinf/cosf minimax
// polynomials on [-pi/4, pi/4]. Valid for |x| < ~1e5; the shaders only pass [0, 2*pi].
inline void sincos_(float x, float* s, float* c) {
    if (text_mode()) { *s = sinf(x); *c = cosf(x); return; }
    float q = floorf(x * 0.636619772f + 0.5f);
    float r = x - q * 1.5703125f;
    r = r - q * 4.837512969970703125e-4f;
    r = r - q * 7.54978995489188e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    int n = (int)q & 3;
    float sv = (n & 1) ? cp : sp;
    float cv = (n & 1) ? sp : cp;
    if (n & 2) sv = -sv;
    if ((n + 1) & 2) cv = -cv;
    *s = sv; *c = cv;
}
inline float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
inline float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }

// exp2: n = floor(x + 0.5), f = x - n in [-0.5, 0.5], Cephes exp2f polynomial, scale by 2^n.
inline float exp2_(float x) {
    if (text_mode()) return exp2f(x);
    if (x >= 128.0f) return bits2f(0x7f800000u);
    if (!(x >= -126.0f)) return 0.0f;   // also NaN -> 0
    float n = floorf(x + 0.5f);
    float f = x - n;
    float p = 1.535336188319500e-4f;
    p = p * f + 1.339887440266574e-3f;
    p = p * f + 9.618437357674640e-3f;
    p = p * f + 5.550332471162809e-2f;
    p = p * f + 2.402264791363012e-1f;
    p = p * f + 6.931472028550421e-1f;
    p = p * f + 1.0f;
    int e = (int)n;
    if (e < -126) e = -126;   // x >= -126 so only n == -126 possible with f >= 0
    if (e > 127) { p = p * 2.0f; e = 127; }
    return p * bits2f((uint32_t)(e + 127) << 23);
}

// log2 for x > 0 (normal or denormal): x = m * 2^e, m in [sqrt(1/2), sqrt(2)), Cephes log2f polynomial.
inline float log2_(float x) {
    if (text_mode()) return log2f(x);
    uint32_t u = f2bits(x);
    int e = 0;
    if ((u & 0x7f800000u) == 0) { x = x * 8388608.0f; u = f2bits(x); e = -23; }
    e += (int)((u >> 23) & 0xff) - 126;              // x = m * 2^e with m in [0.5, 1)
    float m = bits2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = y * m - 1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m - 1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m - 1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m - 2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    y = y - 0.5f * z;      // ln(1+m) ~= m + y
    // log2(1+m) = (m + y) * log2(e), split log2(e) = 1 + 0.44269504088896340735992
    float r = y * 0.44269504088896340735992f;
    r = r + m * 0.44269504088896340735992f;
    r = r + y;
    r = r + m;
    return r + (float)e;
}

// pow(x, 5) / pow(x, 20) of the shaders by repeated multiplication (contract), x >= 0 at every call site
inline float pow5_(float x) { if (text_mode()) return exp2f(5.0f * log2f(x)); float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
inline float pow20_(float x) {
    if (text_mode()) return exp2f(20.0f * log2f(x));
    float x2 = x * x; float x4 = x2 * x2; float x8 = x4 * x4; float x16 = x8 * x8; return x16 * x4;
}
// pow(x, y) for the shader uses (y > 0): x <= 0 (or NaN) -> 0.
inline float pow_(float x, float y) {
    if (text_mode()) return exp2f(y * log2f(x));
    if (!(x > 0.0f)) return 0.0f;
    return exp2_(y * log2_(x));
}
inline float exp_(float x) { if (text_mode()) return expf(x); return exp2_(x * 1.44269504088896340736f); }

inline vec3 pow3(vec3 v, float y) { return {pow_(v.x, y), pow_(v.y, y), pow_(v.z, y)}; }

// --- storage-format conversions (F8 in SURVEY.md §0) ---
// f32 -> f16 bits, round-to-nearest-even, IEEE (overflow -> inf, denormals kept).
inline uint16_t f32_to_f16(float f) {
    uint32_t x = f2bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) {                       // inf / NaN
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x0200u | ((ax >> 13) & 0x3ffu)) : 0u));
    }
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);   // >= 65520 rounds to inf
    if (ax < 0x33000001u) return (uint16_t)sign;                // <= 2^-25 rounds to 0
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }          // denormal half
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x007fffffu; }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q += 1;
    return (uint16_t)(sign | (base + q));
}
inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return bits2f(sign);
        float v = (float)m * 5.9604644775390625e-8f;   // m * 2^-24, exact
        return (sign ? -v : v);
    }
    if (e == 31) return bits2f(sign | 0x7f800000u | (m << 13));
    return bits2f(sign | ((e + 112) << 23) | (m << 13));
}
// f32 -> unorm8: clamp to [0,1], scale, round half up (Vulkan permits either tie rule; fixed here).
inline uint8_t f32_to_unorm8(float v) {
    float c = clamp_(v, 0.0f, 1.0f);   // NaN -> 0 (maxNum)
    return (uint8_t)(int)floorf(c * 255.0f + 0.5f);
}
inline float unorm8_to_f32(uint8_t b) { return (float)b / 255.0f; }

} // namespace orc
