// frt_math.hpp — numeric contract of the HIP path tracer.
//
// The reference shaders (src/shaders/*.wgsl) run on implementation-defined WGSL builtins; this build fixes
// one definition per builtin so that results are reproducible bit-for-bit across launches, GPUs and the
// CPU checker: IEEE-754 binary32, round-to-nearest-even, no FMA contraction in contract code (the whole
// library is compiled with -ffp-contract=off; fmaf is only used, explicitly, in ray/box tests, whose results
// never reach an output), correctly rounded division and sqrt (-fhip-fp32-correctly-rounded-divide-sqrt),
// min/max = minNum/maxNum (v_min_f32 / v_max_f32), polynomial sin/cos/exp2/log2 built from +,-,*.
// Everything here is __host__ __device__ so tests can instantiate the device functions on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FRT_HD __host__ __device__ __forceinline__

namespace frt {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

FRT_HD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
FRT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
FRT_HD f3 splat3(float s) { return mk3(s, s, s); }
FRT_HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
FRT_HD f4 mk4(f3 v, float w) { return mk4(v.x, v.y, v.z, w); }
FRT_HD f3 xyz(f4 v) { return mk3(v.x, v.y, v.z); }

FRT_HD f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
FRT_HD f2 operator-(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
FRT_HD f2 operator*(f2 a, float s) { return mk2(a.x * s, a.y * s); }
FRT_HD f2 operator*(f2 a, f2 b) { return mk2(a.x * b.x, a.y * b.y); }
FRT_HD f2 operator/(f2 a, f2 b) { return mk2(a.x / b.x, a.y / b.y); }

FRT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
FRT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
FRT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
FRT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
FRT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
FRT_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
// vector / scalar: one IEEE reciprocal, then three multiplies (contract; WGSL allows 2.5 ulp for division)
FRT_HD f3 operator/(f3 a, float s) { float r = 1.0f / s; return mk3(a.x * r, a.y * r, a.z * r); }
FRT_HD f3 operator-(float s, f3 a) { return mk3(s - a.x, s - a.y, s - a.z); }
FRT_HD f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
FRT_HD f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }

// minNum / maxNum
FRT_HD float fminn(float a, float b) { return __builtin_fminf(a, b); }
FRT_HD float fmaxn(float a, float b) { return __builtin_fmaxf(a, b); }
FRT_HD float clampf(float x, float lo, float hi) { return fminn(fmaxn(x, lo), hi); }
FRT_HD float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
FRT_HD float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
FRT_HD float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
FRT_HD float sqrtf_(float x) { return __builtin_sqrtf(x); }
FRT_HD float floorf_(float x) { return __builtin_floorf(x); }
FRT_HD float fabsf_(float x) { return __builtin_fabsf(x); }
FRT_HD float rsqrt_exact(float x) { return 1.0f / sqrtf_(x); }

FRT_HD f3 max3(f3 a, f3 b) { return mk3(fmaxn(a.x, b.x), fmaxn(a.y, b.y), fmaxn(a.z, b.z)); }
FRT_HD f3 clamp3(f3 v, f3 lo, f3 hi) { return mk3(clampf(v.x, lo.x, hi.x), clampf(v.y, lo.y, hi.y), clampf(v.z, lo.z, hi.z)); }
FRT_HD f3 mix3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }

FRT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FRT_HD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
FRT_HD float length(f3 v) { return sqrtf_(dot(v, v)); }
FRT_HD float length2(f2 v) { return sqrtf_(v.x * v.x + v.y * v.y); }
FRT_HD f3 normalize(f3 v) { float r = 1.0f / length(v); return v * r; }
FRT_HD float distance(f3 a, f3 b) { return length(a - b); }
FRT_HD f3 reflect(f3 i, f3 n) { return i - n * (2.0f * dot(n, i)); }
FRT_HD f3 refract(f3 i, f3 n, float eta) {
    float ndi = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f) return splat3(0.0f);
    return i * eta - n * (eta * ndi + sqrtf_(k));
}

struct m4 { f4 c[4]; };   // column-major
FRT_HD f4 mul(const m4& m, f4 v) { return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w; }
FRT_HD m4 mul(const m4& a, const m4& b) { m4 r; for (int j = 0; j < 4; ++j) r.c[j] = mul(a, b.c[j]); return r; }
FRT_HD m4 load_m4(const float* p) { m4 m; for (int k = 0; k < 4; ++k) m.c[k] = mk4(p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]); return m; }

FRT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
FRT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// sin/cos: Cody–Waite pi/2 reduction + Cephes minimax polynomials (same algorithm as the checker; contract)
FRT_HD void sincosf_(float x, float& s, float& c) {
    float q = floorf_(x * 0.636619772f + 0.5f);
    float r = x - q * 1.5703125f;
    r = r - q * 4.837512969970703125e-4f;
    r = r - q * 7.54978995489188e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    int n = (int)q & 3;
    float sv = (n & 1) ? cp : sp;
    float cv = (n & 1) ? sp : cp;
    s = (n & 2) ? -sv : sv;
    c = ((n + 1) & 2) ? -cv : cv;
}

FRT_HD float exp2f_(float x) {
    if (x >= 128.0f) return u2f(0x7f800000u);
    if (!(x >= -126.0f)) return 0.0f;
    float n = floorf_(x + 0.5f);
    float f = x - n;
    float p = 1.535336188319500e-4f;
    p = p * f + 1.339887440266574e-3f;
    p = p * f + 9.618437357674640e-3f;
    p = p * f + 5.550332471162809e-2f;
    p = p * f + 2.402264791363012e-1f;
    p = p * f + 6.931472028550421e-1f;
    p = p * f + 1.0f;
    int e = (int)n;
    if (e < -126) e = -126;
    if (e > 127) { p = p * 2.0f; e = 127; }
    return p * u2f((uint32_t)(e + 127) << 23);
}
FRT_HD float log2f_(float x) {
    uint32_t u = f2u(x);
    int e = 0;
    if ((u & 0x7f800000u) == 0) { x = x * 8388608.0f; u = f2u(x); e = -23; }
    e += (int)((u >> 23) & 0xff) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
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
    y = y - 0.5f * z;
    float r = y * 0.44269504088896340735992f;
    r = r + m * 0.44269504088896340735992f;
    r = r + y;
    r = r + m;
    return r + (float)e;
}
// pow(x, 5) and pow(x, 20) of the shaders (Schlick terms restir.wgsl:171, :179; normal weight post.wgsl:125) by repeated
// multiplication, x >= 0 at every call site
FRT_HD float pow5_(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
FRT_HD float pow20_(float x) { float x2 = x * x; float x4 = x2 * x2; float x8 = x4 * x4; float x16 = x8 * x8; return x16 * x4; }
FRT_HD float powf_(float x, float y) { return (x > 0.0f) ? exp2f_(y * log2f_(x)) : 0.0f; }
FRT_HD float expf_(float x) { return exp2f_(x * 1.44269504088896340736f); }

// storage formats (src/renderer.rs:73, :91, :129-131)
FRT_HD uint16_t f32_to_f16_bits(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
FRT_HD float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
FRT_HD uint32_t f32_to_unorm8(float v) { return (uint32_t)(int)floorf_(clampf(v, 0.0f, 1.0f) * 255.0f + 0.5f); }
FRT_HD float unorm8_to_f32(uint32_t b) { return (float)b / 255.0f; }
FRT_HD uint32_t pack_rgba8(f4 c) {
    return f32_to_unorm8(c.x) | (f32_to_unorm8(c.y) << 8) | (f32_to_unorm8(c.z) << 16) | (f32_to_unorm8(c.w) << 24);
}
FRT_HD f4 unpack_rgba8(uint32_t p) {
    return mk4(unorm8_to_f32(p & 0xffu), unorm8_to_f32((p >> 8) & 0xffu), unorm8_to_f32((p >> 16) & 0xffu), unorm8_to_f32(p >> 24));
}
FRT_HD uint2 pack_rgba16f(f4 c) {
    uint2 r;
    r.x = (uint32_t)f32_to_f16_bits(c.x) | ((uint32_t)f32_to_f16_bits(c.y) << 16);
    r.y = (uint32_t)f32_to_f16_bits(c.z) | ((uint32_t)f32_to_f16_bits(c.w) << 16);
    return r;
}
FRT_HD f4 unpack_rgba16f(uint2 p) {
    return mk4(f16_bits_to_f32((uint16_t)(p.x & 0xffffu)), f16_bits_to_f32((uint16_t)(p.x >> 16)),
               f16_bits_to_f32((uint16_t)(p.y & 0xffffu)), f16_bits_to_f32((uint16_t)(p.y >> 16)));
}

// restir.wgsl:132-136
FRT_HD uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

} // namespace frt
