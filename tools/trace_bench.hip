// trace_bench.hip — traversal alone, outside the renderer: what does one ray cost a wave, and which loop shape is cheapest?
//
// Builds the Cornell Box (or the 100-light ReSTIR scene: argv[1] = "restir") with the product's host code, uploads the quad tree and the
// triangle slots, generates the two ray kinds of a bounce — CLOSEST-hit rays leaving a surface point in a cosine-distributed direction and
// SHADOW (any-hit) rays from a surface point towards the quad light — in "tiles" of 64 rays whose origins lie close together (as the primary
// hits of an 8x8 pixel tile do) while their directions are independent, and times kernels that trace them with the per-lane LDS stack,
// 256-thread workgroups and 32 KiB of LDS per workgroup (4 waves per SIMD: the occupancy of the renderer's traced kernels).
// Variants (same hits, checked by checksum against variant 0):
//   0  seq     the product's trace4: shadow ray, then closest-hit ray, one after the other (what path_loop does per bounce)
//   1  early   trace4 with the node loads issued as soon as the next node is known (rotated loop)
//   2  pair    both rays of the lane walked in ONE loop: two independent dependency chains per lane (trace_pair below)
// Output: ns per ray pair, rays/s, VGPRs are in tools/kernel_resources.py --src ../tools/trace_bench.hip.
//   hipcc <Makefile flags> tools/trace_bench.hip fast-raytracing-wgpu_amd/csrc/frt_scene.cpp fast-raytracing-wgpu_amd/csrc/frt_bvh.cpp -o tools/_build/trace_bench
#include "../fast-raytracing-wgpu_amd/csrc/frt_scene.hpp"
#include "../fast-raytracing-wgpu_amd/csrc/frt_trace.hpp"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>

using namespace frt;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct RayRec { float ox, oy, oz, tmax, dx, dy, dz, tmin; };      // 32 B

// ------------------------------------------------------------------------------------------------ variant 1: early fetch
#include "trace_bench_variants.hpp"

// ------------------------------------------------------------------------------------------------ kernels
struct Out { unsigned long long occluded, hits, tri_sum, t_sum, steps; };

template <int VARIANT>
__global__ void __launch_bounds__(256, 4) bench_kernel(SceneView sc, QQuadView qv, const RayRec* __restrict__ shadow, const RayRec* __restrict__ closest, uint32_t n_pairs, uint32_t per_thread, Out* out) {
    __shared__ uint32_t s_stack[kStackDepth * 256];
    uint32_t* stk = &s_stack[threadIdx.x];
    unsigned long long occ = 0, hits = 0, tri_sum = 0, t_sum = 0;
    const uint32_t base = blockIdx.x * 256u * per_thread + (threadIdx.x >> 6) * 64u * per_thread + (threadIdx.x & 63u);
    for (uint32_t k = 0; k < per_thread; ++k) {
        const uint32_t i = base + k * 64u;       // a wave owns per_thread consecutive tiles of 64 rays
        if (i >= n_pairs) break;
        const RayRec a = shadow[i], b = closest[i];
        HitRec ha, hb;
        bool occluded;
        if (VARIANT == 0) {
            trace4<true>(sc, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, stk, 256u, ha);
            occluded = ha.tri != 0xFFFFFFFFu;
            trace4<false>(sc, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax, stk, 256u, hb);
        } else if (VARIANT == 1) {
            trace4_early<true>(sc, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, stk, 256u, ha);
            occluded = ha.tri != 0xFFFFFFFFu;
            trace4_early<false>(sc, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax, stk, 256u, hb);
        } else if (VARIANT == 5) {
            trace4_q16<true>(sc, qv, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, stk, 256u, ha);
            occluded = ha.tri != 0xFFFFFFFFu;
            trace4_q16<false>(sc, qv, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax, stk, 256u, hb);
        } else if (VARIANT == 6) {
            trace4_postpone<true>(sc, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, stk, 256u, ha);
            occluded = ha.tri != 0xFFFFFFFFu;
            trace4_postpone<false>(sc, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax, stk, 256u, hb);
        } else if (VARIANT == 7 || VARIANT == 9) {      // every ray votes (7: plain majority, 9: a node step counts 1.5 x)
            constexpr int WN = VARIANT == 7 ? 2 : 3;
            trace4_vote<true, WN, 2>(sc, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, stk, 256u, ha);
            occluded = ha.tri != 0xFFFFFFFFu;
            trace4_vote<false, WN, 2>(sc, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax, stk, 256u, hb);
        } else if (VARIANT == 8 || VARIANT == 10) {     // shadow rays while-while (the model says voting costs them 4 %), closest-hit rays vote
            constexpr int WN = VARIANT == 8 ? 2 : 3;
            trace4<true>(sc, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, stk, 256u, ha);
            occluded = ha.tri != 0xFFFFFFFFu;
            trace4_vote<false, WN, 2>(sc, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax, stk, 256u, hb);
        } else if (VARIANT == 4) {
            trace4_xload<true>(sc, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, stk, 256u, ha);
            occluded = ha.tri != 0xFFFFFFFFu;
            trace4_xload<false>(sc, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax, stk, 256u, hb);
        } else {
            trace_pair<VARIANT>(sc, true, mk3(a.ox, a.oy, a.oz), mk3(a.dx, a.dy, a.dz), a.tmin, a.tmax, true, mk3(b.ox, b.oy, b.oz), mk3(b.dx, b.dy, b.dz), b.tmin, b.tmax,
                                stk, 256u, occluded, hb);
        }
        occ += occluded ? 1u : 0u;
        if (hb.tri != 0xFFFFFFFFu) { hits += 1u; tri_sum += hb.tri; t_sum += f2u(hb.t) & 0xFFFFu; if (hb.front) t_sum += 7u; t_sum += f2u(hb.u) & 0xFFu; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        occ += __shfl_down(occ, off, 64); hits += __shfl_down(hits, off, 64); tri_sum += __shfl_down(tri_sum, off, 64); t_sum += __shfl_down(t_sum, off, 64);
    }
    // one record per wave, summed on the host (same-address atomics are served at ~13 ns each on this chip: 10^5 waves adding to one
    // word would BE the benchmark)
    if ((threadIdx.x & 63u) == 0u) { Out* o = out + (blockIdx.x * 4u + (threadIdx.x >> 6)); o->occluded = occ; o->hits = hits; o->tri_sum = tri_sum; o->t_sum = t_sum; }
}

// ------------------------------------------------------------------------------------------------ host
static uint32_t g_rng = 12345u;
static float rnd() { g_rng = g_rng * 747796405u + 2891336453u; uint32_t w = ((g_rng >> ((g_rng >> 28u) + 4u)) ^ g_rng) * 277803737u; return (float)(((w >> 22u) ^ w) >> 8) / 16777216.0f; }

template <class T> static T* upload(const std::vector<T>& v) { T* d = nullptr; CHECK(hipMalloc((void**)&d, std::max<size_t>(v.size() * sizeof(T), 16))); CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

static QQuadView g_qv;
static uint32_t g_per_thread = 2;
template <int VARIANT>
static double run(const char* name, const SceneView& sv, const RayRec* d_sh, const RayRec* d_cl, uint32_t n, Out* d_out, Out& res) {
    const uint32_t per_thread = g_per_thread;
    const uint32_t grid = (n + 256u * per_thread - 1u) / (256u * per_thread);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(d_out, 0, sizeof(Out) * grid * 4));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((bench_kernel<VARIANT>), dim3(grid), dim3(256), 0, 0, sv, g_qv, d_sh, d_cl, n, per_thread, d_out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    std::vector<Out> parts((size_t)grid * 4);
    CHECK(hipMemcpy(parts.data(), d_out, sizeof(Out) * parts.size(), hipMemcpyDeviceToHost));
    res = Out{};
    for (const Out& o : parts) { res.occluded += o.occluded; res.hits += o.hits; res.tri_sum += o.tri_sum; res.t_sum += o.t_sum; }
    printf("%-10s %8.3f ms  %7.2f Grays/s  occluded %llu hits %llu tri_sum %llu t_sum %llu\n", name, best, 2.0 * n / (best * 1e-3) / 1e9, res.occluded, res.hits, res.tri_sum, res.t_sum);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}

int main(int argc, char** argv) {
    const bool restir = argc > 1 && !strcmp(argv[1], "restir");
    SceneBuilder b;
    if (restir) scenes::create_restir_scene(b); else scenes::create_cornell_box(b);
    if (!b.built) { fprintf(stderr, "scene: %s\n", b.error.c_str()); return 1; }
    printf("%s: %zu triangles, %zu quad nodes, stack need %u\n", restir ? "restir" : "cornell", b.tris.size(), b.quad_nodes.size(), b.quad_stack_need);
    SceneView sv{};
    // one allocation: quad nodes, then the triangle slots (so that one uniform base + 32-bit offsets reaches both)
    std::vector<float> blob(b.quad_nodes.size() * 32 + b.tri_slots.size() * 12);
    memcpy(blob.data(), b.quad_nodes.data(), b.quad_nodes.size() * 128);
    memcpy(blob.data() + b.quad_nodes.size() * 32, b.tri_slots.data(), b.tri_slots.size() * 48);
    float* d_blob = upload(blob);
    sv.nodes4 = (const float4*)d_blob;
    sv.tris = (const float4*)(d_blob + b.quad_nodes.size() * 32);
    sv.nodes = (const float4*)upload(b.pair_nodes);
    sv.instances = (const InstanceView*)upload(b.instances_dev);
    sv.num_tris = (uint32_t)b.tri_slots.size(); sv.num_nodes = (uint32_t)b.pair_nodes.size();

    {   // quad nodes on the 16-bit grid of the product's quantized pair nodes (b.qmin, b.qstep), rounded outward by an extra quantum
        auto q_lo = [&](double v, int a) { double g = std::floor((v - (double)b.qmin[a]) / (double)b.qstep[a]) - 1.0; return (uint32_t)std::min(65535.0, std::max(0.0, g)); };
        auto q_hi = [&](double v, int a) { double g = std::ceil((v - (double)b.qmin[a]) / (double)b.qstep[a]) + 1.0; return (uint32_t)std::min(65535.0, std::max(0.0, g)); };
        std::vector<uint32_t> qq(b.quad_nodes.size() * 16);
        for (size_t i = 0; i < b.quad_nodes.size(); ++i) {
            const QuadNode& q = b.quad_nodes[i];
            uint32_t* w = &qq[i * 16];
            for (int a = 0; a < 3; ++a) {
                uint32_t lo[4], hi[4];
                for (int c = 0; c < 4; ++c) {
                    uint32_t ref; memcpy(&ref, &q.q[24 + c], 4);
                    if (ref == kNoChild) { lo[c] = 65535u; hi[c] = 0u; }
                    else { lo[c] = q_lo(q.q[8 * a + c], a); hi[c] = q_hi(q.q[8 * a + 4 + c], a); }
                }
                w[4 * a + 0] = lo[0] | (lo[1] << 16); w[4 * a + 1] = lo[2] | (lo[3] << 16); w[4 * a + 2] = hi[0] | (hi[1] << 16); w[4 * a + 3] = hi[2] | (hi[3] << 16);
            }
            memcpy(&w[12], &q.q[24], 16);
        }
        g_qv.nodes = (const uint4*)upload(qq);
        g_qv.qmin = mk3(b.qmin[0], b.qmin[1], b.qmin[2]); g_qv.qstep = mk3(b.qstep[0], b.qstep[1], b.qstep[2]);
    }
    // rays: tiles of 64 with a common base point
    // argv[3]: log2 of the ray pairs (default 23: ~16 rounds of workgroups, the throughput regime; 18 = ONE round: the latency regime of a thin image strip)
    const uint32_t n = 1u << (argc > 3 ? atoi(argv[3]) : 23);
    std::vector<double> cum(b.tris.size());
    double acc = 0.0;
    auto crossv = [](const float* a, const float* c, float* o) { o[0] = a[1] * c[2] - a[2] * c[1]; o[1] = a[2] * c[0] - a[0] * c[2]; o[2] = a[0] * c[1] - a[1] * c[0]; };
    for (size_t i = 0; i < b.tris.size(); ++i) { float nn[3]; crossv(b.tris[i].e1, b.tris[i].e2, nn); acc += 0.5 * std::sqrt((double)nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]); cum[i] = acc; }
    std::vector<RayRec> sh(n), cl(n);
    const frt_light& L = b.lights[0];
    for (uint32_t t = 0; t < n / 64; ++t) {
        const size_t ti = std::lower_bound(cum.begin(), cum.end(), rnd() * acc) - cum.begin();
        const TriRec& T = b.tris[std::min(ti, b.tris.size() - 1)];
        float u = rnd(), v = rnd(); if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
        float p[3], nn[3];
        crossv(T.e1, T.e2, nn);
        const float nl = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
        for (int a = 0; a < 3; ++a) { p[a] = T.v0[a] + T.e1[a] * u + T.e2[a] * v; nn[a] /= nl; }
        // the side of the surface that faces the middle of the scene
        const float toc = -(p[0] * nn[0] + p[1] * nn[1] + p[2] * nn[2]);
        if (toc < 0.0f) for (int a = 0; a < 3; ++a) nn[a] = -nn[a];
        float tb[3] = {T.e1[0], T.e1[1], T.e1[2]};
        const float tl = std::sqrt(tb[0] * tb[0] + tb[1] * tb[1] + tb[2] * tb[2]);
        for (int a = 0; a < 3; ++a) tb[a] /= tl;
        float bt[3]; crossv(nn, tb, bt);
        for (uint32_t k = 0; k < 64; ++k) {
            const uint32_t i = t * 64 + k;
            const float ju = (rnd() - 0.5f) * 0.04f, jv = (rnd() - 0.5f) * 0.04f;
            float o[3];
            for (int a = 0; a < 3; ++a) o[a] = p[a] + tb[a] * ju + bt[a] * jv + nn[a] * 0.001f;
            // cosine-distributed direction about nn
            const float r1 = rnd(), r2 = rnd();
            const float rr = std::sqrt(r1), ph = 6.2831853f * r2, cz = std::sqrt(std::max(0.0f, 1.0f - r1));
            float d[3];
            for (int a = 0; a < 3; ++a) d[a] = tb[a] * rr * std::cos(ph) + bt[a] * rr * std::sin(ph) + nn[a] * cz;
            cl[i] = RayRec{o[0], o[1], o[2], 100.0f, d[0], d[1], d[2], 0.001f};
            // shadow ray towards a point of light 0
            const float su = rnd() * 2.0f - 1.0f, sv2 = rnd() * 2.0f - 1.0f;
            float tg[3], dd[3];
            for (int a = 0; a < 3; ++a) { tg[a] = L.position[a] + L.u[a] * su + L.v[a] * sv2; dd[a] = tg[a] - o[a]; }
            const float dl = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
            sh[i] = RayRec{o[0], o[1], o[2], dl * 0.999f, dd[0] / dl, dd[1] / dl, dd[2] / dl, 0.001f};
        }
    }
    // argv[5]: ray order. 0 = tiles (above). 1 = "queue order": the pairs shuffled, every lane of a wave from a different tile (what a continuation
    // launch sees: parked paths in arrival order). 2 = shuffled, then binned by the direction octant of the closest-hit ray and a 2x2x2 cell of its
    // origin (64 bins, stable). 3 = binned by octant only.
    const int order = argc > 5 ? atoi(argv[5]) : 0;
    if (order > 0 && order < 7) {
        std::vector<uint32_t> perm(n);
        for (uint32_t i = 0; i < n; ++i) perm[i] = i;
        for (uint32_t i = n - 1; i > 0; --i) { const uint32_t j = (uint32_t)(rnd() * (float)(i + 1)) % (i + 1); std::swap(perm[i], perm[j]); }
        if (order >= 2) {
            auto key = [&](uint32_t i) {
                const RayRec& r = cl[i];
                uint32_t k = (r.dx < 0 ? 1u : 0u) | (r.dy < 0 ? 2u : 0u) | (r.dz < 0 ? 4u : 0u);
                if (order == 2) k |= ((r.ox > 0 ? 1u : 0u) | (r.oy > 0 ? 2u : 0u) | (r.oz > 0 ? 4u : 0u)) << 3;
                // 4: 2x2x2 origin cell only; 5: 4x4x4 cells only; 6: 4x4x4 cells + octant (scene box [-1, 1]^3)
                auto c4 = [](float v) { int c = (int)((v + 1.0f) * 2.0f); return (uint32_t)(c < 0 ? 0 : (c > 3 ? 3 : c)); };
                if (order == 4) k = (r.ox > 0 ? 1u : 0u) | (r.oy > 0 ? 2u : 0u) | (r.oz > 0 ? 4u : 0u);
                if (order == 5) k = c4(r.ox) | (c4(r.oy) << 2) | (c4(r.oz) << 4);
                if (order == 6) k |= (c4(r.ox) | (c4(r.oy) << 2) | (c4(r.oz) << 4)) << 3;
                return k;
            };
            std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b2) { return key(a) < key(b2); });
        }
        std::vector<RayRec> sh2(n), cl2(n);
        for (uint32_t i = 0; i < n; ++i) { sh2[i] = sh[perm[i]]; cl2[i] = cl[perm[i]]; }
        sh.swap(sh2); cl.swap(cl2);
    }
    // 7 / 8: tile order kept, but inside every block of 256 (7) or 64 (8) consecutive pairs — a workgroup's / a wave's rays — the pairs are sorted by
    // the direction octant of the closest-hit ray: what exchanging rays between the lanes of a workgroup before a bounce could buy
    if (order == 7 || order == 8) {
        const uint32_t blk = order == 7 ? 256u : 64u;
        auto oct = [&](const RayRec& r) { return (r.dx < 0 ? 1u : 0u) | (r.dy < 0 ? 2u : 0u) | (r.dz < 0 ? 4u : 0u); };
        for (uint32_t b0 = 0; b0 + blk <= n; b0 += blk) {
            std::vector<uint32_t> idx(blk);
            for (uint32_t i = 0; i < blk; ++i) idx[i] = b0 + i;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b2) { return oct(cl[a]) < oct(cl[b2]); });
            std::vector<RayRec> s2(blk), c2(blk);
            for (uint32_t i = 0; i < blk; ++i) { s2[i] = sh[idx[i]]; c2[i] = cl[idx[i]]; }
            std::copy(s2.begin(), s2.end(), sh.begin() + b0); std::copy(c2.begin(), c2.end(), cl.begin() + b0);
        }
    }
    RayRec* d_sh = upload(sh); RayRec* d_cl = upload(cl);
    Out* d_out; CHECK(hipMalloc((void**)&d_out, sizeof(Out) * (size_t)(n / 64 + 1024)));
    Out r0{}, r{};
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    if (argc > 4) g_per_thread = (uint32_t)atoi(argv[4]);
    run<0>("seq", sv, d_sh, d_cl, n, d_out, r0);
    auto same = [&](const Out& x) { return x.occluded == r0.occluded && x.hits == r0.hits && x.tri_sum == r0.tri_sum && x.t_sum == r0.t_sum; };
    bool ok = true;
    if (only < 0 || only == 1) { run<1>("early", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 2) { run<2>("pair", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 3) { run<3>("pair3", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 4) { run<4>("early+4ld", sv, d_sh, d_cl, n, d_out, r); }
    if (only < 0 || only == 5) { run<5>("q16 4ld", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 6) { run<6>("postpone", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 7) { run<7>("vote", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 8) { run<8>("vote cl", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 9) { run<9>("vote 3:2", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    if (only < 0 || only == 10) { run<10>("vote3:2 cl", sv, d_sh, d_cl, n, d_out, r); ok &= same(r); }
    printf(ok ? "checksums equal\n" : "CHECKSUM MISMATCH\n");
    return ok ? 0 : 2;
}
