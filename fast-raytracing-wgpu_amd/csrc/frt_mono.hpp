// frt_mono.hpp — trace_path (restir.wgsl:460-737 / restir_spatial.wgsl:480-762) and the temporal / spatial stage bodies as
// straight-line per-pixel functions: the form the default kernels run (one thread per pixel). The resumable form of the same
// arithmetic, used by the compacting kernels, is frt_path.hpp; both are checked bit for bit against the oracle.
#pragma once
#include "frt_path.hpp"

namespace frt {

struct Surf {        // HitInfo subset, restir.wgsl:81-90
    f3 pos, normal, ffnormal; f2 uv; bool front_face; float t; f4 tangent; uint32_t mat_id;
};

// restir.wgsl:375-381 (VARIANT 0) vs restir_spatial.wgsl:380-400 (VARIANT 1). true = unoccluded.
template <int VARIANT, class Ctx>
FRT_HD bool trace_shadow_ray(Ctx& c, f3 origin, f3 dir, float dist) {
    float t_max = fmaxn(dist * 0.999f, 0.0f);
    float t_min = VARIANT == 0 ? 0.001f : 0.0001f;
    if (VARIANT == 1 && t_min >= t_max) return true;
    return !c.any(origin, dir, t_min, t_max);
}

template <int VARIANT, class Ctx>
FRT_HD f3 eval_direct_lighting(Ctx& c, const Surf& hit, f3 wo, const MatParams& m, f3 base_color, const LightSmp& ls, float weight) {   // :443-459
    f3 offset_pos = hit.pos + hit.ffnormal * 0.001f;
    f3 L = normalize(ls.pos - offset_pos);
    float dist = distance(ls.pos, offset_pos);
    float n_dot_l = fmaxn(dot(hit.ffnormal, L), 0.0f);
    float l_dot_n = fmaxn(dot(-L, ls.normal), 0.0f);
    if (n_dot_l > 0.0f && l_dot_n > 0.0f) {
        if (trace_shadow_ray<VARIANT>(c, offset_pos, L, dist)) {
            f3 f = eval_bsdf(hit.ffnormal, L, wo, m, base_color);
            float G = (n_dot_l * l_dot_n) / (dist * dist);
            return xyz(ls.emission) * ls.emission.w * f * G * weight;
        }
    }
    return splat3(0.0f);
}
template <int VARIANT, class Ctx>
FRT_HD f3 nee(Ctx& c, const Surf& hit, f3 wo, const MatParams& m, f3 base_color, f3 throughput) {   // :558-571 == :707-720
    uint32_t nl = c.fv.cam.num_lights;
    if (nl > 0u) {
        uint32_t light_idx = (uint32_t)(c.rand() * (float)nl);
        if (light_idx < nl) {
            LightSmp ls = sample_light(c, light_idx);
            float pdf_nee = ls.pdf * (1.0f / (float)nl);
            float p_bsdf = eval_pdf(hit.ffnormal, normalize(ls.pos - hit.pos), wo, m, base_color);
            float mis_weight_nee = pdf_nee / (pdf_nee + p_bsdf);
            float weight = mis_weight_nee / pdf_nee;
            return eval_direct_lighting<VARIANT>(c, hit, wo, m, base_color, ls, weight) * throughput;
        }
    }
    return splat3(0.0f);
}

// ---- one bounce with its two rays pulled apart (the form the stream kernel runs, frt_kernels.hip) -----------------------------
// path_loop traces the closest-hit ray, shades, traces the shadow ray of the next-event estimate in the middle of the shading, and
// goes on. Here one iteration is cut at its rays:   [closest-hit ray]  ->  bounce_shade  ->  [shadow ray]  ->  add the estimate
// bounce_shade does everything else of the iteration — hit, material, emission, light hit, the estimate's value (as if unoccluded), BSDF
// sample, the next iteration's roulette — and hands back the shadow ray with the value to add when it is unoccluded. Same operations
// on the same operands as path_loop: the estimate is a pure function of the hit (evaluating it before the visibility test instead of
// after changes nothing), it is the only addition to `accumulated` between the light-hit test and the end of the iteration, nothing
// after it reads `accumulated`, and no ray draws a random number, so the rand() sequence is the reference's.
// dark: what path_loop adds when the estimate is NOT lit — zero, or zero times the throughput (NaN where the throughput is inf / NaN: the
// reference's GGX term overflows on the roughness-0.01 box), so that even those paths stay bit-identical.
struct ShadowReq { bool want; bool add_now; f3 o, d; float tmin, tmax; f3 contrib, dark; };

template <int VARIANT, class Ctx>
FRT_HD void nee_request(Ctx& c, const Surf& hit, f3 wo, const MatParams& m, f3 base_color, f3 throughput, ShadowReq& req) {   // nee() + eval_direct_lighting() up to the ray
    req.want = false; req.add_now = false; req.contrib = splat3(0.0f); req.dark = splat3(0.0f);
    req.o = splat3(0.0f); req.d = splat3(0.0f); req.tmin = 0.0f; req.tmax = 0.0f;
    uint32_t nl = c.fv.cam.num_lights;
    if (nl == 0u) return;
    uint32_t light_idx = (uint32_t)(c.rand() * (float)nl);
    if (!(light_idx < nl)) return;
    req.dark = splat3(0.0f) * throughput;      // from here on nee() returns eval_direct_lighting(...) * throughput
    LightSmp ls = sample_light(c, light_idx);
    float pdf_nee = ls.pdf * (1.0f / (float)nl);
    float p_bsdf = eval_pdf(hit.ffnormal, normalize(ls.pos - hit.pos), wo, m, base_color);
    float mis_weight_nee = pdf_nee / (pdf_nee + p_bsdf);
    float weight = mis_weight_nee / pdf_nee;
    f3 offset_pos = hit.pos + hit.ffnormal * 0.001f;
    f3 L = normalize(ls.pos - offset_pos);
    float dist = distance(ls.pos, offset_pos);
    float n_dot_l = fmaxn(dot(hit.ffnormal, L), 0.0f);
    float l_dot_n = fmaxn(dot(-L, ls.normal), 0.0f);
    if (n_dot_l > 0.0f && l_dot_n > 0.0f) {
        float t_max = fmaxn(dist * 0.999f, 0.0f);
        float t_min = VARIANT == 0 ? 0.001f : 0.0001f;
        f3 f = eval_bsdf(hit.ffnormal, L, wo, m, base_color);
        float G = (n_dot_l * l_dot_n) / (dist * dist);
        req.contrib = (xyz(ls.emission) * ls.emission.w * f * G * weight) * throughput;
        if (VARIANT == 1 && t_min >= t_max) req.add_now = true;      // restir_spatial.wgsl:380-400: "too close" counts as unoccluded, no ray
        else { req.want = true; req.o = offset_pos; req.d = L; req.tmin = t_min; req.tmax = t_max; }
    }
}

// State of a path between two bounce iterations (restir.wgsl:590), after the Russian-roulette test that opens the next iteration:
// everything the rest of that iteration reads that earlier code wrote. A path can be cut here, parked in HBM (frt_kernels.hip:
// continuation queue) and resumed by another lane.
struct LoopState {
    f3 pos, ffnormal, throughput, accumulated, next_dir, v1_pos;
    float last_bsdf_pdf;
    bool previous_was_diffuse, is_glass, alive;
};

// restir.wgsl:460-584 (VARIANT 0) / restir_spatial.wgsl:480-610 (VARIANT 1): the primary hit, read from the G-buffer.
// Returns with s.alive == false when the path ended here (background, light surface, zero BSDF weight).
// req != nullptr: the shadow ray of the primary hit's next-event estimate is NOT traced here but handed back (nee_request): the caller adds
// (req->add_now || (req->want && unoccluded)) ? req->contrib : req->dark to s.accumulated — the only addition path_head makes after it, so the
// sum has the same operands in the same order, and no ray draws a random number (the collective kernels, frt_kernels.hip: pixel_kernel_wg).
template <int VARIANT, class Ctx>
FRT_HD void path_head(Ctx& c, uint32_t pix, uint32_t seed, LoopState& s, ShadowReq* req = nullptr) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    if (req) { req->want = false; req->add_now = false; req->contrib = splat3(0.0f); req->dark = splat3(0.0f); req->o = splat3(0.0f); req->d = splat3(0.0f); req->tmin = 0.0f; req->tmax = 0.0f; }
    c.rng = seed;
    s.accumulated = splat3(0.0f); s.v1_pos = splat3(0.0f); s.throughput = splat3(1.0f); s.next_dir = splat3(0.0f);
    s.pos = splat3(0.0f); s.ffnormal = splat3(0.0f); s.last_bsdf_pdf = 0.0f;
    s.previous_was_diffuse = false; s.is_glass = false; s.alive = false;
    float4 pos_w = fv.gpos[pix];
    if (pos_w.w < 0.0f) return;
    float4 normal_w = fv.gnormal[pix];
    f4 albedo_raw = unpack_rgba8(fv.galbedo[pix]);

    Surf hit;
    hit.pos = mk3(pos_w.x, pos_w.y, pos_w.z);
    hit.normal = decode_octahedral_normal(normal_w.x, normal_w.y);
    hit.front_face = true;
    hit.ffnormal = hit.normal;
    hit.uv = mk2(normal_w.z, normal_w.w);
    hit.t = 0.0f; hit.tangent = mk4(0, 0, 0, 0);

    uint32_t mat_id = (uint32_t)(pos_w.w + 0.1f);
    MatParams m; f3 base_color; f3 emissive_factor; int32_t light_index; uint32_t tex1, tex2;
    if (mat_id < sc.num_materials) {
        const MaterialView& ms = sc.materials[mat_id];
        m.roughness = ms.roughness; m.metallic = ms.metallic; m.transmission = ms.transmission; m.ior = ms.ior;
        light_index = ms.light_index; tex1 = ms.tex_info_1; tex2 = ms.tex_info_2;
        emissive_factor = mk3(ms.emissive_factor[0], ms.emissive_factor[1], ms.emissive_factor[2]);
        if (VARIANT == 0 || m.transmission < 0.01f) base_color = xyz(albedo_raw);     // restir.wgsl:494 vs restir_spatial.wgsl:514-516
        else base_color = mk3(ms.base_color[0], ms.base_color[1], ms.base_color[2]);
    } else {   // restir.wgsl:495-501: zero-initialised `var mat` with four fields set
        m.roughness = 0.0f; m.metallic = albedo_raw.w; m.transmission = 0.0f; m.ior = 1.0f;
        light_index = -1; tex1 = 0u; tex2 = 0u; emissive_factor = splat3(0.0f);
        base_color = xyz(albedo_raw);
    }
    uint32_t mr_tex_id = tex2 & 0xFFFFu;
    if (mr_tex_id != 65535u) {
        f4 mr = sample_layer<false>(sc, mr_tex_id, hit.uv);
        m.metallic = mr.z * m.metallic;
        m.roughness = mr.y * m.roughness;
    }
    f3 accumulated = splat3(0.0f);
    f3 throughput = splat3(1.0f);
    f3 wo = normalize(mk3(fv.cam.view_pos[0], fv.cam.view_pos[1], fv.cam.view_pos[2]) - hit.pos);
    uint32_t emissive_tex_id = tex1 >> 16u;

    if (mat_id < sc.num_materials && light_index == -1) {   // :523-533
        f3 emission = emissive_factor;
        if (emissive_tex_id != 65535u) emission = emission * xyz(sample_layer<true>(sc, emissive_tex_id, hit.uv));
        accumulated = accumulated + emission;
    }
    if (light_index >= 0) {   // :543-552
        f3 emission = emissive_factor;
        if (emissive_tex_id != 65535u) emission = emission * xyz(sample_layer<true>(sc, emissive_tex_id, hit.uv));
        accumulated = accumulated + emission;
        s.accumulated = accumulated;
        return;
    }
    const bool is_glass = m.transmission > 0.01f;
    bool previous_was_diffuse;
    if (!(is_glass || m.roughness < 0.05f)) {   // :556
        if (req) nee_request<VARIANT>(c, hit, wo, m, base_color, throughput, *req);
        else accumulated = accumulated + nee<VARIANT>(c, hit, wo, m, base_color, throughput);
        previous_was_diffuse = true;
    } else previous_was_diffuse = false;

    BsdfSmp sc0 = sample_bsdf(c, wo, hit.ffnormal, hit.front_face, m, base_color);
    s.accumulated = accumulated;
    if (sc0.weight.x <= 0.0f && sc0.weight.y <= 0.0f && sc0.weight.z <= 0.0f) return;
    s.last_bsdf_pdf = sc0.pdf;
    s.throughput = throughput * sc0.weight;
    s.next_dir = sc0.wi;
    s.pos = hit.pos; s.ffnormal = hit.ffnormal;
    s.previous_was_diffuse = previous_was_diffuse; s.is_glass = is_glass;
    s.alive = 1u < fv.max_depth;      // iteration 1 exists (no roulette before depth 3)
}

// Bounce loop, iterations depth_begin .. depth_end-1 of `for (depth = 1; depth < MAX_DEPTH; depth++)` (restir.wgsl:590-733).
// On return s.alive tells whether the path is still running (it reached depth_end without terminating).
template <int VARIANT, class Ctx>
FRT_HD void path_loop(Ctx& c, LoopState& s, uint32_t depth_begin, uint32_t depth_end) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    Surf hit;
    hit.pos = s.pos; hit.ffnormal = s.ffnormal; hit.normal = s.ffnormal; hit.front_face = true;
    hit.uv = mk2(0.0f, 0.0f); hit.t = 0.0f; hit.tangent = mk4(0, 0, 0, 0); hit.mat_id = 0u;
    f3 accumulated = s.accumulated, throughput = s.throughput, next_dir = s.next_dir;
    float last_bsdf_pdf = s.last_bsdf_pdf;
    bool previous_was_diffuse = s.previous_was_diffuse;
    const bool is_glass = s.is_glass;
    bool alive = true;
    for (uint32_t depth = depth_begin; depth < depth_end; depth++) {
        alive = false;
        f3 offset_dir = hit.ffnormal * signf(dot(hit.ffnormal, next_dir));
        f3 origin = hit.pos + offset_dir * 0.001f;
        HitRec h;
        c.closest(origin, next_dir, 0.001f, 100.0f, h);
        if (h.tri == 0xFFFFFFFFu) break;
        HitGeom g = fetch_hit_geometry(sc, h);   // reconstruct_geometry_hit, :383-441
        hit.normal = g.normal_w;
        hit.uv = g.uv;
        hit.front_face = h.front;
        hit.ffnormal = h.front ? g.normal_w : -g.normal_w;
        hit.t = h.t;
        hit.pos = origin + next_dir * h.t;
        hit.mat_id = g.mat_id;
        if (depth == 1u) s.v1_pos = hit.pos;
        f3 wo = -next_dir;
        const MaterialView& mb = sc.materials[hit.mat_id];
        MatParams m;
        m.roughness = mb.roughness; m.metallic = mb.metallic; m.transmission = mb.transmission; m.ior = mb.ior;
        int32_t light_index_b = mb.light_index;
        uint32_t t0i = mb.tex_info_0, t1i = mb.tex_info_1;
        f4 tex_color = mk4(1.0f, 1.0f, 1.0f, 1.0f);
        uint32_t tex_id = t0i & 0xFFFFu, normal_tex_id = t0i >> 16u;
        if (tex_id != 65535u) tex_color = sample_layer<true>(sc, tex_id, hit.uv);
        float occlusion = 1.0f;
        uint32_t occlusion_tex_id = t1i & 0xFFFFu, emissive_tex_id_b = t1i >> 16u;
        if (occlusion_tex_id != 65535u) occlusion = sample_layer<false>(sc, occlusion_tex_id, hit.uv).x;
        f3 base_color = mk3(mb.base_color[0], mb.base_color[1], mb.base_color[2]) * xyz(tex_color) * occlusion;
        if (normal_tex_id != 65535u) {
            f3 nm = xyz(sample_layer<false>(sc, normal_tex_id, hit.uv));
            f4 tg = hit_tangent(sc, g);
            hit.ffnormal = perturb_normal(hit.ffnormal, xyz(tg), tg.w, nm);
        }
        if (light_index_b == -1 && emissive_tex_id_b != 65535u) {   // :675-678
            f3 emissive_col = xyz(sample_layer<true>(sc, emissive_tex_id_b, hit.uv));
            accumulated = accumulated + emissive_col * throughput;
        }
        if (light_index_b >= 0) {   // :683-700
            if (hit.front_face) {
                const LightView& light = sc.lights[light_index_b];
                f3 Le = mk3(light.emission[0], light.emission[1], light.emission[2]) * light.emission[3];
                float mis_weight = 1.0f;
                if (previous_was_diffuse) {
                    float dist_sq = hit.t * hit.t;
                    float light_cos = fmaxn(dot(hit.ffnormal, -wo), 0.0f);
                    float p_bsdf = last_bsdf_pdf;
                    float p_nee = (1.0f / light.area) * (dist_sq / light_cos) * (1.0f / (float)fv.cam.num_lights);
                    if (light_cos > 0.001f) mis_weight = p_bsdf / (p_bsdf + p_nee);
                    else mis_weight = 0.0f;
                }
                accumulated = accumulated + Le * throughput * mis_weight;
            }
            break;
        }
        if (!(is_glass || m.roughness < 0.05f)) {   // :705 — the PRIMARY hit's is_glass (reference quirk)
            accumulated = accumulated + nee<VARIANT>(c, hit, wo, m, base_color, throughput);
            previous_was_diffuse = true;
        } else previous_was_diffuse = false;
        BsdfSmp sb = sample_bsdf(c, wo, hit.ffnormal, hit.front_face, m, base_color);
        if (sb.weight.x <= 0.0f && sb.weight.y <= 0.0f && sb.weight.z <= 0.0f) break;
        last_bsdf_pdf = sb.pdf;
        throughput = throughput * sb.weight;
        next_dir = sb.wi;
        // Russian roulette of the NEXT iteration (restir.wgsl:593-598, the first statements of the loop body), taken here so that a
        // path the roulette kills is never parked at a cut. Nothing draws from the stream between sample_bsdf and this test, and the
        // draw is skipped exactly when the reference's loop condition would end the loop, so the sequence of rand() calls is the same.
        const uint32_t next = depth + 1u;
        if (next >= fv.max_depth) break;
        if (next >= 3u) {
            float p = fmaxn(throughput.x, fmaxn(throughput.y, throughput.z));
            float survival_prob = clampf(p, 0.05f, 0.95f);
            if (c.rand() > survival_prob) break;
            throughput = throughput / survival_prob;
        }
        alive = true;
    }
    s.pos = hit.pos; s.ffnormal = hit.ffnormal; s.accumulated = accumulated; s.throughput = throughput; s.next_dir = next_dir;
    s.last_bsdf_pdf = last_bsdf_pdf; s.previous_was_diffuse = previous_was_diffuse;
    s.alive = alive;      // still running: the roulette for iteration depth_end has been passed
}

// Origin of the closest-hit ray of the iteration that starts from state `s` (restir.wgsl:600-605).
FRT_HD f3 bounce_origin(const LoopState& s) {
    f3 offset_dir = s.ffnormal * signf(dot(s.ffnormal, s.next_dir));
    return s.pos + offset_dir * 0.001f;
}
// Iteration `depth` after its closest-hit ray returned `h` (fired from bounce_origin(s) along s.next_dir). Leaves s as path_loop would
// at the end of the iteration (s.alive: the next iteration exists and its roulette has been passed) except for the estimate, which the
// caller adds:  s.accumulated += (req.add_now || (req.want && unoccluded)) ? req.contrib : req.dark.
template <int VARIANT>
FRT_HD void bounce_shade(PathCtx& c, LoopState& s, uint32_t depth, const HitRec& h, ShadowReq& req) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    req.want = false; req.add_now = false; req.contrib = splat3(0.0f); req.dark = splat3(0.0f);
    req.o = splat3(0.0f); req.d = splat3(0.0f); req.tmin = 0.0f; req.tmax = 0.0f;
    s.alive = false;
    if (h.tri == 0xFFFFFFFFu) return;
    const f3 origin = bounce_origin(s);
    const f3 next_dir = s.next_dir;
    HitGeom g = fetch_hit_geometry(sc, h);
    Surf hit;
    hit.normal = g.normal_w; hit.uv = g.uv; hit.front_face = h.front;
    hit.ffnormal = h.front ? g.normal_w : -g.normal_w;
    hit.t = h.t; hit.pos = origin + next_dir * h.t; hit.mat_id = g.mat_id; hit.tangent = mk4(0, 0, 0, 0);
    if (depth == 1u) s.v1_pos = hit.pos;
    f3 wo = -next_dir;
    const MaterialView& mb = sc.materials[hit.mat_id];
    MatParams m;
    m.roughness = mb.roughness; m.metallic = mb.metallic; m.transmission = mb.transmission; m.ior = mb.ior;
    int32_t light_index_b = mb.light_index;
    uint32_t t0i = mb.tex_info_0, t1i = mb.tex_info_1;
    f4 tex_color = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    uint32_t tex_id = t0i & 0xFFFFu, normal_tex_id = t0i >> 16u;
    if (tex_id != 65535u) tex_color = sample_layer<true>(sc, tex_id, hit.uv);
    float occlusion = 1.0f;
    uint32_t occlusion_tex_id = t1i & 0xFFFFu, emissive_tex_id_b = t1i >> 16u;
    if (occlusion_tex_id != 65535u) occlusion = sample_layer<false>(sc, occlusion_tex_id, hit.uv).x;
    f3 base_color = mk3(mb.base_color[0], mb.base_color[1], mb.base_color[2]) * xyz(tex_color) * occlusion;
    if (normal_tex_id != 65535u) {
        f3 nm = xyz(sample_layer<false>(sc, normal_tex_id, hit.uv));
        f4 tg = hit_tangent(sc, g);
        hit.ffnormal = perturb_normal(hit.ffnormal, xyz(tg), tg.w, nm);
    }
    s.pos = hit.pos; s.ffnormal = hit.ffnormal;
    if (light_index_b == -1 && emissive_tex_id_b != 65535u) {   // :675-678
        f3 emissive_col = xyz(sample_layer<true>(sc, emissive_tex_id_b, hit.uv));
        s.accumulated = s.accumulated + emissive_col * s.throughput;
    }
    if (light_index_b >= 0) {   // :683-700
        if (hit.front_face) {
            const LightView& light = sc.lights[light_index_b];
            f3 Le = mk3(light.emission[0], light.emission[1], light.emission[2]) * light.emission[3];
            float mis_weight = 1.0f;
            if (s.previous_was_diffuse) {
                float dist_sq = hit.t * hit.t;
                float light_cos = fmaxn(dot(hit.ffnormal, -wo), 0.0f);
                float p_bsdf = s.last_bsdf_pdf;
                float p_nee = (1.0f / light.area) * (dist_sq / light_cos) * (1.0f / (float)fv.cam.num_lights);
                if (light_cos > 0.001f) mis_weight = p_bsdf / (p_bsdf + p_nee);
                else mis_weight = 0.0f;
            }
            s.accumulated = s.accumulated + Le * s.throughput * mis_weight;
        }
        return;
    }
    if (!(s.is_glass || m.roughness < 0.05f)) {   // :705 — the PRIMARY hit's is_glass (reference quirk)
        nee_request<VARIANT>(c, hit, wo, m, base_color, s.throughput, req);
        s.previous_was_diffuse = true;
    } else s.previous_was_diffuse = false;
    BsdfSmp sb = sample_bsdf(c, wo, hit.ffnormal, hit.front_face, m, base_color);
    if (sb.weight.x <= 0.0f && sb.weight.y <= 0.0f && sb.weight.z <= 0.0f) return;
    s.last_bsdf_pdf = sb.pdf;
    s.throughput = s.throughput * sb.weight;
    s.next_dir = sb.wi;
    const uint32_t next = depth + 1u;
    if (next >= fv.max_depth) return;
    if (next >= 3u) {
        float p = fmaxn(s.throughput.x, fmaxn(s.throughput.y, s.throughput.z));
        float survival_prob = clampf(p, 0.05f, 0.95f);
        if (c.rand() > survival_prob) return;
        s.throughput = s.throughput / survival_prob;
    }
    s.alive = true;
}
// path_loop in the pulled-apart form, one lane at a time (tests/hostcheck: must equal path_loop bit for bit).
template <int VARIANT, class Ctx>
FRT_HD void path_loop_split(Ctx& c, LoopState& s, uint32_t depth_begin, uint32_t depth_end) {
    bool alive = true;
    for (uint32_t depth = depth_begin; depth < depth_end && alive; depth++) {
        HitRec h;
        c.closest(bounce_origin(s), s.next_dir, 0.001f, 100.0f, h);
        ShadowReq req;
        bounce_shade<VARIANT>(c, s, depth, h, req);
        bool lit = req.add_now;
        if (req.want) lit = !c.any(req.o, req.d, req.tmin, req.tmax);
        s.accumulated = s.accumulated + (lit ? req.contrib : req.dark);
        alive = s.alive;
    }
    s.alive = alive;
}

// ---- continuation records: a LoopState parked in HBM between two launches -------------------------------------------------
// SoA over slots: word k of slot i lives at words[k * capacity + i], so a wave parking / fetching consecutive slots moves full
// 256-byte rows per word. 22 words (88 B) per path; the spatial stage adds what its finalisation reads of the merged reservoir: y (the seed being
// re-traced), w_sum and M — spatial_finalize (frt_path.hpp, restir_spatial.wgsl:996-1015) overwrites W, s_path and p_hat, so those five words neither
// travel through the queue nor stay live in registers across the bounces.
// `count` may run past `capacity`: a path that finds the queue full is finished in place by the lane that holds it (never dropped) and
// counted in `overflow`; readers use min(*count, capacity) slots.
// nsub > 1: the queue is cut into nsub regions of capacity / nsub slots, each with its own counter (count[0 .. nsub)): tens of thousands of
// atomics on ONE address cost ~13 ns each on this chip (measured: 8,192 waves taking one ticket each = 120 us), eight addresses run side by side.
// The two words behind *overflow + 2 hold a pointer (or null) to a word of host memory mapped into the device; the continuation launch that finds
// its input queue overfull stores 1 there, so that the host notices at its next call — without a copy, a synchronisation or a stats query —
// that the queues want growing (frt_kernels.hip: continue_kernel; frt_renderer.hip: grow_queues_if_overflowed).
struct ContQueue { uint32_t* words; uint32_t* count; uint32_t capacity; uint32_t* overflow; uint32_t nsub; };
static constexpr int kOverflowBlockWords = 4;      // per traced stage: {count, pad, pointer to the mapped flag (2 words)}
static constexpr int kContWordsPath = 22, kContWordsSpatial = 25;

FRT_HD void cont_store(const ContQueue& q, uint32_t slot, uint32_t pix, uint32_t rng, bool owned, const LoopState& s, const ReservoirView* r) {
    uint32_t* w = q.words + slot;
    const size_t cap = q.capacity;
    w[0 * cap] = pix; w[1 * cap] = rng;
    w[2 * cap] = (s.previous_was_diffuse ? 1u : 0u) | (s.is_glass ? 2u : 0u) | (owned ? 4u : 0u);
    const float f[19] = {s.pos.x, s.pos.y, s.pos.z, s.ffnormal.x, s.ffnormal.y, s.ffnormal.z, s.throughput.x, s.throughput.y, s.throughput.z,
                         s.accumulated.x, s.accumulated.y, s.accumulated.z, s.next_dir.x, s.next_dir.y, s.next_dir.z,
                         s.v1_pos.x, s.v1_pos.y, s.v1_pos.z, s.last_bsdf_pdf};
#pragma unroll
    for (int k = 0; k < 19; ++k) w[(size_t)(3 + k) * cap] = f2u(f[k]);
    if (r) { w[22 * cap] = r->y; w[23 * cap] = f2u(r->w_sum); w[24 * cap] = r->M; }
}
FRT_HD void cont_load(const ContQueue& q, uint32_t slot, uint32_t& pix, uint32_t& rng, bool& owned, LoopState& s, ReservoirView* r) {
    const uint32_t* w = q.words + slot;
    const size_t cap = q.capacity;
    pix = w[0 * cap]; rng = w[1 * cap];
    uint32_t fl = w[2 * cap];
    s.previous_was_diffuse = fl & 1u; s.is_glass = fl & 2u; owned = fl & 4u;
    float f[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) f[k] = u2f(w[(size_t)(3 + k) * cap]);
    s.pos = mk3(f[0], f[1], f[2]); s.ffnormal = mk3(f[3], f[4], f[5]); s.throughput = mk3(f[6], f[7], f[8]);
    s.accumulated = mk3(f[9], f[10], f[11]); s.next_dir = mk3(f[12], f[13], f[14]); s.v1_pos = mk3(f[15], f[16], f[17]);
    s.last_bsdf_pdf = f[18];
    s.alive = true;
    if (r) { r->y = w[22 * cap]; r->w_sum = u2f(w[23 * cap]); r->M = w[24 * cap]; r->W = 0.0f; r->sx = r->sy = r->sz = 0.0f; r->p_hat = 0.0f; }
}

struct PathOut { f3 radiance; f3 v1_pos; };

// restir.wgsl:460-737 (VARIANT 0) / restir_spatial.wgsl:480-762 (VARIANT 1), uncut.
template <int VARIANT>
FRT_HD PathOut trace_path(PathCtx& c, uint32_t pix, uint32_t seed) {
    LoopState s;
    path_head<VARIANT>(c, pix, seed, s);
    if (s.alive) path_loop<VARIANT>(c, s, 1u, c.fv.max_depth);
    PathOut out; out.radiance = s.accumulated; out.v1_pos = s.v1_pos;
    return out;
}

// ---- stage bodies, split at trace_path so that a path can be finished by another lane -------------------------------------
FRT_HD void make_path_state(PathState& st, uint32_t pix, f3 radiance, f3 v1_pos) { st.pix = pix; st.accum = radiance; st.v1_pos = v1_pos; }

// temporal: restir.wgsl:788-918. begin -> [trace_path] -> temporal_finalize (frt_path.hpp)
FRT_HD void temporal_pixel(PathCtx& c, uint32_t px, uint32_t py) {
    const uint32_t pix = px + py * c.fv.W;
    if (c.fv.gpos[pix].w < 0.0f) { c.fv.res_temporal[pix] = zero_reservoir(); return; }   // :805-811
    PathOut path = trace_path<0>(c, pix, temporal_seed(c.fv, pix));
    PathState st;
    make_path_state(st, pix, path.radiance, path.v1_pos);
    temporal_finalize(c, st);
}

// spatial: restir_spatial.wgsl:857-1016. Neighbour loop (:912-993) -> merged reservoir r; [trace_path(r.y)] -> spatial_finalize.
// false: background pixel (outputs written).
template <class Ctx>
FRT_HD bool spatial_neighbors(Ctx& c, uint32_t pix, ReservoirView& r) {
    SpatialState ss;
    if (!spatial_begin(c, ss, pix)) return false;
    const SpatialCentre centre = spatial_centre(c.fv, pix);
    while (ss.i < ss.n) {
        AnyReq req;
        req.want = false; req.o = splat3(0.0f); req.d = splat3(0.0f); req.tmin = 0.0f; req.tmax = 0.0f;
        spatial_neighbor_prepare(c, ss, req, centre);
        bool visible = true;
        if (req.want) visible = !c.any(req.o, req.d, req.tmin, req.tmax);
        spatial_neighbor_finish(ss, visible);
    }
    r = ss.r;
    return true;
}
FRT_HD void spatial_tail(PathCtx& c, uint32_t pix, const ReservoirView& r, f3 radiance, f3 v1_pos) {
    SpatialState ss; ss.r = r; ss.pix = pix;
    PathState st;
    make_path_state(st, pix, radiance, v1_pos);
    spatial_finalize(c, ss, st);
}
FRT_HD void spatial_pixel(PathCtx& c, uint32_t px, uint32_t py) {
    const uint32_t pix = py * c.fv.W + px;
    ReservoirView r;
    if (!spatial_neighbors(c, pix, r)) return;
    PathOut fin = trace_path<1>(c, pix, r.y);
    spatial_tail(c, pix, r, fin.radiance, fin.v1_pos);
}

} // namespace frt
