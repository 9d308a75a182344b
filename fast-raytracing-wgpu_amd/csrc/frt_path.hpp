// frt_path.hpp — trace_path and the two ReSTIR stages as RESUMABLE per-lane state machines.
//
// The reference runs trace_path (restir.wgsl:460-737 / restir_spatial.wgsl:480-762) as one function per thread. On a
// 64-wide wavefront that leaves ~70 % of the lanes idle (paths end at different depths; measured in profiles/r1_v1_*).
// Here the same arithmetic, in the same order and with the same rand() sequence per pixel, is cut at its two ray queries:
//     [pre_closest] -> closest-hit trace -> [shade: material, emission, NEE set-up] -> any-hit trace -> [post_any: NEE add, BSDF sample]
// so that a wave can run lanes at different depths — and freshly regenerated lanes at depth 0 — through shared traversal
// phases (frt_kernels.hip). The spatial stage's neighbour loop issues its visibility rays through the same any-hit phase.
// Every function is __host__ __device__; tests/hostcheck drives the identical state machine one pixel at a time.
#pragma once
#include "frt_shade.hpp"

namespace frt {

struct AnyReq { bool want; f3 o, d; float tmin, tmax; };

enum : uint32_t { NEE_NONE = 0, NEE_ZERO = 1, NEE_TRACE = 2, NEE_VISIBLE = 3 };

struct PathState {
    uint32_t pix, depth;
    bool done, prev_diffuse, is_glass, front_face;
    f3 pos, ffnormal;
    float hit_t;
    f3 throughput, accum, next_dir, v1_pos;
    float last_pdf;
    // live across the any-hit phase
    f3 wo, base_color;
    MatParams m;
    // NEE sample waiting for its shadow ray: direction, geometry terms, MIS weight, emission
    f3 nee_L, nee_Le;
    float nee_gw, nee_w;   // geometry term G and MIS weight / pdf
    uint32_t nee;
};

FRT_HD void path_begin(PathCtx& c, PathState& st, uint32_t pix, uint32_t seed) {
    c.rng = seed;
    st.pix = pix; st.depth = 0u; st.done = false; st.prev_diffuse = false; st.is_glass = false; st.front_face = true;
    st.pos = splat3(0.0f); st.ffnormal = splat3(0.0f); st.hit_t = 0.0f;
    st.throughput = splat3(1.0f); st.accum = splat3(0.0f); st.next_dir = splat3(0.0f); st.v1_pos = splat3(0.0f);
    st.last_pdf = 0.0f; st.nee = NEE_NONE; st.nee_L = splat3(0.0f); st.nee_Le = splat3(0.0f); st.nee_gw = 0.0f; st.nee_w = 0.0f;
    st.wo = splat3(0.0f); st.base_color = splat3(0.0f);
    st.m.roughness = 0.0f; st.m.metallic = 0.0f; st.m.transmission = 0.0f; st.m.ior = 1.0f;
}

// depth >= 1: Russian roulette and ray set-up (restir.wgsl:593-605). false -> the path ended (st.done).
FRT_HD bool path_pre_closest(PathCtx& c, PathState& st, f3& origin) {
    if (st.depth >= 3u) {
        float p = fmaxn(st.throughput.x, fmaxn(st.throughput.y, st.throughput.z));
        float survival_prob = clampf(p, 0.05f, 0.95f);
        if (c.rand() > survival_prob) { st.done = true; return false; }
        st.throughput = st.throughput / survival_prob;
    }
    f3 offset_dir = st.ffnormal * signf(dot(st.ffnormal, st.next_dir));
    origin = st.pos + offset_dir * 0.001f;
    return true;
}

// NEE set-up shared by the primary hit and every bounce (restir.wgsl:558-571 == :707-720 + eval_direct_lighting :443-459),
// up to the shadow ray; nee_finish evaluates the BSDF only for unoccluded samples, like the reference.
template <int VARIANT>
FRT_HD void nee_prepare(PathCtx& c, PathState& st, AnyReq& req) {
    st.nee = NEE_NONE;
    uint32_t nl = c.fv.cam.num_lights;
    if (nl == 0u) return;
    uint32_t light_idx = (uint32_t)(c.rand() * (float)nl);
    if (!(light_idx < nl)) return;
    LightSmp ls = sample_light(c, light_idx);
    float pdf_nee = ls.pdf * (1.0f / (float)nl);
    float p_bsdf = eval_pdf(st.ffnormal, normalize(ls.pos - st.pos), st.wo, st.m, st.base_color);
    float mis_weight_nee = pdf_nee / (pdf_nee + p_bsdf);
    float weight = mis_weight_nee / pdf_nee;
    f3 offset_pos = st.pos + st.ffnormal * 0.001f;
    f3 L = normalize(ls.pos - offset_pos);
    float dist = distance(ls.pos, offset_pos);
    float n_dot_l = fmaxn(dot(st.ffnormal, L), 0.0f);
    float l_dot_n = fmaxn(dot(-L, ls.normal), 0.0f);
    st.nee = NEE_ZERO;
    if (n_dot_l > 0.0f && l_dot_n > 0.0f) {
        float G = (n_dot_l * l_dot_n) / (dist * dist);
        st.nee_L = L; st.nee_Le = xyz(ls.emission) * ls.emission.w; st.nee_gw = G; st.nee_w = weight;
        // trace_shadow_ray: restir.wgsl:375-381 (VARIANT 0) vs restir_spatial.wgsl:380-400 (VARIANT 1)
        float t_max = fmaxn(dist * 0.999f, 0.0f);
        float t_min = VARIANT == 0 ? 0.001f : 0.0001f;
        if (VARIANT == 1 && t_min >= t_max) { st.nee = NEE_VISIBLE; return; }
        st.nee = NEE_TRACE;
        req.want = true; req.o = offset_pos; req.d = L; req.tmin = t_min; req.tmax = t_max;
    }
}

// Shading up to the NEE shadow ray. depth 0: the hit comes from the G-buffer (restir.wgsl:475-576); depth >= 1: from the
// closest-hit query `h` fired from `origin` along st.next_dir (restir.wgsl:607-724).
template <int VARIANT>
FRT_HD void path_shade(PathCtx& c, PathState& st, const HitRec& h, f3 origin, AnyReq& req) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    bool specular;
    if (st.depth == 0u) {
        float4 pos_w = fv.gpos[st.pix];
        if (pos_w.w < 0.0f) { st.done = true; return; }
        float4 normal_w = fv.gnormal[st.pix];
        f4 albedo_raw = unpack_rgba8(fv.galbedo[st.pix]);
        st.pos = mk3(pos_w.x, pos_w.y, pos_w.z);
        st.ffnormal = decode_octahedral_normal(normal_w.x, normal_w.y);   // primary hit: treated as front-facing (:485-486)
        st.front_face = true;
        f2 uv = mk2(normal_w.z, normal_w.w);
        uint32_t mat_id = (uint32_t)(pos_w.w + 0.1f);
        f3 emissive_factor; int32_t light_index; uint32_t tex1, tex2;
        if (mat_id < sc.num_materials) {
            const MaterialView& ms = sc.materials[mat_id];
            st.m.roughness = ms.roughness; st.m.metallic = ms.metallic; st.m.transmission = ms.transmission; st.m.ior = ms.ior;
            light_index = ms.light_index; tex1 = ms.tex_info_1; tex2 = ms.tex_info_2;
            emissive_factor = mk3(ms.emissive_factor[0], ms.emissive_factor[1], ms.emissive_factor[2]);
            if (VARIANT == 0 || st.m.transmission < 0.01f) st.base_color = xyz(albedo_raw);   // restir.wgsl:494 vs restir_spatial.wgsl:514-516
            else st.base_color = mk3(ms.base_color[0], ms.base_color[1], ms.base_color[2]);
        } else {   // restir.wgsl:495-501: zero-initialised `var mat` with four fields set
            st.m.roughness = 0.0f; st.m.metallic = albedo_raw.w; st.m.transmission = 0.0f; st.m.ior = 1.0f;
            light_index = -1; tex1 = 0u; tex2 = 0u; emissive_factor = splat3(0.0f);
            st.base_color = xyz(albedo_raw);
        }
        uint32_t mr_tex_id = tex2 & 0xFFFFu;
        if (mr_tex_id != 65535u) {
            f4 mr = sample_layer<false>(sc, mr_tex_id, uv);
            st.m.metallic = mr.z * st.m.metallic;
            st.m.roughness = mr.y * st.m.roughness;
        }
        st.wo = normalize(mk3(fv.cam.view_pos[0], fv.cam.view_pos[1], fv.cam.view_pos[2]) - st.pos);
        uint32_t emissive_tex_id = tex1 >> 16u;
        if (mat_id < sc.num_materials && light_index == -1) {   // :523-533
            f3 emission = emissive_factor;
            if (emissive_tex_id != 65535u) emission = emission * xyz(sample_layer<true>(sc, emissive_tex_id, uv));
            st.accum = st.accum + emission;
        }
        if (light_index >= 0) {   // :543-552
            f3 emission = emissive_factor;
            if (emissive_tex_id != 65535u) emission = emission * xyz(sample_layer<true>(sc, emissive_tex_id, uv));
            st.accum = st.accum + emission;
            st.done = true;
            return;
        }
        st.is_glass = st.m.transmission > 0.01f;
        specular = st.is_glass || st.m.roughness < 0.05f;   // :556
    } else {
        if (h.tri == 0xFFFFFFFFu) { st.done = true; return; }   // :609-611
        HitGeom g = fetch_hit_geometry(sc, h);                   // reconstruct_geometry_hit, :383-441
        st.front_face = h.front;
        st.ffnormal = h.front ? g.normal_w : -g.normal_w;
        st.hit_t = h.t;
        st.pos = origin + st.next_dir * h.t;
        if (st.depth == 1u) st.v1_pos = st.pos;                  // :625-629
        st.wo = -st.next_dir;
        const MaterialView& mb = sc.materials[g.mat_id];
        st.m.roughness = mb.roughness; st.m.metallic = mb.metallic; st.m.transmission = mb.transmission; st.m.ior = mb.ior;
        int32_t light_index = mb.light_index;
        uint32_t t0i = mb.tex_info_0, t1i = mb.tex_info_1;
        f4 tex_color = mk4(1.0f, 1.0f, 1.0f, 1.0f);
        uint32_t tex_id = t0i & 0xFFFFu, normal_tex_id = t0i >> 16u;
        if (tex_id != 65535u) tex_color = sample_layer<true>(sc, tex_id, g.uv);
        float occlusion = 1.0f;
        uint32_t occlusion_tex_id = t1i & 0xFFFFu, emissive_tex_id = t1i >> 16u;
        if (occlusion_tex_id != 65535u) occlusion = sample_layer<false>(sc, occlusion_tex_id, g.uv).x;
        st.base_color = mk3(mb.base_color[0], mb.base_color[1], mb.base_color[2]) * xyz(tex_color) * occlusion;
        if (normal_tex_id != 65535u) {   // :657-671
            f3 nm = xyz(sample_layer<false>(sc, normal_tex_id, g.uv));
            f4 tg = hit_tangent(sc, g);
            st.ffnormal = perturb_normal(st.ffnormal, xyz(tg), tg.w, nm);
        }
        if (light_index == -1 && emissive_tex_id != 65535u) {   // :675-678
            f3 emissive_col = xyz(sample_layer<true>(sc, emissive_tex_id, g.uv));
            st.accum = st.accum + emissive_col * st.throughput;
        }
        if (light_index >= 0) {   // :683-700
            if (st.front_face) {
                const LightView& light = sc.lights[light_index];
                f3 Le = mk3(light.emission[0], light.emission[1], light.emission[2]) * light.emission[3];
                float mis_weight = 1.0f;
                if (st.prev_diffuse) {
                    float dist_sq = st.hit_t * st.hit_t;
                    float light_cos = fmaxn(dot(st.ffnormal, -st.wo), 0.0f);
                    float p_bsdf = st.last_pdf;
                    float p_nee = (1.0f / light.area) * (dist_sq / light_cos) * (1.0f / (float)fv.cam.num_lights);
                    if (light_cos > 0.001f) mis_weight = p_bsdf / (p_bsdf + p_nee);
                    else mis_weight = 0.0f;
                }
                st.accum = st.accum + Le * st.throughput * mis_weight;
            }
            st.done = true;
            return;
        }
        specular = st.is_glass || st.m.roughness < 0.05f;   // :705 — the PRIMARY hit's is_glass (reference quirk)
    }
    st.nee = NEE_NONE;
    if (!specular) {
        nee_prepare<VARIANT>(c, st, req);
        st.prev_diffuse = true;
    } else st.prev_diffuse = false;
}

// After the shadow ray: add the NEE term, sample the BSDF, advance (restir.wgsl:569, :577-584 and :718, :727-732).
FRT_HD void path_post_any(PathCtx& c, PathState& st, bool visible) {
    if (st.nee != NEE_NONE) {
        bool lit = (st.nee == NEE_VISIBLE) || (st.nee == NEE_TRACE && visible);
        f3 direct = splat3(0.0f);
        if (lit) {   // eval_direct_lighting :453-455: Le * f * G * weight
            f3 f = eval_bsdf(st.ffnormal, st.nee_L, st.wo, st.m, st.base_color);
            direct = st.nee_Le * f * st.nee_gw * st.nee_w;
        }
        st.accum = st.accum + direct * st.throughput;
    }
    BsdfSmp s = sample_bsdf(c, st.wo, st.ffnormal, st.front_face, st.m, st.base_color);
    if (s.weight.x <= 0.0f && s.weight.y <= 0.0f && s.weight.z <= 0.0f) { st.done = true; return; }
    st.last_pdf = s.pdf;
    st.throughput = st.throughput * s.weight;
    st.next_dir = s.wi;
    st.depth += 1u;
    if (!(st.depth < c.fv.max_depth)) st.done = true;   // for (depth = 1; depth < MAX_DEPTH; depth++), :590
}

FRT_HD void update_reservoir(ReservoirView& r, uint32_t seed_cand, float w, float rnd, uint32_t cnt, float p_hat_new, f3 s_path_new) {   // :746-756
    r.w_sum += w;
    r.M += cnt;
    if (rnd * r.w_sum < w) { r.y = seed_cand; r.p_hat = p_hat_new; r.sx = s_path_new.x; r.sy = s_path_new.y; r.sz = s_path_new.z; }
}

// ================================================================================================ stage 1: restir.wgsl:788-918
FRT_HD bool is_valid_neighbor_temporal(f3 cp, f3 cn, uint32_t cm, f3 pp, f3 pn, uint32_t pm, f3 cam) {   // :758-778
    if (cm != pm) return false;
    if (dot(cn, pn) < 0.99f) return false;
    float dist_diff_sq = dot(cp - pp, cp - pp);
    float dist_to_camera_sq = dot(cp - cam, cp - cam);
    float threshold = fmaxn(0.00001f, dist_to_camera_sq * 0.001f);
    return !(dist_diff_sq > threshold);
}
FRT_HD uint32_t temporal_seed(const FrameView& fv, uint32_t pixel_idx) { return pcg_hash(pixel_idx + fv.cam.frame_count * 927163u); }   // :797-798

// false: background pixel, zero reservoir written, nothing to trace.
FRT_HD bool temporal_begin(PathCtx& c, PathState& st, uint32_t pixel_idx) {
    const FrameView& fv = c.fv;
    if (fv.gpos[pixel_idx].w < 0.0f) { fv.res_temporal[pixel_idx] = zero_reservoir(); return false; }   // :805-811
    path_begin(c, st, pixel_idx, temporal_seed(fv, pixel_idx));
    return true;
}
// The temporal stage is split at the only point where it touches the previous frame ("T-trace / T-merge"):
//   T-trace  = trace_path of the fresh candidate (restir.wgsl:797-825): a function of this frame's G-buffer and (pixel, frame) seed only;
//              its whole result is the candidate record (v1_pos, p_hat = luminance(radiance)), 16 bytes per pixel;
//   T-merge  = RIS with that candidate, reprojection, merge with the previous frame's SPATIAL reservoir, store (:826-917): no rays.
// T-trace(f+1) can therefore run while spatial(f) is still in flight; only T-merge sits on the frame-to-frame dependency chain.
FRT_HD float4 temporal_candidate(f3 radiance, f3 v1_pos) { return make_float4(v1_pos.x, v1_pos.y, v1_pos.z, luminance(radiance)); }

// RIS with the fresh candidate, temporal merge with the previous frame's spatial reservoir, store (:826-917).
FRT_HD void temporal_merge(const SceneView& sc, const FrameView& fv, uint32_t pixel_idx, float4 cand) {
    uint32_t px = pixel_idx % fv.W, py = pixel_idx / fv.W;
    uint32_t seed_base = pixel_idx + fv.cam.frame_count * 927163u;
    uint32_t seed_candidate = pcg_hash(seed_base);
    uint32_t local_seed = seed_base;
    float4 pos_w = fv.gpos[pixel_idx];
    ReservoirView r = zero_reservoir();
    float p_hat = cand.w;
    update_reservoir(r, seed_candidate, p_hat, 0.5f, 1u, p_hat, mk3(cand.x, cand.y, cand.z));
    r.W = p_hat > 0.0f ? 1.0f : 0.0f;
    float2 motion = fv.gmotion[pixel_idx];
    f2 size = mk2((float)fv.W, (float)fv.H);
    f2 uv = (mk2((float)px, (float)py) + mk2(0.5f, 0.5f)) / size;
    f2 prev_uv = uv + mk2(motion.x, motion.y);
    if (prev_uv.x >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y >= 0.0f && prev_uv.y <= 1.0f) {
        f2 pf = prev_uv * size;
        uint32_t qx = (uint32_t)pf.x, qy = (uint32_t)pf.y;
        bool inb = qx < fv.W && qy < fv.H;          // prev_uv == 1.0: out-of-range texel reads give zeros
        uint32_t prev_idx = inb ? qy * fv.W + qx : 0u;
        if (inb && (qy < fv.prev_y0 || qy >= fv.prev_y1)) note_halo_overflow(fv);
        float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float4 prev_pos = inb ? fv.gpos_prev[prev_idx] : zero4;
        float4 prev_nrm = inb ? fv.gnormal_prev[prev_idx] : zero4;
        f3 prev_normal = decode_octahedral_normal(prev_nrm.x, prev_nrm.y);
        uint32_t prev_mat_id = (uint32_t)(prev_pos.w + 0.1f);
        float4 cur_nrm = fv.gnormal[pixel_idx];
        f3 curr_normal = decode_octahedral_normal(cur_nrm.x, cur_nrm.y);
        uint32_t curr_mat_id = (uint32_t)(pos_w.w + 0.1f);
        const MaterialView& mat = sc.materials[curr_mat_id];
        bool is_specular = mat.roughness < 0.2f || mat.metallic > 0.8f || (mat.transmission > 0.01f);
        f3 cam = mk3(fv.cam.view_pos[0], fv.cam.view_pos[1], fv.cam.view_pos[2]);
        if (is_valid_neighbor_temporal(mk3(pos_w.x, pos_w.y, pos_w.z), curr_normal, curr_mat_id,
                                       mk3(prev_pos.x, prev_pos.y, prev_pos.z), prev_normal, prev_mat_id, cam) && !is_specular) {
            ReservoirView prev_r = inb ? fv.res_spatial[prev_idx] : zero_reservoir();
            f3 curr_albedo = xyz(unpack_rgba8(fv.galbedo[pixel_idx]));
            f3 prev_albedo = inb ? xyz(unpack_rgba8(fv.galbedo_prev[prev_idx])) : splat3(0.0f);
            float l_curr = luminance(curr_albedo) + 0.001f;
            float l_prev = luminance(prev_albedo) + 0.001f;
            float albedo_ratio = l_curr / l_prev;
            if (albedo_ratio < 3.0f && albedo_ratio > 0.33f) {
                float p_hat_new = prev_r.p_hat * albedo_ratio;
                if (p_hat_new > 0.0f) {
                    uint32_t clamped_M = prev_r.M < 16u ? prev_r.M : 16u;   // MAX_RESERVOIR_M_TEMPORAL, :851
                    float w_prev = p_hat_new * prev_r.W * (float)clamped_M;
                    update_reservoir(r, prev_r.y, w_prev, rand_lcg(local_seed), clamped_M, p_hat_new, mk3(prev_r.sx, prev_r.sy, prev_r.sz));
                }
            }
        }
    }
    float p_hat_final = r.p_hat;
    if (p_hat_final > 0.0f) r.W = (1.0f / p_hat_final) * (r.w_sum / (float)r.M);
    else { r.W = 0.0f; r.p_hat = 0.0f; }
    fv.res_temporal[pixel_idx] = r;
}
// One pixel of the T-merge pass (also the background case of restir.wgsl:805-811).
FRT_HD void temporal_merge_pixel(const SceneView& sc, const FrameView& fv, uint32_t pixel_idx) {
    if (fv.gpos[pixel_idx].w < 0.0f) { fv.res_temporal[pixel_idx] = zero_reservoir(); return; }
    temporal_merge(sc, fv, pixel_idx, fv.cand[pixel_idx]);
}
// Fused form (the reference's order: trace, then merge, in one invocation).
FRT_HD void temporal_finalize(PathCtx& c, const PathState& st) { temporal_merge(c.sc, c.fv, st.pix, temporal_candidate(st.accum, st.v1_pos)); }

// ================================================================================================ stage 2: restir_spatial.wgsl:857-1016
FRT_HD bool is_valid_neighbor_spatial(const SceneView& sc, f3 cp, f3 cn, uint32_t cm, f3 pp, f3 pn, uint32_t pm, f3 cam) {   // :783-814
    if (cm != pm) return false;
    const MaterialView& mat = sc.materials[cm];
    bool is_specular = mat.roughness < 0.2f || mat.metallic > 0.8f || (mat.transmission > 0.01f);
    if (is_specular) {
        if (dot(cn, pn) < 0.998f) return false;
        if (distance(cp, pp) > 0.01f) return false;
    } else {
        if (dot(cn, pn) < 0.995f) return false;
        float dist_to_camera_sq = dot(cp - cam, cp - cam);
        float threshold = fmaxn(0.00001f, dist_to_camera_sq * 0.001f);
        float dist_diff_sq = dot(cp - pp, cp - pp);
        if (dist_diff_sq > threshold) return false;
    }
    return true;
}
FRT_HD float calculate_jacobian(f3 curr_pos, f3 curr_normal, f3 curr_albedo, f3 n_v1, f3 n_pos, f3 n_normal, f3 n_albedo) {   // :822-854
    float cos_curr = fmaxn(dot(curr_normal, normalize(n_v1 - curr_pos)), 0.0f);
    float cos_neigh = fmaxn(dot(n_normal, normalize(n_v1 - n_pos)), 0.0f);
    if (cos_neigh <= 0.001f) return 0.0f;
    float jacobian = cos_curr / cos_neigh;
    float lum_curr = luminance(curr_albedo) + 0.001f;
    float lum_neigh = luminance(n_albedo) + 0.001f;
    jacobian = jacobian * (lum_curr / lum_neigh);
    return clampf(jacobian, 0.1f, 10.0f);
}

struct SpatialState {
    ReservoirView r;
    uint32_t pix, local_seed, i, n;
    bool narrow, pending;
    // the candidate waiting for its visibility ray
    uint32_t cand_y, cand_M;
    float cand_weight, cand_p_hat;
    f3 cand_s_path;
};

// false: background pixel (zero reservoir + zero radiance written).
FRT_HD bool spatial_begin(PathCtx& c, SpatialState& ss, uint32_t pixel_idx) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    float4 pos_w4 = fv.gpos[pixel_idx];
    if (pos_w4.w < 0.0f) {   // :874-884
        fv.res_spatial[pixel_idx] = zero_reservoir();
        fv.raw[pixel_idx] = pack_rgba16f(mk4(0.0f, 0.0f, 0.0f, 0.0f));
        return false;
    }
    ss.pix = pixel_idx;
    ss.local_seed = pixel_idx + fv.frame_count * 0x12345678u;   // seed_init, :866-870 (scene_info.y == frame_count)
    ss.r = fv.res_temporal[pixel_idx];
    if (ss.r.M > 20u) { ss.r.w_sum *= 20.0f / (float)ss.r.M; ss.r.M = 20u; }
    const MaterialView& mat = sc.materials[(uint32_t)(pos_w4.w + 0.1f)];
    ss.narrow = mat.roughness < 0.1f || mat.metallic > 0.9f || mat.transmission > 0.1f;   // :906 and :957
    ss.n = ss.narrow ? 3u : 5u;
    ss.i = 0u;
    ss.pending = false;
    return true;
}
// One iteration of the neighbour loop up to its visibility ray (:912-982). Sets ss.pending when a candidate awaits the ray
// result (req.want says whether a ray is actually needed).
// The centre pixel's own G-buffer values: the same for every neighbour, decoded once per pixel by the kernels.
struct SpatialCentre { f3 pos_w, normal, albedo, camera_pos; uint32_t mat_id; };
FRT_HD SpatialCentre spatial_centre(const FrameView& fv, uint32_t pix) {
    SpatialCentre k;
    float4 pos_w4 = fv.gpos[pix];
    k.pos_w = mk3(pos_w4.x, pos_w4.y, pos_w4.z);
    float4 normal_w = fv.gnormal[pix];
    k.normal = decode_octahedral_normal(normal_w.x, normal_w.y);
    k.mat_id = (uint32_t)(pos_w4.w + 0.1f);
    k.albedo = xyz(unpack_rgba8(fv.galbedo[pix]));
    k.camera_pos = mk3(fv.cam.view_pos[0], fv.cam.view_pos[1], fv.cam.view_pos[2]);
    return k;
}
template <class Ctx>
FRT_HD void spatial_neighbor_prepare(Ctx& c, SpatialState& ss, AnyReq& req, const SpatialCentre& k) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    ss.pending = false;
    uint32_t px = ss.pix % fv.W, py = ss.pix / fv.W;
    float radius = ss.narrow ? 4.0f : 10.0f;
    float r1 = rand_lcg(ss.local_seed);
    float r2 = rand_lcg(ss.local_seed);
    float angle = 2.0f * kPI * r1;
    float rad = sqrtf_(r2) * radius;
    float sa, ca;
    sincosf_(angle, sa, ca);
    f2 offset = mk2(ca, sa) * rad;
    int nx = (int)px + (int)offset.x, ny = (int)py + (int)offset.y;   // vec2<i32>(offset) truncates toward zero
    if (nx < 0 || nx >= (int)fv.W || ny < 0 || ny >= (int)fv.H) return;
    uint32_t nidx = (uint32_t)ny * fv.W + (uint32_t)nx;
    float4 n_pos4 = fv.gpos[nidx];
    if (n_pos4.w < 0.0f) return;
    const f3 pos_w = k.pos_w, normal = k.normal, albedo = k.albedo, camera_pos = k.camera_pos;
    const uint32_t mat_id = k.mat_id;
    f3 n_pos = mk3(n_pos4.x, n_pos4.y, n_pos4.z);
    float4 n_nrm = fv.gnormal[nidx];
    f3 n_normal = decode_octahedral_normal(n_nrm.x, n_nrm.y);
    uint32_t n_mat_id = (uint32_t)(n_pos4.w + 0.1f);
    f3 n_albedo = xyz(unpack_rgba8(fv.galbedo[nidx]));
    if (!is_valid_neighbor_spatial(sc, pos_w, normal, mat_id, n_pos, n_normal, n_mat_id, camera_pos)) return;
    ReservoirView nr = fv.res_temporal[nidx];
    if (nr.p_hat <= 0.0f) return;
    f3 n_s_path = mk3(nr.sx, nr.sy, nr.sz);
    float jacobian = calculate_jacobian(pos_w, normal, albedo, n_s_path, n_pos, n_normal, n_albedo);
    if (ss.narrow) { if (jacobian < 0.5f || jacobian > 2.0f) return; }
    f3 dir_to_v1 = n_s_path - pos_w;
    float dist_to_v1 = length(dir_to_v1);
    if (!(dot(normal, dir_to_v1) > 0.0f)) return;
    if (!(dist_to_v1 > 0.001f)) return;
    f3 ray_dir = normalize(dir_to_v1);
    float dist = fmaxn(dist_to_v1, 0.0f);
    // trace_shadow_ray (spatial variant, restir_spatial.wgsl:380-400)
    float t_max = fmaxn(dist * 0.999f, 0.0f);
    float t_min = 0.0001f;
    ss.pending = true;
    ss.cand_p_hat = nr.p_hat * jacobian;
    ss.cand_M = nr.M < 20u ? nr.M : 20u;
    ss.cand_weight = ss.cand_p_hat * nr.W * (float)ss.cand_M;
    ss.cand_y = nr.y;
    ss.cand_s_path = n_s_path;
    if (t_min >= t_max) return;   // "too close": treated as unoccluded without a ray
    req.want = true; req.o = pos_w; req.d = ray_dir; req.tmin = t_min; req.tmax = t_max;
}
FRT_HD void spatial_neighbor_prepare(PathCtx& c, SpatialState& ss, AnyReq& req) {
    spatial_neighbor_prepare(c, ss, req, spatial_centre(c.fv, ss.pix));
}
FRT_HD void spatial_neighbor_finish(SpatialState& ss, bool visible) {
    if (ss.pending && visible)
        update_reservoir(ss.r, ss.cand_y, ss.cand_weight, rand_lcg(ss.local_seed), ss.cand_M, ss.cand_p_hat, ss.cand_s_path);   // :992
    ss.pending = false;
    ss.i += 1u;
}
// Re-traced path finished: clamp W, write reservoir and radiance (:996-1015).
FRT_HD void spatial_finalize(PathCtx& c, SpatialState& ss, const PathState& st) {
    const FrameView& fv = c.fv;
    ReservoirView r = ss.r;
    f3 final_color = splat3(0.0f);
    float p_hat_final = luminance(st.accum);
    r.sx = st.v1_pos.x; r.sy = st.v1_pos.y; r.sz = st.v1_pos.z;
    if (p_hat_final > 0.0f) {
        float w_unclamped = (1.0f / p_hat_final) * (r.w_sum / (float)r.M);
        r.W = clampf(w_unclamped, 0.0f, 20.0f);
        final_color = st.accum * r.W;
        r.p_hat = p_hat_final;
    } else { r.W = 0.0f; r.p_hat = 0.0f; }
    fv.res_spatial[ss.pix] = r;
    fv.raw[ss.pix] = pack_rgba16f(mk4(final_color, 1.0f));
}

// ================================================================================================ sequential drivers
// One pixel at a time through the same state machine (tests/hostcheck; also documents the control flow of the kernels).
template <int VARIANT>
FRT_HD void run_path(PathCtx& c, PathState& st) {
    while (!st.done) {
        HitRec h; h.tri = 0xFFFFFFFFu; h.t = 0.0f; h.u = h.v = 0.0f; h.inst = 0u; h.front = false;
        f3 origin = splat3(0.0f);
        if (st.depth >= 1u) {
            if (!path_pre_closest(c, st, origin)) break;
            c.n_closest++;
            trace<false>(c.sc, origin, st.next_dir, 0.001f, 100.0f, c.stk, c.stride, h);
        }
        AnyReq req; req.want = false;
        path_shade<VARIANT>(c, st, h, origin, req);
        if (st.done) break;
        bool visible = true;
        if (req.want) {
            HitRec s;
            c.n_any++;
            trace<true>(c.sc, req.o, req.d, req.tmin, req.tmax, c.stk, c.stride, s);
            visible = s.tri == 0xFFFFFFFFu;
        }
        path_post_any(c, st, visible);
    }
}
FRT_HD void temporal_pixel_sm(PathCtx& c, uint32_t px, uint32_t py) {
    PathState st;
    if (!temporal_begin(c, st, px + py * c.fv.W)) return;
    run_path<0>(c, st);
    temporal_finalize(c, st);
}
FRT_HD void spatial_pixel_sm(PathCtx& c, uint32_t px, uint32_t py) {
    SpatialState ss;
    if (!spatial_begin(c, ss, py * c.fv.W + px)) return;
    while (ss.i < ss.n) {
        AnyReq req; req.want = false;
        spatial_neighbor_prepare(c, ss, req);
        bool visible = true;
        if (req.want) {
            HitRec s;
            c.n_any++;
            trace<true>(c.sc, req.o, req.d, req.tmin, req.tmax, c.stk, c.stride, s);
            visible = s.tri == 0xFFFFFFFFu;
        }
        spatial_neighbor_finish(ss, visible);
    }
    PathState st;
    path_begin(c, st, ss.pix, ss.r.y);
    run_path<1>(c, st);
    spatial_finalize(c, ss, st);
}

} // namespace frt
