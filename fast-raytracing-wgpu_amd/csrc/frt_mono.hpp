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
template <int VARIANT>
FRT_HD bool trace_shadow_ray(PathCtx& c, f3 origin, f3 dir, float dist) {
    float t_max = fmaxn(dist * 0.999f, 0.0f);
    float t_min = VARIANT == 0 ? 0.001f : 0.0001f;
    if (VARIANT == 1 && t_min >= t_max) return true;
    HitRec h;
    c.n_any++;
    trace<true>(c.sc, origin, dir, t_min, t_max, c.stk, c.stride, h);
    return h.tri == 0xFFFFFFFFu;
}

template <int VARIANT>
FRT_HD f3 eval_direct_lighting(PathCtx& c, const Surf& hit, f3 wo, const MatParams& m, f3 base_color, const LightSmp& ls, float weight) {   // :443-459
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
template <int VARIANT>
FRT_HD f3 nee(PathCtx& c, const Surf& hit, f3 wo, const MatParams& m, f3 base_color, f3 throughput) {   // :558-571 == :707-720
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

struct PathOut { f3 radiance; f3 v1_pos; };

// restir.wgsl:460-737 (VARIANT 0) / restir_spatial.wgsl:480-762 (VARIANT 1)
template <int VARIANT>
FRT_HD PathOut trace_path(PathCtx& c, uint32_t pix, uint32_t seed) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    c.rng = seed;
    PathOut out; out.radiance = splat3(0.0f); out.v1_pos = splat3(0.0f);
    float4 pos_w = fv.gpos[pix];
    if (pos_w.w < 0.0f) return out;
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
        out.radiance = accumulated;
        return out;
    }
    const bool is_glass = m.transmission > 0.01f;
    bool previous_was_diffuse;
    if (!(is_glass || m.roughness < 0.05f)) {   // :556
        accumulated = accumulated + nee<VARIANT>(c, hit, wo, m, base_color, throughput);
        previous_was_diffuse = true;
    } else previous_was_diffuse = false;

    BsdfSmp sc0 = sample_bsdf(c, wo, hit.ffnormal, hit.front_face, m, base_color);
    if (sc0.weight.x <= 0.0f && sc0.weight.y <= 0.0f && sc0.weight.z <= 0.0f) { out.radiance = accumulated; return out; }
    float last_bsdf_pdf = sc0.pdf;
    throughput = throughput * sc0.weight;
    f3 next_dir = sc0.wi;

    for (uint32_t depth = 1u; depth < fv.max_depth; depth++) {   // :590
        if (depth >= 3u) {
            float p = fmaxn(throughput.x, fmaxn(throughput.y, throughput.z));
            float survival_prob = clampf(p, 0.05f, 0.95f);
            if (c.rand() > survival_prob) break;
            throughput = throughput / survival_prob;
        }
        f3 offset_dir = hit.ffnormal * signf(dot(hit.ffnormal, next_dir));
        f3 origin = hit.pos + offset_dir * 0.001f;
        HitRec h;
        c.n_closest++;
        trace<false>(sc, origin, next_dir, 0.001f, 100.0f, c.stk, c.stride, h);
        if (h.tri == 0xFFFFFFFFu) break;
        HitGeom g = fetch_hit_geometry(sc, h);   // reconstruct_geometry_hit, :383-441
        hit.normal = g.normal_w;
        hit.tangent = mk4(g.tangent_w, g.tangent_sign);
        hit.uv = g.uv;
        hit.front_face = h.front;
        hit.ffnormal = h.front ? g.normal_w : -g.normal_w;
        hit.t = h.t;
        hit.pos = origin + next_dir * h.t;
        hit.mat_id = g.mat_id;
        if (depth == 1u) out.v1_pos = hit.pos;
        wo = -next_dir;
        const MaterialView& mb = sc.materials[hit.mat_id];
        m.roughness = mb.roughness; m.metallic = mb.metallic; m.transmission = mb.transmission; m.ior = mb.ior;
        int32_t light_index_b = mb.light_index;
        uint32_t t0i = mb.tex_info_0, t1i = mb.tex_info_1;
        f4 tex_color = mk4(1.0f, 1.0f, 1.0f, 1.0f);
        uint32_t tex_id = t0i & 0xFFFFu, normal_tex_id = t0i >> 16u;
        if (tex_id != 65535u) tex_color = sample_layer<true>(sc, tex_id, hit.uv);
        float occlusion = 1.0f;
        uint32_t occlusion_tex_id = t1i & 0xFFFFu, emissive_tex_id_b = t1i >> 16u;
        if (occlusion_tex_id != 65535u) occlusion = sample_layer<false>(sc, occlusion_tex_id, hit.uv).x;
        base_color = mk3(mb.base_color[0], mb.base_color[1], mb.base_color[2]) * xyz(tex_color) * occlusion;
        if (normal_tex_id != 65535u) {
            f3 nm = xyz(sample_layer<false>(sc, normal_tex_id, hit.uv));
            hit.ffnormal = perturb_normal(hit.ffnormal, xyz(hit.tangent), hit.tangent.w, nm);
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
    }
    out.radiance = accumulated;
    return out;
}

// ================================================================================================ stage 1: restir.wgsl:788-918
FRT_HD void temporal_pixel(PathCtx& c, uint32_t px, uint32_t py) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    uint32_t pixel_idx = px + py * fv.W;
    uint32_t seed_base = pixel_idx + fv.cam.frame_count * 927163u;
    uint32_t seed_candidate = pcg_hash(seed_base);
    uint32_t local_seed = seed_base;
    float4 pos_w = fv.gpos[pixel_idx];
    if (pos_w.w < 0.0f) { fv.res_temporal[pixel_idx] = zero_reservoir(); return; }
    ReservoirView r = zero_reservoir();
    PathOut path = trace_path<0>(c, pixel_idx, seed_candidate);
    float p_hat = luminance(path.radiance);
    update_reservoir(r, seed_candidate, p_hat, 0.5f, 1u, p_hat, path.v1_pos);
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

// ================================================================================================ stage 2: restir_spatial.wgsl:857-1016
FRT_HD void spatial_pixel(PathCtx& c, uint32_t px, uint32_t py) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    uint32_t pixel_idx = py * fv.W + px;
    uint32_t seed_init = py * fv.W + px + fv.frame_count * 0x12345678u;   // scene_info.y (restir_spatial.rs execute)
    uint32_t local_seed = seed_init;
    float4 pos_w4 = fv.gpos[pixel_idx];
    if (pos_w4.w < 0.0f) {
        fv.res_spatial[pixel_idx] = zero_reservoir();
        fv.raw[pixel_idx] = pack_rgba16f(mk4(0.0f, 0.0f, 0.0f, 0.0f));
        return;
    }
    f3 pos_w = mk3(pos_w4.x, pos_w4.y, pos_w4.z);
    float4 normal_w = fv.gnormal[pixel_idx];
    f3 normal = decode_octahedral_normal(normal_w.x, normal_w.y);
    uint32_t mat_id = (uint32_t)(pos_w4.w + 0.1f);
    f3 albedo = xyz(unpack_rgba8(fv.galbedo[pixel_idx]));
    ReservoirView r = fv.res_temporal[pixel_idx];
    if (r.M > 20u) { r.w_sum *= 20.0f / (float)r.M; r.M = 20u; }
    f3 camera_pos = mk3(fv.cam.view_pos[0], fv.cam.view_pos[1], fv.cam.view_pos[2]);
    const MaterialView& mat = sc.materials[mat_id];
    const bool narrow = mat.roughness < 0.1f || mat.metallic > 0.9f || mat.transmission > 0.1f;   // :906 and :957
    uint32_t num_neighbors = narrow ? 3u : 5u;
    float radius = narrow ? 4.0f : 10.0f;
    for (uint32_t i = 0u; i < num_neighbors; i++) {
        float r1 = rand_lcg(local_seed);
        float r2 = rand_lcg(local_seed);
        float angle = 2.0f * kPI * r1;
        float rad = sqrtf_(r2) * radius;
        float sa, ca;
        sincosf_(angle, sa, ca);
        f2 offset = mk2(ca, sa) * rad;
        int nx = (int)px + (int)offset.x, ny = (int)py + (int)offset.y;   // vec2<i32>(offset) truncates toward zero
        if (nx < 0 || nx >= (int)fv.W || ny < 0 || ny >= (int)fv.H) continue;
        uint32_t nidx = (uint32_t)ny * fv.W + (uint32_t)nx;
        float4 n_pos4 = fv.gpos[nidx];
        if (n_pos4.w < 0.0f) continue;
        f3 n_pos = mk3(n_pos4.x, n_pos4.y, n_pos4.z);
        float4 n_nrm = fv.gnormal[nidx];
        f3 n_normal = decode_octahedral_normal(n_nrm.x, n_nrm.y);
        uint32_t n_mat_id = (uint32_t)(n_pos4.w + 0.1f);
        f3 n_albedo = xyz(unpack_rgba8(fv.galbedo[nidx]));
        if (!is_valid_neighbor_spatial(sc, pos_w, normal, mat_id, n_pos, n_normal, n_mat_id, camera_pos)) continue;
        ReservoirView nr = fv.res_temporal[nidx];
        if (nr.p_hat <= 0.0f) continue;
        f3 n_s_path = mk3(nr.sx, nr.sy, nr.sz);
        float jacobian = calculate_jacobian(pos_w, normal, albedo, n_s_path, n_pos, n_normal, n_albedo);
        if (narrow) { if (jacobian < 0.5f || jacobian > 2.0f) continue; }
        f3 dir_to_v1 = n_s_path - pos_w;
        float dist_to_v1 = length(dir_to_v1);
        bool visible = false;
        if (dot(normal, dir_to_v1) > 0.0f) {
            if (dist_to_v1 > 0.001f) {
                f3 ray_dir = normalize(dir_to_v1);
                float t_max = fmaxn(dist_to_v1, 0.0f);
                if (trace_shadow_ray<1>(c, pos_w, ray_dir, t_max)) visible = true;
            }
        }
        if (!visible) continue;
        float p_hat_corrected = nr.p_hat * jacobian;
        uint32_t M_new = nr.M < 20u ? nr.M : 20u;
        float weight = p_hat_corrected * nr.W * (float)M_new;
        update_reservoir(r, nr.y, weight, rand_lcg(local_seed), M_new, p_hat_corrected, n_s_path);
    }
    PathOut fin = trace_path<1>(c, pixel_idx, r.y);
    f3 final_color = splat3(0.0f);
    float p_hat_final = luminance(fin.radiance);
    r.sx = fin.v1_pos.x; r.sy = fin.v1_pos.y; r.sz = fin.v1_pos.z;
    if (p_hat_final > 0.0f) {
        float w_unclamped = (1.0f / p_hat_final) * (r.w_sum / (float)r.M);
        r.W = clampf(w_unclamped, 0.0f, 20.0f);
        final_color = fin.radiance * r.W;
        r.p_hat = p_hat_final;
    } else { r.W = 0.0f; r.p_hat = 0.0f; }
    fv.res_spatial[pixel_idx] = r;
    fv.raw[pixel_idx] = pack_rgba16f(mk4(final_color, 1.0f));
}


} // namespace frt
