"""A physical pin for the estimator, independent of the oracle's own arithmetic: what next-event estimation + BSDF sampling under
multiple importance sampling (restir.wgsl:443-459, :558-571, :683-700) must add up to, worked out from the rendering equation and
evaluated by quadrature in double precision with BSDF and pdf restated here from the shader text (restir.wgsl:170-200, :249-305).
Scene: a large diffuse-glossy floor under the Cornell Box's quad light, nothing else; MAX_DEPTH = 2, so a candidate path is exactly
{light sample at the primary hit, weighted w_nee} + {BSDF-sampled ray that may hit the light, weighted w_bsdf}.
The direct-lighting integral is
        L(x, wo) = integral over the light quad of  Le * f(x, wi, wo) * cos(theta_x) * cos(theta_l) / r^2  dA.
The reference does NOT estimate all of it, and the test pins exactly what it does estimate (a property of the reference, kept on purpose):
  * a BSDF-sampled ray that hits the light after a diffuse bounce gets weight 0: restir.wgsl:690 computes light_cos = max(dot(ffnormal, -wo), 0)
    with wo = -ray direction and ffnormal turned against the ray, which is never positive, and :693-695 then set mis_weight = 0;
  * the light sample's weight w_nee = pdf_nee / (pdf_nee + p_bsdf) compares an area-measure density (1 / area / num_lights, :563) with a
    solid-angle density (eval_pdf, :564).
So E[candidate] = integral of Le * f * G * w_nee dA, a few per cent below L. The candidate's target value p_hat = luminance(radiance) of the
temporal stage (frame 0, no history: the reservoir holds the fresh candidate, restir.wgsl:826-840) is averaged over many seeds and compared
with that integral (must agree within Monte-Carlo error) and with the full L (must be below it by the predicted amount)."""
import numpy as np

PI = 3.14159265359


def _bsdf(n, wi, wo, base, rough, metal):
    """eval_bsdf, restir.wgsl:278-305, in float64 (arrays of directions)."""
    ndl = (wi * n).sum(-1); ndv = (wo * n).sum(-1)
    h = wi + wo; h /= np.linalg.norm(h, axis=-1, keepdims=True)
    ndh = np.maximum((h * n).sum(-1), 0.0); hdv = np.maximum((h * wo).sum(-1), 0.0)
    f0 = 0.04 * (1 - metal) + base * metal
    a = rough * rough; a2 = a * a
    d = a2 / (PI * (ndh * ndh * (a2 - 1) + 1) ** 2)                      # ndf_ggx :182
    g1 = lambda x: 2 * x / (x + np.sqrt(rough * rough + (1 - rough * rough) * x * x))   # geometry_schlick_ggx :189 (a2 = roughness^2 there)
    g = g1(ndl) * g1(ndv)
    fr = f0 + (1 - f0) * np.clip(1 - hdv, 0, 1) ** 5                     # fresnel_schlick :170
    spec = d * g * fr / np.maximum(4 * ndl * ndv, 0.001)
    diff = (1 - fr) * (1 - metal) * base / PI
    return np.where((ndl > 0) & (ndv > 0), diff + spec, 0.0)


def _pdf(n, wi, wo, base, rough, metal):
    """eval_pdf, restir.wgsl:249-276, in float64."""
    ndl = (wi * n).sum(-1); ndv = (wo * n).sum(-1)
    f0 = 0.04 * (1 - metal) + base * metal
    fv = f0 + (1 - f0) * np.clip(1 - np.maximum(ndv, 0), 0, 1) ** 5
    lum_spec, lum_diff = fv, base * (1 - metal)                          # grey material: luminance(c) = c
    prob_spec = np.clip(lum_spec / (lum_spec + lum_diff + 0.0001), 0.001, 0.999)
    h = wi + wo; h /= np.linalg.norm(h, axis=-1, keepdims=True)
    ndh = np.maximum((h * n).sum(-1), 0.0)
    a = rough * rough; a2 = a * a
    d = a2 / (PI * (ndh * ndh * (a2 - 1) + 1) ** 2)
    g1 = 2 * ndv / (ndv + np.sqrt(rough * rough + (1 - rough * rough) * ndv * ndv))
    pdf = prob_spec * d * g1 / (4 * ndv) + (1 - prob_spec) * np.maximum(ndl, 0) / PI
    return np.where((ndl > 0) & (ndv > 0), pdf, 0.0)


def test_nee_with_the_references_mis_weights_integrates_to_the_predicted_share_of_the_direct_light(frt, orc):
    import _scenes
    b = _scenes.DualBuilder(frt, orc)
    plane = b.add_mesh(*_scenes._geo(frt, "create_plane"))
    base, rough, metal = 0.6, 0.5, 0.0
    m = frt.material_new([base, base, base, 1.0]); m.roughness = rough; m.metallic = metal
    floor = b.add_material(m)
    lm = b.add_material(_scenes._emissive(frt, 0, (1, 1, 1), 10.0))
    ref = frt.scenes.create_cornell_box().get("instances")
    b.add_instance(plane, floor, _scenes._mat(0, -1, 0, 6, 6, 6))
    b.add_instance(plane, lm, ref[5, 5:21].view(np.float32))            # the Cornell light quad: 0.5 x 0.5 at y = 0.99, facing down (scenes.rs)
    b.add_light(_scenes._quad_light(frt, (0, 0.99, 0), 0.25, (1, 1, 1, 10)))
    fs, os_ = b.build()
    W, H, N = 64, 48, 240
    ro = os_.renderer(W, H, 2, True, 8)
    acc = np.zeros((H, W)); acc2 = np.zeros((H, W))
    for k in range(N):
        ro.reset()
        cam = frt.CameraController().build_uniform(W / H, 0, 1)
        cam.frame_count = 7919 * k + 1                                   # the candidate's seed (restir.wgsl:797-798); renderer frame 0: no history
        ro.render_phases(cam, 1 | 2, 0, H)
        p_hat = ro.read(4, 0).view(np.float32).reshape(H, W, 8)[..., 7].astype(np.float64)
        acc += p_hat; acc2 += p_hat * p_hat
    est = acc / N
    sem = np.sqrt(np.maximum(acc2 / N - est * est, 0) / N)
    gpos = ro.read(0, 0).view(np.float32).reshape(H, W, 4).astype(np.float64)
    on_floor = (gpos[..., 3] == float(floor)) & (np.abs(gpos[..., 1] + 1.0) < 1e-4)
    assert on_floor.sum() > 800
    # the integral, by midpoint quadrature over the light quad
    x = gpos[on_floor][:, :3]
    cam_pos = np.array([0.0, 0.0, 3.0])
    wo = cam_pos - x; wo /= np.linalg.norm(wo, axis=-1, keepdims=True)
    n = np.array([0.0, 1.0, 0.0]); nl = np.array([0.0, -1.0, 0.0])
    Q = 48
    u = (np.arange(Q) + 0.5) / Q * 0.5 - 0.25
    ly = np.stack(np.meshgrid(u, u, indexing="ij"), -1).reshape(-1, 2)
    lp = np.stack([ly[:, 0], np.full(len(ly), 0.99), ly[:, 1]], -1)      # [Q*Q, 3]
    d = lp[None, :, :] - x[:, None, :]
    r2 = (d * d).sum(-1); wi = d / np.sqrt(r2)[..., None]
    cos_x = np.maximum((wi * n).sum(-1), 0); cos_l = np.maximum((-wi * nl).sum(-1), 0)
    wo_b = np.broadcast_to(wo[:, None, :], wi.shape)
    f = _bsdf(n, wi, wo_b, base, rough, metal)
    pdf_nee = 1.0 / 0.25                                                  # 1 / area / num_lights, area measure (:563)
    w_nee = pdf_nee / (pdf_nee + _pdf(n, wi, wo_b, base, rough, metal))  # :565 (eval_pdf takes normalize(ls.pos - hit.pos): the same wi up to 1e-3)
    full = (10.0 * f * cos_x * cos_l / r2).sum(-1) * (0.25 / (Q * Q))    # the direct-lighting integral L; dA = 0.5 * 0.5 / Q^2
    want = (10.0 * f * cos_x * cos_l / r2 * w_nee).sum(-1) * (0.25 / (Q * Q))   # what the reference's estimator converges to
    got, err = est[on_floor], sem[on_floor]
    lit = want > 0.02 * want.max()
    ratio = got[lit].sum() / want[lit].sum()
    z = (got[lit] - want[lit]) / np.maximum(err[lit], 1e-9)
    share = want[lit].sum() / full[lit].sum()
    print(f"\ndirect light, {lit.sum()} floor pixels x {N} seeds: sum(estimate) / sum(predicted) = {ratio:.4f}; per-pixel z-scores mean {z.mean():+.3f}, std {z.std():.3f}; "
          f"predicted / full direct-lighting integral = {share:.4f}, estimate / full = {got[lit].sum() / full[lit].sum():.4f}")
    assert abs(ratio - 1.0) < 0.01                          # within 1 % of the predicted integral over the floor
    assert abs(z.mean()) < 0.25 and 0.7 < z.std() < 1.4     # and no pixel-wise bias: errors look like their own Monte-Carlo noise
    assert 0.90 < share < 0.99                              # the reference's estimator leaves a few per cent of the direct light out (see above)
    # MAX_DEPTH = 1 (no bounce at all) must give the same expectation: the bounce contributes nothing to the direct light (weight 0)
    ro1 = os_.renderer(W, H, 1, True, 8)
    acc1 = np.zeros((H, W))
    for k in range(60):
        ro1.reset()
        cam = frt.CameraController().build_uniform(W / H, 0, 1); cam.frame_count = 104729 * k + 3
        ro1.render_phases(cam, 1 | 2, 0, H)
        acc1 += ro1.read(4, 0).view(np.float32).reshape(H, W, 8)[..., 7]
    r1 = (acc1 / 60)[on_floor][lit].sum() / want[lit].sum()
    assert abs(r1 - 1.0) < 0.02, r1
