// frt_round4_walks.hpp — round 4's collective walks (FRT_FLAG_WG_TRACE): built, bit-identical, measured slower than the plain kernels (Cornell Box
// 1.65 ms octant-sorted / 1.55 ms dense-only vs 1.43 ms per frame: profiles/r4_experiments/collective_walks.md) and therefore compiled into
// lib/libfrt_exp.so only. Included by frt_kernels.hip inside namespace frt, behind the product kernels it shares helpers with.
#pragma once

// ---- collective walks: a workgroup's rays re-dealt to dense, direction-sorted waves ------------------------------------------------------------------
// A lane's walk costs its WAVE the steps of the wave's slowest, most different ray, and in the traced kernels only 50 - 65 % of a wave's lanes bring a
// ray to a walk at all. Here the walks of a 16x16 workgroup are collective: every thread calls wg_trace (want = it has a ray), the rays are counted per
// direction octant, written to LDS in (octant, wave, lane) order — 8 words: origin, direction, t_max, source thread | t_min flag — and walked by the
// FIRST `total` threads: dense waves of rays that point the same way; waves beyond them wait at the barrier. The hit (5 words; any-hit: 1) goes back
// through the same LDS rows to the thread that asked. Only the ray travels: a path's state stays in its thread's registers. Which lane walks a ray
// cannot change its hit (hit semantics: frt_trace.hpp), so pixels and ray counts are those of the plain kernels.
// Host model of the lockstep walk (tools/bvh_quality.cpp, argument 13; Cornell Box bounce rays, 60 % of the lanes with a ray): 9,540 -> 6,180 wave-level
// node + leaf steps (-35 %): two thirds from the dense waves, one third from the octant order.
// Four barriers per call (counts | rays in place | rays read — the rows may now take results | results in place): all 256 threads of the workgroup must
// reach every call, so the kernels below run their paths in workgroup-uniform loops over the pulled-apart forms of frt_mono.hpp / frt_path.hpp
// (path_head with its ShadowReq, bounce_shade, spatial_neighbor_prepare / _finish).
static constexpr int kXRows = 8;      // exchange rows of 256 words behind the stack rows of a collective kernel's dynamic LDS
struct WgX { uint32_t* x; uint4* bins; };      // x: kXRows x 256 words; bins: [octant] -> the four waves' counts (in the shared row)
// t_min travels as a flag: the shaders fire rays with t_min 0.001 (primary, bounce, temporal shadow rays) or 0.0001 (spatial shadow / visibility rays)
template <bool ANY, bool VOTE>
__device__ __forceinline__ uint32_t wg_trace(const SceneView& sc, uint32_t* stk, const uint32_t* lds_top, const WgX& X, bool want, f3 o, f3 d, float tmin, float tmax, HitRec& h) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifndef FRT_WG_SORT
#define FRT_WG_SORT 1      // (A/B builds) 0: dense waves in arrival order, no octant sort
#endif
    uint32_t rank = 0u, pos, total = 0u;
    uint32_t* const bins_w = reinterpret_cast<uint32_t*>(X.bins);
#if FRT_WG_SORT
    const uint32_t key = (f2u(d.x) >> 31) | ((f2u(d.y) >> 31) << 1) | ((f2u(d.z) >> 31) << 2);
#pragma unroll
    for (uint32_t b = 0; b < 8u; ++b) {
        const bool mine = want && key == b;
        const unsigned long long m = __ballot(mine);
        if (mine) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == b) bins_w[b * 4u + wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    pos = rank;
#pragma unroll
    for (uint32_t b = 0; b < 8u; ++b) {
        const uint4 c4 = X.bins[b];
        const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) {
            total += cw[w];
            if (b < key || (b == key && w < wave)) pos += cw[w];
        }
    }
#else
    {
        const unsigned long long m = __ballot(want);
        rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0u) bins_w[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        const uint4 c4 = X.bins[0];
        const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
        pos = rank;
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) { total += cw[w]; if (w < wave) pos += cw[w]; }
    }
#endif
    if (total == 0u) { h.tri = 0xFFFFFFFFu; __syncthreads(); return 0u; }      // nobody asked (workgroup-uniform); the barrier keeps the bins from the next call's writers
    uint32_t* const x = X.x;
    if (want) {
        x[0 * kBlock + pos] = f2u(o.x); x[1 * kBlock + pos] = f2u(o.y); x[2 * kBlock + pos] = f2u(o.z);
        x[3 * kBlock + pos] = f2u(d.x); x[4 * kBlock + pos] = f2u(d.y); x[5 * kBlock + pos] = f2u(d.z);
        x[6 * kBlock + pos] = f2u(tmax); x[7 * kBlock + pos] = tid | (tmin < 0.0005f ? 0x100u : 0u);
    }
    __syncthreads();
    const bool work = tid < total;
    f3 ro = o, rd = d; float rtmax = tmax, rtmin = tmin; uint32_t src = tid;
    if (work) {
        ro = mk3(u2f(x[0 * kBlock + tid]), u2f(x[1 * kBlock + tid]), u2f(x[2 * kBlock + tid]));
        rd = mk3(u2f(x[3 * kBlock + tid]), u2f(x[4 * kBlock + tid]), u2f(x[5 * kBlock + tid]));
        rtmax = u2f(x[6 * kBlock + tid]);
        const uint32_t meta = x[7 * kBlock + tid];
        src = meta & 0xFFu; rtmin = (meta & 0x100u) ? 0.0001f : 0.001f;
    }
    __syncthreads();
    if (work) {
        HitRec hr;
        trace4<ANY, VOTE>(sc, ro, rd, rtmin, rtmax, stk, (uint32_t)kBlock, hr, lds_top);
        if (ANY) x[src] = hr.tri;
        else {
            x[0 * kBlock + src] = f2u(hr.t); x[1 * kBlock + src] = f2u(hr.u); x[2 * kBlock + src] = f2u(hr.v);
            x[3 * kBlock + src] = hr.tri; x[4 * kBlock + src] = hr.inst | (hr.front ? 0x80000000u : 0u);
        }
    }
    __syncthreads();
    h.tri = 0xFFFFFFFFu;
    if (want) {
        if (ANY) h.tri = x[tid];
        else {
            h.t = u2f(x[0 * kBlock + tid]); h.u = u2f(x[1 * kBlock + tid]); h.v = u2f(x[2 * kBlock + tid]);
            h.tri = x[3 * kBlock + tid];
            const uint32_t iw = x[4 * kBlock + tid];
            h.inst = iw & 0x7FFFFFFFu; h.front = (iw >> 31) != 0u;
        }
    }
    return total;
}

// The shared row of a collective kernel's LDS: [0..1] ray-count sums, [8..] reservation scratch, [32..191] quad nodes 0 .. 4, [192..223] wg_trace's bins.
struct WgLds { uint32_t* stack; uint32_t* cnt; uint32_t* tmp; WgX X; };
__device__ __forceinline__ WgLds wg_lds(uint32_t* dyn, uint32_t rows) {
    WgLds L;
    L.stack = dyn; L.cnt = dyn + (rows - 1u) * (uint32_t)kBlock; L.tmp = L.cnt + 8;
    L.X.x = dyn + rows * (uint32_t)kBlock; L.X.bins = reinterpret_cast<uint4*>(L.cnt + 192);
    return L;
}
// The bounces [d0, d1) of a workgroup's paths, collectively: path_loop_split (frt_mono.hpp) with every thread of the workgroup in step.
template <int VARIANT, bool VOTE>
__device__ __forceinline__ void wg_path_loop(PathCtx& c, const WgX& X, LoopState& s, bool have, uint32_t d0, uint32_t d1) {
    for (uint32_t depth = d0; depth < d1; ++depth) {
        const bool run = have && s.alive;
        HitRec h;
        if (run) c.n_closest++;
        const uint32_t n = wg_trace<false, VOTE>(c.sc, c.stk, c.lds_top, X, run, bounce_origin(s), s.next_dir, 0.001f, 100.0f, h);
        if (n == 0u) break;      // no path of the workgroup is alive any more (uniform)
        ShadowReq rq;
        rq.want = false; rq.add_now = false; rq.o = splat3(0.0f); rq.d = splat3(1.0f); rq.tmin = 0.001f; rq.tmax = 0.0f; rq.contrib = splat3(0.0f); rq.dark = splat3(0.0f);
        if (run) bounce_shade<VARIANT>(c, s, depth, h, rq);
        const bool w = run && rq.want;
        if (w) c.n_any++;
        HitRec hs;
        wg_trace<true, VOTE>(c.sc, c.stk, c.lds_top, X, w, rq.o, rq.d, rq.tmin, rq.tmax, hs);
        if (run) {
            bool lit = rq.add_now;
            if (rq.want) lit = hs.tri == 0xFFFFFFFFu;
            s.accumulated = s.accumulated + (lit ? rq.contrib : rq.dark);
        }
    }
}

// pixel_kernel (above) with collective walks. Dynamic LDS: `rows` stack rows (the scene's quad-tree stack need + the shared row) + kXRows exchange rows.
template <int STAGE, bool VOTE>
__global__ void __launch_bounds__(kBlock, FRT_WAVES) pixel_kernel_wg(SceneView sc, FrameView fv, ContQueue q, uint32_t cut, uint32_t* zero_counts, uint32_t rows) {
    extern __shared__ uint32_t s_wg[];
    const WgLds L = wg_lds(s_wg, rows);
    PathCtx c(sc, fv, L.stack + threadIdx.x, (uint32_t)kBlock);
    if (threadIdx.x < 2u) L.cnt[threadIdx.x] = 0u;
    c.lds_top = stage_top_nodes(sc, L.cnt);
    __syncthreads();
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    if (zero_counts && blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x <= (uint32_t)kMaxCuts) zero_counts[threadIdx.x] = 0u;
    uint32_t px, py;
    const bool active = tile_pixel(fv, px, py);
    const uint32_t pix = py * fv.W + px;
    const bool counted = active && py >= fv.own_y0 && py < fv.own_y1;
    LoopState s;
    s.alive = false; s.pos = splat3(0.0f); s.ffnormal = splat3(0.0f); s.next_dir = splat3(1.0f); s.throughput = splat3(0.0f); s.accumulated = splat3(0.0f);
    s.v1_pos = splat3(0.0f); s.last_bsdf_pdf = 0.0f; s.previous_was_diffuse = false; s.is_glass = false;
    ReservoirView r = zero_reservoir();
    bool traced = false;
    uint32_t seed = 0u;
    if (STAGE == 1) {
        if (active && !(fv.gpos[pix].w < 0.0f)) { seed = temporal_seed(fv, pix); traced = true; }   // background: merge_kernel writes the zero reservoir (restir.wgsl:805-811)
    } else {
        // the neighbour loop of restir_spatial.wgsl:912-993, every thread of the workgroup in step (a pixel takes 3 or 5 trips: spatial_begin)
        SpatialState ss;
        ss.i = 0u; ss.n = 0u; ss.pending = false;
        const bool sp = active && spatial_begin(c, ss, pix);
        SpatialCentre centre;
        if (sp) centre = spatial_centre(fv, pix);
        for (int it = 0; it < 5; ++it) {
            AnyReq rq;
            rq.want = false; rq.o = splat3(0.0f); rq.d = splat3(1.0f); rq.tmin = 0.0001f; rq.tmax = 0.0f;
            const bool mine = sp && ss.i < ss.n;
            if (mine) spatial_neighbor_prepare(c, ss, rq, centre);
            const bool w = mine && rq.want;
            if (w) c.n_any++;
            HitRec hv;
            wg_trace<true, VOTE>(sc, c.stk, c.lds_top, L.X, w, rq.o, rq.d, rq.tmin, rq.tmax, hv);
            if (mine) spatial_neighbor_finish(ss, w ? hv.tri == 0xFFFFFFFFu : true);
        }
        if (sp) { r = ss.r; seed = r.y; traced = true; }
    }
    {   // the primary hit (from the G-buffer) and the shadow ray of its next-event estimate
        ShadowReq rq;
        rq.want = false; rq.add_now = false; rq.o = splat3(0.0f); rq.d = splat3(1.0f); rq.tmin = 0.001f; rq.tmax = 0.0f; rq.contrib = splat3(0.0f); rq.dark = splat3(0.0f);
        if (traced) path_head<VARIANT>(c, pix, seed, s, &rq);
        const bool w = traced && rq.want;
        if (w) c.n_any++;
        HitRec hs;
        wg_trace<true, VOTE>(sc, c.stk, c.lds_top, L.X, w, rq.o, rq.d, rq.tmin, rq.tmax, hs);
        if (traced) {
            bool lit = rq.add_now;
            if (rq.want) lit = hs.tri == 0xFFFFFFFFu;
            s.accumulated = s.accumulated + (lit ? rq.contrib : rq.dark);
        }
    }
    const uint32_t d1 = cut < fv.max_depth ? cut : fv.max_depth;
    wg_path_loop<VARIANT, VOTE>(c, L.X, s, traced, 1u, d1);
    // park the survivors: ONE atomic for the workgroup; a lane that finds the queue full keeps its path and finishes it in place (plain walks)
    const uint32_t slot = workgroup_reserve(q.count, traced && s.alive, L.tmp);
    const bool parked = traced && s.alive && slot < q.capacity;
    const bool rest = traced && s.alive && !parked;
    if (parked) cont_store(q, slot, pix, c.rng, counted, s, STAGE == 2 ? &r : nullptr);
    if (__ballot(rest) != 0ull) {
        note_queue_overflow(q, rest);
        if (rest) path_loop<VARIANT>(c, s, d1, fv.max_depth);
    }
    if (traced && !parked) finish_path<STAGE>(c, pix, r, s);
    flush_ray_counters(fv, counted ? c.n_closest : 0u, counted ? c.n_any : 0u, L.cnt);
}

// continue_kernel (above) with collective walks.
template <int STAGE, bool VOTE>
__global__ void __launch_bounds__(kBlock, FRT_WAVES) continue_kernel_wg(SceneView sc, FrameView fv, ContQueue qin, ContQueue qout, uint32_t d0, uint32_t d1, uint32_t rows) {
    extern __shared__ uint32_t s_wg[];
    const WgLds L = wg_lds(s_wg, rows);
    const uint32_t filled = *qin.count;
    const uint32_t n = filled < qin.capacity ? filled : qin.capacity;
    if (blockIdx.x == 0u && threadIdx.x == 0u && filled > qin.capacity && qin.overflow) {      // (continue_kernel: the mapped overflow flag)
        uint32_t* const seen = *reinterpret_cast<uint32_t* const*>(qin.overflow + 2);
        if (seen) *reinterpret_cast<volatile uint32_t*>(seen) = 1u;
    }
    if (blockIdx.x * (uint32_t)kBlock >= n) return;   // uniform per workgroup
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    PathCtx c(sc, fv, L.stack + threadIdx.x, (uint32_t)kBlock);
    if (threadIdx.x < 2u) L.cnt[threadIdx.x] = 0u;
    c.lds_top = stage_top_nodes(sc, L.cnt);
    __syncthreads();
    uint32_t cnt_closest = 0u, cnt_any = 0u;
    for (uint32_t base = blockIdx.x * (uint32_t)kBlock; base < n; base += gridDim.x * (uint32_t)kBlock) {
        const uint32_t slot_in = base + threadIdx.x;
        LoopState s;
        s.alive = false; s.pos = splat3(0.0f); s.ffnormal = splat3(0.0f); s.next_dir = splat3(1.0f); s.throughput = splat3(0.0f); s.accumulated = splat3(0.0f);
        s.v1_pos = splat3(0.0f); s.last_bsdf_pdf = 0.0f; s.previous_was_diffuse = false; s.is_glass = false;
        ReservoirView r = zero_reservoir();
        uint32_t pix = 0u;
        bool owned = false;
        c.n_closest = 0u; c.n_any = 0u;
        const bool have = slot_in < n;
        if (have) cont_load(qin, slot_in, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
        wg_path_loop<VARIANT, VOTE>(c, L.X, s, have, d0, d1);
        // park the survivors for the next launch (one atomic per wave); a lane that finds the queue full finishes its path in place
        const bool alive = have && s.alive;
        const uint32_t slot = wave_reserve(qout.count, alive);
        const bool parked = alive && slot < qout.capacity;
        const bool rest = alive && !parked;
        if (parked) cont_store(qout, slot, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
        if (__ballot(rest) != 0ull) {
            note_queue_overflow(qout, rest);
            if (rest) path_loop<VARIANT>(c, s, d1, fv.max_depth);
        }
        if (have && !parked) finish_path<STAGE>(c, pix, r, s);
        if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
    }
    flush_ray_counters(fv, cnt_closest, cnt_any, L.cnt);
}

