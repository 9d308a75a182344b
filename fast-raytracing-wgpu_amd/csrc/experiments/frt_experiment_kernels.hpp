// frt_experiment_kernels.hpp — kernel designs that were built, checked bit for bit against the oracle, measured and NOT kept as the default
// (HISTORY.md section 6, "What was tried"; numbers in profiles/r2_experiments/). They are compiled only into lib/libfrt_exp.so
// (`make experiments`: -DFRT_EXPERIMENTS=1), which the sweep scripts under tools/ and tests/test_experiments.py load; the product library
// lib/libfrt.so contains none of this code and reads none of its environment knobs.
//   compact_kernel            FRT_FLAG_COMPACTION   workgroup-level path compaction over the resumable state machine of frt_path.hpp
//   bounce_kernel             FRT_REFILL            one bounce per loop trip with lane refill from the continuation queue
//   stream_kernel             FRT_STREAM            resumable traversal: every lane a little state machine over its two rays
//   wf_trace / wf_shade       FRT_WAVEFRONT         ray-level wavefront: a traversal launch and a shading launch per bounce depth
//   resident_*_kernel         FRT_RESIDENT          quantized pair nodes + triangles cached in LDS, persistent 1024-thread workgroups
//   TileOrder                 FRT_TILE_ORDER        sweep of the tile rows from the expensive end of the image (resident pixel kernel only)
// Included by frt_kernels.hip inside namespace frt, behind the product kernels it shares helpers with.
#pragma once


// Which row of 16x16 tiles a workgroup of a traced stage takes. The time a tile needs varies threefold over the image (ceiling vs
// floor of a Cornell Box) and a launch ends when its last workgroup does, so the sweep over the tile rows starts at the expensive
// end of the image and finishes on the cheap one: top to bottom or bottom to top, chosen from the workgroup times the top and the
// bottom eighth of the image reported in the stage's previous launch. A sweep, not a sort: orders that scatter the rows (or single
// tiles) by cost were measured and lose more to the broken neighbourhood of consecutive workgroups (2.45 / 2.37 ms) than the shorter
// tail gains (2.35 ms against 2.42 top to bottom). Scheduling only: which pixels a workgroup computes, never what it computes.
// OPT-IN since the quad-tree kernels (FRT_TILE_ORDER=1): with them the sweep costs 1-2 % on one stream and on two (frt_renderer.hip).
// State per traced stage (device memory, 6 words): [0] flip (1 = bottom tile row first), [1] ticket of finished workgroups,
// [2..3] summed time of the top eighth, [4..5] of the bottom eighth. The LAST workgroup of a launch to finish (ticket) turns the two
// sums into the next launch's direction and clears them: no extra kernel, nothing on the critical path.
struct TileOrder { uint32_t* st; uint32_t nrows; };
__device__ __forceinline__ uint32_t ordered_tile_row(const TileOrder& to) {
    if (!to.st) return blockIdx.y;
    return to.st[0] ? to.nrows - 1u - blockIdx.y : blockIdx.y;
}
__device__ __forceinline__ void report_tile_cost(const TileOrder& to, uint32_t tile_row, unsigned long long t_begin) {
    if (!to.st || threadIdx.x != 0u) return;
    const unsigned long long cost = (__builtin_amdgcn_s_memtime() - t_begin) >> 8;
    const uint32_t k = to.nrows / 8u > 0u ? to.nrows / 8u : 1u;
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(to.st + 2);
    // No fences (a device-scope fence invalidates the CU's vector L1, i.e. the BVH working set of the workgroups still running there:
    // measured +12 % on the frame). All four operations are L2 atomics; the ticket add is made to depend on the values the cost adds
    // return, so it is issued after they have been performed, and the last workgroup reads the sums with atomics again.
    unsigned long long seen = 0ull;
    if (tile_row < k) seen |= atomicAdd(&sums[0], cost);
    if (tile_row + k >= to.nrows) seen |= atomicAdd(&sums[1], cost);
    const uint32_t one = (seen == ~0ull) ? 2u : 1u;      // always 1 (the sums never reach 2^64 - 1); keeps the dependency
    if (atomicAdd(&to.st[1], one) == gridDim.x * gridDim.y - 1u) {   // every other workgroup has read st[0] and added its cost
        const unsigned long long top = atomicExch(&sums[0], 0ull), bottom = atomicExch(&sums[1], 0ull);
        to.st[0] = top <= bottom ? 1u : 0u;   // the sweep should END on the cheaper eighth of the image
        to.st[1] = 0u;
    }
}

// Opt-in (FRT_FLAG_COMPACTION) temporal (STAGE 1) / spatial + shade (STAGE 2) kernels: workgroup-level path compaction.
//
// One 512-thread workgroup (8 waves) owns a 32x16 pixel block, one wave per 8x8 tile, and runs the resumable path state
// machine of frt_path.hpp one bounce per iteration:  A closest-hit rays -> B shade / NEE set-up -> C shadow rays -> D BSDF
// sample, retire finished paths. Paths end at very different depths (roulette, light hits, the open front of the box), so
// after a couple of bounces most lanes of every wave would idle (23 % lane utilisation measured on the thread-per-pixel
// kernel, profiles/r1_v1_pmc_summary.md). Before each bounce the workgroup therefore takes a census with one wave ballot
// per wave; whenever the surviving paths fit into fewer waves, each survivor computes its rank (ballot prefix + the counts
// of the waves before it), parks its 24-32 words of path state in LDS at that rank, and the first ceil(live / 64) waves pick
// the states up: the wave count shrinks 8 -> 4 -> 2 -> 1 as the paths die, and the remaining waves stay dense. The exchange
// buffer is the traversal-stack memory (no ray is in flight at that point), so compaction costs no extra LDS.
// A parked state carries its pixel index, so any lane can finish any pixel; results do not depend on the lane a path runs in.
static constexpr int kBlockC = 512;
static constexpr int kWavesC = kBlockC / 64;

__device__ __forceinline__ void xput(uint32_t* x, uint32_t cap, uint32_t slot, int k, uint32_t v) { x[(uint32_t)k * cap + slot] = v; }
__device__ __forceinline__ void xputf(uint32_t* x, uint32_t cap, uint32_t slot, int k, float v) { x[(uint32_t)k * cap + slot] = f2u(v); }
__device__ __forceinline__ uint32_t xget(const uint32_t* x, uint32_t cap, uint32_t slot, int k) { return x[(uint32_t)k * cap + slot]; }
__device__ __forceinline__ float xgetf(const uint32_t* x, uint32_t cap, uint32_t slot, int k) { return u2f(x[(uint32_t)k * cap + slot]); }

template <int STAGE>
__device__ __forceinline__ void park_state(uint32_t* x, uint32_t cap, uint32_t slot, const PathState& st, uint32_t rng, bool owned, const ReservoirView& r) {
    xput(x, cap, slot, 0, st.pix); xput(x, cap, slot, 1, st.depth);
    xput(x, cap, slot, 2, (st.prev_diffuse ? 1u : 0u) | (st.is_glass ? 2u : 0u) | (st.front_face ? 4u : 0u) | (owned ? 8u : 0u));
    xput(x, cap, slot, 3, rng);
    xputf(x, cap, slot, 4, st.pos.x); xputf(x, cap, slot, 5, st.pos.y); xputf(x, cap, slot, 6, st.pos.z);
    xputf(x, cap, slot, 7, st.ffnormal.x); xputf(x, cap, slot, 8, st.ffnormal.y); xputf(x, cap, slot, 9, st.ffnormal.z);
    xputf(x, cap, slot, 10, st.throughput.x); xputf(x, cap, slot, 11, st.throughput.y); xputf(x, cap, slot, 12, st.throughput.z);
    xputf(x, cap, slot, 13, st.accum.x); xputf(x, cap, slot, 14, st.accum.y); xputf(x, cap, slot, 15, st.accum.z);
    xputf(x, cap, slot, 16, st.next_dir.x); xputf(x, cap, slot, 17, st.next_dir.y); xputf(x, cap, slot, 18, st.next_dir.z);
    xputf(x, cap, slot, 19, st.v1_pos.x); xputf(x, cap, slot, 20, st.v1_pos.y); xputf(x, cap, slot, 21, st.v1_pos.z);
    xputf(x, cap, slot, 22, st.hit_t); xputf(x, cap, slot, 23, st.last_pdf);
    if (STAGE == 2) {
        xput(x, cap, slot, 24, r.y); xputf(x, cap, slot, 25, r.w_sum); xput(x, cap, slot, 26, r.M); xputf(x, cap, slot, 27, r.W);
        xputf(x, cap, slot, 28, r.sx); xputf(x, cap, slot, 29, r.sy); xputf(x, cap, slot, 30, r.sz); xputf(x, cap, slot, 31, r.p_hat);
    }
}
template <int STAGE>
__device__ __forceinline__ void fetch_state(const uint32_t* x, uint32_t cap, uint32_t slot, PathState& st, uint32_t& rng, bool& owned, ReservoirView& r) {
    st.pix = xget(x, cap, slot, 0); st.depth = xget(x, cap, slot, 1);
    uint32_t fl = xget(x, cap, slot, 2);
    st.prev_diffuse = fl & 1u; st.is_glass = fl & 2u; st.front_face = fl & 4u; owned = fl & 8u;
    rng = xget(x, cap, slot, 3);
    st.pos = mk3(xgetf(x, cap, slot, 4), xgetf(x, cap, slot, 5), xgetf(x, cap, slot, 6));
    st.ffnormal = mk3(xgetf(x, cap, slot, 7), xgetf(x, cap, slot, 8), xgetf(x, cap, slot, 9));
    st.throughput = mk3(xgetf(x, cap, slot, 10), xgetf(x, cap, slot, 11), xgetf(x, cap, slot, 12));
    st.accum = mk3(xgetf(x, cap, slot, 13), xgetf(x, cap, slot, 14), xgetf(x, cap, slot, 15));
    st.next_dir = mk3(xgetf(x, cap, slot, 16), xgetf(x, cap, slot, 17), xgetf(x, cap, slot, 18));
    st.v1_pos = mk3(xgetf(x, cap, slot, 19), xgetf(x, cap, slot, 20), xgetf(x, cap, slot, 21));
    st.hit_t = xgetf(x, cap, slot, 22); st.last_pdf = xgetf(x, cap, slot, 23);
    st.done = false;
    if (STAGE == 2) {
        r.y = xget(x, cap, slot, 24); r.w_sum = xgetf(x, cap, slot, 25); r.M = xget(x, cap, slot, 26); r.W = xgetf(x, cap, slot, 27);
        r.sx = xgetf(x, cap, slot, 28); r.sy = xgetf(x, cap, slot, 29); r.sz = xgetf(x, cap, slot, 30); r.p_hat = xgetf(x, cap, slot, 31);
    }
}

template <int STAGE>
__global__ void __launch_bounds__(kBlockC) compact_kernel(SceneView sc, FrameView fv) {
    __shared__ uint32_t s_mem[kStackDepth * kBlockC];   // per-lane traversal stacks; the exchange buffer while compacting
    __shared__ uint32_t s_live[2][kWavesC];
    __shared__ uint32_t s_cnt[2];
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    if (tid < 2u) s_cnt[tid] = 0u;
    const uint32_t px = blockIdx.x * 32u + (wave & 3u) * 8u + (lane & 7u);
    const uint32_t py = fv.y0 + blockIdx.y * 16u + (wave >> 2) * 8u + (lane >> 3);
    PathCtx c(sc, fv, &s_mem[tid], (uint32_t)kBlockC);
    PathState st;
    SpatialState ss;
    ReservoirView r = zero_reservoir();
    st.done = true; st.depth = 0u; st.pix = 0u;
    bool alive = false, owned = false;
    uint32_t n_closest = 0u, n_any = 0u;

    if (px < fv.W && py < fv.y1) {
        const uint32_t pix = py * fv.W + px;
        owned = py >= fv.own_y0 && py < fv.own_y1;
        if (STAGE == 1) alive = temporal_begin(c, st, pix);
        else if (spatial_begin(c, ss, pix)) {
            // neighbour loop (restir_spatial.wgsl:912-993): coherent across the tile, stays in its lane
            while (ss.i < ss.n) {
                AnyReq req;
                req.want = false; req.o = splat3(0.0f); req.d = splat3(0.0f); req.tmin = 0.0f; req.tmax = 0.0f;
                spatial_neighbor_prepare(c, ss, req);
                bool visible = true;
                if (req.want) {
                    HitRec s;
                    if (owned) n_any++;
                    trace<true>(sc, req.o, req.d, req.tmin, req.tmax, c.stk, c.stride, s);
                    visible = s.tri == 0xFFFFFFFFu;
                }
                spatial_neighbor_finish(ss, visible);
            }
            r = ss.r;
            path_begin(c, st, pix, r.y);
            alive = true;
        }
    }

    uint32_t cur_waves = kWavesC;
    for (uint32_t it = 0;; ++it) {
        // ---- census: one ballot per wave, counts shared through LDS (double-buffered: one barrier per iteration)
        const unsigned long long m = __ballot(alive);
        uint32_t* live = s_live[it & 1u];
        if (lane == 0u) live[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t total = 0u, base = 0u;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)kWavesC; ++w) { uint32_t v = live[w]; total += v; base += (w < wave) ? v : 0u; }
        if (total == 0u) break;
        const uint32_t new_waves = (total + 63u) >> 6;
        if (new_waves < cur_waves) {
            // ---- compaction: survivors park their state at their rank; the first new_waves waves pick the states up
            const uint32_t cap = new_waves * 64u;
            if (alive) {
                uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                park_state<STAGE>(s_mem, cap, rank, st, c.rng, owned, r);
            }
            __syncthreads();
            alive = tid < total;
            if (alive) fetch_state<STAGE>(s_mem, cap, tid, st, c.rng, owned, r);
            cur_waves = new_waves;
            __syncthreads();   // the exchange buffer becomes stack memory again
        }
        if (!alive) continue;   // whole idle waves only meet the barriers

        // ---- A: closest-hit ray of this bounce (depth >= 1; the depth-0 hit is the G-buffer)
        HitRec h;
        h.tri = 0xFFFFFFFFu; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.inst = 0u; h.front = false;
        f3 origin = splat3(0.0f);
        if (st.depth >= 1u) {
            if (path_pre_closest(c, st, origin)) {
                if (owned) n_closest++;
                trace<false>(sc, origin, st.next_dir, 0.001f, 100.0f, c.stk, c.stride, h);
            }
        }
        // ---- B: shade up to the shadow ray
        AnyReq req;
        req.want = false; req.o = splat3(0.0f); req.d = splat3(0.0f); req.tmin = 0.0f; req.tmax = 0.0f;
        if (!st.done) path_shade<VARIANT>(c, st, h, origin, req);
        // ---- C: NEE shadow ray
        bool visible = true;
        if (req.want) {
            HitRec s;
            if (owned) n_any++;
            trace<true>(sc, req.o, req.d, req.tmin, req.tmax, c.stk, c.stride, s);
            visible = s.tri == 0xFFFFFFFFu;
        }
        // ---- D: NEE add + BSDF sample; retire finished paths
        if (!st.done) path_post_any(c, st, visible);
        if (st.done) {
            if (STAGE == 1) temporal_finalize(c, st);
            else { ss.r = r; ss.pix = st.pix; spatial_finalize(c, ss, st); }
            alive = false;
        }
    }
    flush_ray_counters(fv, n_closest, n_any, s_cnt);
}

// Bounce kernel: every bounce of every parked path, with LANE REFILL. The continuation kernels above start dense and thin out: a wave that
// resumes 64 paths at depth 3 has a dozen left two bounces later (15 % lane utilisation, profiles/r2a_pmc.txt). Here a wave runs ONE bounce
// per trip of its loop and, before each trip, hands the lanes whose path has ended a fresh parked path from the queue (one atomic per wave
// and trip; the queue was filled completely by the previous launch, so there is nothing to wait for). Lanes of a wave are then at different
// depths of different paths — which is fine: a bounce iteration is the same code at every depth (restir.wgsl:590-733), a path's arithmetic
// and rand() sequence do not depend on the lane that runs it, and the per-lane `depth` feeds the two places that look at it (the v1 capture
// at depth 1 and the loop bound). The wave stays dense until the queue runs dry; with the cut at depth 1 (the pixel kernel then does the
// primary hit only) all bounce work of a stage runs this way.
template <int STAGE>
__global__ void __launch_bounds__(kBlock, 4) bounce_kernel(SceneView sc, FrameView fv, ContQueue qin, uint32_t* head, uint32_t d0, uint32_t refill_min) {
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    const uint32_t filled = *qin.count;
    const uint32_t n = filled < qin.capacity ? filled : qin.capacity;
    PathCtx c(sc, fv, &s_stack[threadIdx.x], (uint32_t)kBlock);
    LoopState s;
    s.alive = false;
    s.pos = s.ffnormal = s.throughput = s.accumulated = s.next_dir = s.v1_pos = splat3(0.0f);
    s.last_bsdf_pdf = 0.0f; s.previous_was_diffuse = false; s.is_glass = false;
    ReservoirView r = zero_reservoir();
    uint32_t pix = 0u, depth = d0, cnt_closest = 0u, cnt_any = 0u;
    bool owned = false, more = true;      // more: the queue may still hold unclaimed paths (wave-uniform)
    for (;;) {
        const unsigned long long dead = __ballot(!s.alive);
        const uint32_t k = (uint32_t)__popcll(dead);
        if (more && k >= refill_min) {
            const int leader = __ffsll((long long)dead) - 1;
            uint32_t base = 0u;
            if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(head, k);
            base = __shfl(base, leader, 64);
            const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
            if (!s.alive && slot < n) {
                if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }      // the rays of the path this lane ran before
                c.n_closest = 0u; c.n_any = 0u;
                cont_load(qin, slot, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
                depth = d0;
            }
            if (base + k >= n) more = false;
        }
        if (__ballot(s.alive) == 0ull) {
            if (!more) break;
            continue;      // (only when refill_min > the dead lanes of an all-dead wave, i.e. never: 64 >= refill_min)
        }
        if (s.alive) {
            path_loop<VARIANT>(c, s, depth, depth + 1u);
            depth += 1u;
            if (!s.alive) finish_path<STAGE>(c, pix, r, s);
        }
    }
    if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
    flush_ray_counters(fv, cnt_closest, cnt_any, s_cnt);
}

// Stream kernel: the bounce kernel with the traversal made RESUMABLE, so that a wave stops paying for its slowest ray.
// Measured with lane refill alone (profiles/r2g_refill_pmc.txt): even when all 64 lanes start every bounce together, lane utilisation stays
// at 25-27 %. The loss is inside the bounce: incoherent rays need 5 to 40 node steps and a while-while traversal runs until the LAST lane
// of the wave is done, twice per bounce (closest hit, shadow); the shading between them waits for those stragglers too.
// Here every lane is a little state machine
//        CLOSEST --(ray done)--> SHADE --bounce_shade--> SHADOW --(ray done: add the estimate)--> CLOSEST of the next bounce ...
// and the wave alternates between two kinds of trips: a TRAVERSAL round (a few node steps + one leaf step for every lane that has a ray
// in flight, shadow or closest alike — same code) and a SHADING trip (bounce_shade, frt_mono.hpp, for the lanes whose closest-hit ray has
// finished), taken once enough lanes wait for it or nobody is traversing. A ray that needs 40 steps simply stays in flight over several
// rounds while its neighbours shade, fire their shadow rays and start the next bounce; lanes whose path has ended are refilled from the
// queue as in bounce_kernel. Per lane the sequence of operations is path_loop's (bounce_shade is checked against path_loop on the CPU,
// tests/hostcheck "stream"), so pixels and ray counts are unchanged.
enum : uint32_t { M_IDLE = 0u, M_SHADOW = 1u, M_CLOSEST = 2u, M_SHADE = 3u };
struct LaneRay { f3 o, d; float tmin, tmax; uint32_t cur; int sp; float t, u, v, det; uint32_t tri, inst; };
// (node steps per traversal round before the wave looks at its lanes again: the `slice` argument, 8 by default)

__device__ __forceinline__ void lane_ray_begin(LaneRay& tr, f3 o, f3 d, float tmin, float tmax) {
    tr.o = o; tr.d = d; tr.tmin = tmin; tr.tmax = tmax; tr.cur = 0u; tr.sp = 0;
    tr.t = tmax; tr.u = 0.0f; tr.v = 0.0f; tr.det = 0.0f; tr.tri = 0xFFFFFFFFu; tr.inst = 0u;
}
// One traversal round for the lanes with `go` set: up to `slice` node steps (while any of them is at an inner node), then one leaf
// (all its triangles) for the lanes that hold one. tr.cur == 0xFFFFFFFF afterwards: the ray is finished. Same box / triangle arithmetic and
// the same visiting order per ray as trace() (frt_trace.hpp): the closest hit and the any-hit answer are the ones trace() finds.
__device__ __forceinline__ void traverse_round(const SceneView& sc, LaneRay& tr, bool go, bool any_hit, uint32_t* stk, uint32_t stride, int slice) {
    const uint32_t kDone = 0xFFFFFFFFu;
    f3 inv = mk3(prune_rcp(tr.d.x), prune_rcp(tr.d.y), prune_rcp(tr.d.z));
    f3 oinv = mk3(-tr.o.x * inv.x, -tr.o.y * inv.y, -tr.o.z * inv.z);
#pragma nounroll
    for (int it = 0; it < slice; ++it) {
        const bool at_node = go && !(tr.cur & 0x80000000u);
        if (__ballot(at_node) == 0ull) break;
        if (at_node) {
            const float4* n = sc.nodes + (size_t)tr.cur * 4u;
            float4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            float tlim = any_hit ? tr.tmax : tr.t;
            float t0, t1;
            bool h0, h1;
            slab2(q0, q1, q2, inv, oinv, tr.tmin, tlim, t0, t1, h0, h1);
            uint32_t r0 = f2u(q3.x), r1 = f2u(q3.y);
            h0 = h0 && (r0 != kDone);
            h1 = h1 && (r1 != kDone);
            if (h0 && h1) {
                bool swap = t1 < t0;
                uint32_t nearr = swap ? r1 : r0, farr = swap ? r0 : r1;
                stk[(uint32_t)tr.sp * stride] = farr; ++tr.sp;
                tr.cur = nearr;
            } else if (h0) tr.cur = r0;
            else if (h1) tr.cur = r1;
            else if (tr.sp == 0) tr.cur = kDone;
            else { --tr.sp; tr.cur = stk[(uint32_t)tr.sp * stride]; }
        }
    }
    if (go && (tr.cur & 0x80000000u) && tr.cur != kDone) {
        const uint32_t first = tr.cur & 0x00FFFFFFu, count = (tr.cur >> 24) & 0x7Fu;
        bool found = false;
        for (uint32_t k = 0; k < count && !found; ++k) {
            const float4* tp = sc.tris + (size_t)(first + k) * 3u;
            float4 a = tp[0], b = tp[1], c4 = tp[2];
            float t, u, v, det;
            if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c4.x, c4.y, c4.z), tr.o, tr.d, tr.tmin, tr.tmax, t, u, v, det)) {
                uint32_t id = f2u(a.w);
                if (any_hit) { tr.tri = id; tr.t = t; found = true; }
                else if (t < tr.t || (t == tr.t && id < tr.tri)) { tr.t = t; tr.u = u; tr.v = v; tr.tri = id; tr.inst = f2u(b.w); tr.det = det; }
            }
        }
        if (found || tr.sp == 0) tr.cur = kDone;
        else { --tr.sp; tr.cur = stk[(uint32_t)tr.sp * stride]; }
    }
}

template <int STAGE>
__global__ void __launch_bounds__(kBlock, 4) stream_kernel(SceneView sc, FrameView fv, ContQueue qin, uint32_t* head, uint32_t d0, uint32_t refill_min, uint32_t shade_min, int slice) {
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    const uint32_t filled = *qin.count;
    const uint32_t n = filled < qin.capacity ? filled : qin.capacity;
    PathCtx c(sc, fv, &s_stack[threadIdx.x], (uint32_t)kBlock);
    LoopState s;
    s.alive = false;
    s.pos = s.ffnormal = s.throughput = s.accumulated = s.next_dir = s.v1_pos = splat3(0.0f);
    s.last_bsdf_pdf = 0.0f; s.previous_was_diffuse = false; s.is_glass = false;
    ReservoirView r = zero_reservoir();
    LaneRay tr;
    lane_ray_begin(tr, splat3(0.0f), splat3(0.0f), 0.0f, 0.0f);
    f3 contrib = splat3(0.0f), dark = splat3(0.0f);
    uint32_t pix = 0u, depth = d0, cnt_closest = 0u, cnt_any = 0u, mode = M_IDLE;
    bool owned = false, more = true;
    // a lane that has no ray in flight and nothing to shade: next ray of its path, or the path is finished
    auto next_ray = [&]() {
        if (s.alive) { c.n_closest++; lane_ray_begin(tr, bounce_origin(s), s.next_dir, 0.001f, 100.0f); mode = M_CLOSEST; }
        else { finish_path<STAGE>(c, pix, r, s); mode = M_IDLE; }
    };
    for (;;) {
        // ---- refill: lanes whose path has ended take a parked path from the queue
        const unsigned long long idle = __ballot(mode == M_IDLE);
        const uint32_t k = (uint32_t)__popcll(idle);
        if (more && k >= refill_min) {
            const int leader = __ffsll((long long)idle) - 1;
            uint32_t base = 0u;
            if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(head, k);
            base = __shfl(base, leader, 64);
            const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (mode == M_IDLE && slot < n) {
                if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
                c.n_closest = 0u; c.n_any = 0u;
                cont_load(qin, slot, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
                depth = d0;
                next_ray();      // (s.alive: a parked path always has its next iteration)
            }
            if (base + k >= n) more = false;
        }
        const uint32_t n_trav = (uint32_t)__popcll(__ballot(mode == M_SHADOW || mode == M_CLOSEST));
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(mode == M_SHADE));
        if (n_trav == 0u && n_shade == 0u) {
            if (!more) break;
            continue;
        }
        if (n_shade >= shade_min || n_trav == 0u) {
            // ---- shading trip
            if (mode == M_SHADE) {
                HitRec h;
                h.t = tr.t; h.u = tr.u; h.v = tr.v; h.tri = tr.tri; h.inst = tr.inst; h.front = false;
                if (h.tri != 0xFFFFFFFFu) {
                    bool front = tr.det > 0.0f;
                    if (sc.instances[h.inst].flip) front = !front;
                    h.front = front;
                }
                ShadowReq req;
                bounce_shade<VARIANT>(c, s, depth, h, req);
                depth += 1u;
                if (req.want) {
                    contrib = req.contrib; dark = req.dark;
                    c.n_any++;
                    lane_ray_begin(tr, req.o, req.d, req.tmin, req.tmax);
                    mode = M_SHADOW;
                } else {
                    s.accumulated = s.accumulated + (req.add_now ? req.contrib : req.dark);
                    next_ray();
                }
            }
        } else {
            // ---- traversal round
            const bool go = mode == M_SHADOW || mode == M_CLOSEST;
            traverse_round(sc, tr, go, mode == M_SHADOW, c.stk, c.stride, slice);
            if (go && tr.cur == 0xFFFFFFFFu) {
                if (mode == M_SHADOW) {
                    s.accumulated = s.accumulated + (tr.tri == 0xFFFFFFFFu ? contrib : dark);
                    next_ray();
                } else mode = M_SHADE;
            }
        }
    }
    if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
    flush_ray_counters(fv, cnt_closest, cnt_any, s_cnt);
}

// ---- ray-level wavefront (opt-in, FRT_WAVEFRONT=1) -----------------------------------------------------------------------------------
// What the experiments above point to (HISTORY.md §6): the traced kernels lose their lanes to wave-level TRAVERSAL divergence, so traversal
// gets a kernel of its own. After the pixel kernel (primary hit only, cut at depth 1) each bounce depth d is two launches:
//   wf_trace_kernel   persistent waves take RAYS — the closest-hit ray of a parked path's iteration d, or the shadow ray its iteration d-1
//                     left pending — from an item list, one ray per lane, and a lane that finishes its ray takes the next one (resumable
//                     traverse_round, refill by ballot + one atomic per wave): no lane waits for the slowest ray of its wave, and with
//                     ~50 VGPRs the kernel runs 8 waves per SIMD, twice the latency hiding of the shading kernels;
//   wf_shade_kernel   one lane per parked path, dense: adds the pending estimate (lit or dark), shades the hit with bounce_shade — the
//                     iteration with its rays pulled apart, frt_mono.hpp, checked against path_loop on the CPU — and parks the survivor
//                     (and any path that still owes a shadow ray) for depth d + 1, emitting its ray items.
// Records are the continuation records (30 words) + the pending shadow ray and its two possible contributions (14 words); hits come back
// through a 7-word side buffer. Same per-path arithmetic and rand() order as path_loop: pixels and ray counts unchanged.
static constexpr int kWfWords = 44, kWfSub = 8;
enum : uint32_t { WF_SHADOW = 8u, WF_ENDED = 16u };
// Queues and item lists are cut into kWfSub regions with a counter each (frt_mono.hpp: ContQueue::nsub): a workgroup parks into the region
// blockIdx % kWfSub, one atomic per WORKGROUP and counter; the trace kernel takes its rays by static chunking, no atomics at all.
// (First build: one counter per list, one atomic per wave: every launch cost >= 120 us and the stages were 4x slower than the plain kernels.)
struct WfPass {
    uint32_t* qin; const uint32_t* n_in;          // records of depth d (kWfWords x capacity, SoA), counts per region
    uint32_t* qout; uint32_t* n_out;              // records of depth d + 1
    const uint32_t* items_in; const uint32_t* n_items_in;   // rays to trace for depth d: slot << 1 | kind (0 closest, 1 shadow), 2 x capacity, regions; null = every record, closest
    uint32_t* items_out; uint32_t* n_items_out;
    uint32_t* hits;                               // 7 x capacity: t, u, v, tri, inst, front | unoccluded
    uint32_t capacity;
    uint32_t* overflow;
};

template <int STACK>
__global__ void __launch_bounds__(kBlock, 8) wf_trace_kernel(SceneView sc, FrameView fv, WfPass io, uint32_t refill_min, int slice) {
    __shared__ uint32_t s_stack[STACK * kBlock];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const size_t cap = io.capacity;
    const uint32_t rcap = io.capacity / (uint32_t)kWfSub, icap = 2u * rcap;
    // chunks of 64 items, region after region; wave w takes chunks w, w + W, w + 2W, ...
    uint32_t cnt[kWfSub], first_chunk[kWfSub + 1];
    first_chunk[0] = 0u;
#pragma unroll
    for (int j = 0; j < kWfSub; ++j) {
        const uint32_t v = io.items_in ? io.n_items_in[j] : io.n_in[j], lim = io.items_in ? icap : rcap;
        cnt[j] = v < lim ? v : lim;
        first_chunk[j + 1] = first_chunk[j] + (cnt[j] + 63u) / 64u;
    }
    const uint32_t total_chunks = first_chunk[kWfSub], n_waves = gridDim.x * (uint32_t)(kBlock / 64);
    uint32_t chunk = blockIdx.x * (uint32_t)(kBlock / 64) + (threadIdx.x >> 6);
    uint32_t cur = 0u, end = 0u, region = 0u;      // wave-uniform: next item of the current chunk, its end, its region
    bool more = chunk < total_chunks;
    auto open_chunk = [&]() {
        region = 0u;
#pragma unroll
        for (int j = 1; j < kWfSub; ++j) if (chunk >= first_chunk[j]) region = (uint32_t)j;
        uint32_t c0 = 0u, n = 0u;
#pragma unroll
        for (int j = 0; j < kWfSub; ++j) if (region == (uint32_t)j) { c0 = first_chunk[j]; n = cnt[j]; }
        cur = (chunk - c0) * 64u;
        end = cur + 64u < n ? cur + 64u : n;
    };
    if (more) open_chunk();
    uint32_t* stk = &s_stack[threadIdx.x];
    LaneRay tr;
    lane_ray_begin(tr, splat3(0.0f), splat3(0.0f), 0.0f, 0.0f);
    uint32_t item = 0u, cnt_closest = 0u, cnt_any = 0u;
    bool busy = false;
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        const uint32_t k = (uint32_t)__popcll(idle);
        if (more && k >= refill_min) {
            const uint32_t avail = end - cur, take = k < avail ? k : avail;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!busy && rank < take) {
                const uint32_t idx = cur + rank;
                item = io.items_in ? io.items_in[(size_t)region * icap + idx] : ((region * rcap + idx) << 1);
                const uint32_t slot = item >> 1;
                const uint32_t* w = io.qin + slot;
                const bool owned = (w[2 * cap] & 4u) != 0u;
                if (item & 1u) {
                    lane_ray_begin(tr, mk3(u2f(w[30 * cap]), u2f(w[31 * cap]), u2f(w[32 * cap])), mk3(u2f(w[33 * cap]), u2f(w[34 * cap]), u2f(w[35 * cap])),
                                   u2f(w[36 * cap]), u2f(w[37 * cap]));
                    if (owned) cnt_any++;
                } else {
                    LoopState s;
                    s.pos = mk3(u2f(w[3 * cap]), u2f(w[4 * cap]), u2f(w[5 * cap]));
                    s.ffnormal = mk3(u2f(w[6 * cap]), u2f(w[7 * cap]), u2f(w[8 * cap]));
                    s.next_dir = mk3(u2f(w[15 * cap]), u2f(w[16 * cap]), u2f(w[17 * cap]));
                    lane_ray_begin(tr, bounce_origin(s), s.next_dir, 0.001f, 100.0f);
                    if (owned) cnt_closest++;
                }
                busy = true;
            }
            cur += take;
            if (cur == end) {
                chunk += n_waves;
                more = chunk < total_chunks;
                if (more) open_chunk();
            }
        }
        if (__ballot(busy) == 0ull) {
            if (!more) break;
            continue;
        }
        traverse_round(sc, tr, busy, (item & 1u) != 0u, stk, (uint32_t)kBlock, slice);
        if (busy && tr.cur == 0xFFFFFFFFu) {
            uint32_t* h = io.hits + (item >> 1);
            if (item & 1u) h[6 * cap] = tr.tri == 0xFFFFFFFFu ? 1u : 0u;
            else {
                bool front = false;
                if (tr.tri != 0xFFFFFFFFu) { front = tr.det > 0.0f; if (sc.instances[tr.inst].flip) front = !front; }
                h[0] = f2u(tr.t); h[1 * cap] = f2u(tr.u); h[2 * cap] = f2u(tr.v); h[3 * cap] = tr.tri; h[4 * cap] = tr.inst; h[5 * cap] = front ? 1u : 0u;
            }
            busy = false;
        }
    }
    flush_ray_counters(fv, cnt_closest, cnt_any, s_cnt);
}

template <int STAGE>
__global__ void __launch_bounds__(kBlock, 4) wf_shade_kernel(SceneView sc, FrameView fv, WfPass io, uint32_t depth) {
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_tmp[8];
    const uint32_t region = blockIdx.x % (uint32_t)kWfSub, blk = blockIdx.x / (uint32_t)kWfSub, nblk = gridDim.x / (uint32_t)kWfSub;
    const uint32_t rcap = io.capacity / (uint32_t)kWfSub, icap = 2u * rcap, rbase = region * rcap;
    const uint32_t n = io.n_in[region] < rcap ? io.n_in[region] : rcap;
    if (blk * (uint32_t)kBlock >= n) return;   // uniform per workgroup
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    const size_t cap = io.capacity;
    PathCtx c(sc, fv, &s_stack[threadIdx.x], (uint32_t)kBlock);
    uint32_t cnt_closest = 0u, cnt_any = 0u;
    for (uint32_t base = blk * (uint32_t)kBlock; base < n; base += nblk * (uint32_t)kBlock) {
        const uint32_t local = base + threadIdx.x, slot = rbase + local;
        const bool have = local < n;
        LoopState s;
        s.alive = false;
        ReservoirView r = zero_reservoir();
        uint32_t pix = 0u, flags = 0u;
        bool owned = false, keep = false, want = false, ended = true;
        ShadowReq req;
        req.want = false; req.add_now = false; req.contrib = req.dark = req.o = req.d = splat3(0.0f); req.tmin = req.tmax = 0.0f;
        c.n_closest = 0u; c.n_any = 0u;
        if (have) {
            ContQueue qv; qv.words = io.qin; qv.count = nullptr; qv.capacity = io.capacity; qv.overflow = nullptr; qv.nsub = 1u;
            cont_load(qv, slot, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
            const uint32_t* w = io.qin + slot;
            flags = w[2 * cap];
            const uint32_t* h = io.hits + slot;
            if (flags & WF_SHADOW) {      // the estimate the previous iteration left pending
                const bool lit = h[6 * cap] != 0u;
                const int o = lit ? 38 : 41;
                s.accumulated = s.accumulated + mk3(u2f(w[(size_t)o * cap]), u2f(w[(size_t)(o + 1) * cap]), u2f(w[(size_t)(o + 2) * cap]));
            }
            ended = (flags & WF_ENDED) != 0u;
            if (!ended) {
                HitRec hr;
                hr.t = u2f(h[0]); hr.u = u2f(h[1 * cap]); hr.v = u2f(h[2 * cap]); hr.tri = h[3 * cap]; hr.inst = h[4 * cap]; hr.front = h[5 * cap] != 0u;
                bounce_shade<VARIANT>(c, s, depth, hr, req);
                if (!req.want) s.accumulated = s.accumulated + (req.add_now ? req.contrib : req.dark);
                ended = !s.alive;
                want = req.want;
            }
            keep = want || !ended;
        }
        const uint32_t lo = workgroup_reserve(io.n_out + region, keep, s_tmp);
        const bool fits = keep && lo < rcap;
        const bool overflowed = keep && !fits;
        if (overflowed) {      // the next queue is full: finish the path in place (never dropped)
            if (want) s.accumulated = s.accumulated + (c.any(req.o, req.d, req.tmin, req.tmax) ? req.dark : req.contrib);
            if (!ended) { s.alive = true; path_loop<VARIANT>(c, s, depth + 1u, fv.max_depth); }
        }
        note_queue_overflow(ContQueue{nullptr, nullptr, 0u, io.overflow, 1u}, overflowed);
        const uint32_t slot_out = rbase + lo;
        if (fits) {
            ContQueue qo; qo.words = io.qout; qo.count = nullptr; qo.capacity = io.capacity; qo.overflow = nullptr; qo.nsub = 1u;
            s.alive = true;
            cont_store(qo, slot_out, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
            uint32_t* w = io.qout + slot_out;
            w[2 * cap] = (s.previous_was_diffuse ? 1u : 0u) | (s.is_glass ? 2u : 0u) | (owned ? 4u : 0u) | (want ? WF_SHADOW : 0u) | (ended ? WF_ENDED : 0u);
            if (want) {
                const float f[14] = {req.o.x, req.o.y, req.o.z, req.d.x, req.d.y, req.d.z, req.tmin, req.tmax,
                                     req.contrib.x, req.contrib.y, req.contrib.z, req.dark.x, req.dark.y, req.dark.z};
#pragma unroll
                for (int k = 0; k < 14; ++k) w[(size_t)(30 + k) * cap] = f2u(f[k]);
            }
        }
        // ray items of depth + 1: the next closest-hit ray, the pending shadow ray (two reservations, one atomic each per workgroup)
        const uint32_t ic = workgroup_reserve(io.n_items_out + region, fits && !ended, s_tmp);
        if (fits && !ended && ic < icap) io.items_out[(size_t)region * icap + ic] = slot_out << 1;
        const uint32_t is = workgroup_reserve(io.n_items_out + region, fits && want, s_tmp);
        if (fits && want && is < icap) io.items_out[(size_t)region * icap + is] = (slot_out << 1) | 1u;
        if (have && !fits) finish_path<STAGE>(c, pix, r, s);      // ended with nothing pending, or finished in place
        if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
    }
    flush_ray_counters(fv, cnt_closest, cnt_any, s_cnt);
}

// ---- resident kernels: the BVH lives in LDS ---------------------------------------------------------------------------------------
// Same stages, same arithmetic, another execution shape. One persistent 1024-thread workgroup per CU (16 waves = the 4 waves per SIMD the
// register budget allows anyway) first copies the quantized pair nodes — all of them when they fit, the top of the breadth-first tree
// otherwise — and, when there is room, every triangle slot into LDS (Cornell Box: 25 KB + 62 KB next to 64 KB of traversal stacks), then
// each WAVE takes 16x16 pixel blocks from a global counter (four 8x8 tiles one after the other) until the counter runs out.
//  * a node step is two ds_read_b128 (~100 cycles, no TA / L1 / L2 round trip) instead of four global loads (~500-900 cycles under
//    load); a triangle test three ds_read_b128. The vector-memory pipe keeps the per-pixel streams, shading records and the queues.
//  * waves are independent: no workgroup waits for its slowest wave, and the sweep over the image is the order of the counter.
// Used when the tree is shallow enough for the 16-entry stacks (bvh_depth <= 17); deeper trees take pixel_kernel / continue_kernel.
static constexpr int kResThreads = 1024, kResStack = 16;
struct ResidentArgs {
    uint32_t n_lds;        // pair nodes cached in LDS: [0, n_lds)
    uint32_t tris_lds;     // 1: all triangle slots cached too
    uint32_t* work;        // [0] next tile / chunk, [1] ticket of finished workgroups; both zero between launches
    uint32_t batch;        // 8x8 tiles a wave takes per fetch (4 = a whole 16x16 block; 1 when tiles are scarce)
};
// BVH accessor of the resident kernels: LDS-typed pointers (ds_read_b128, not flat loads) for the cached part, HBM for the rest.
typedef __attribute__((address_space(3))) const uint32_t* lds_u32_ptr;
typedef __attribute__((address_space(3))) const float* lds_f32_ptr;
struct LdsBvh {
    lds_u32_ptr a_lds; lds_u32_ptr b_lds; uint32_t n_lds;    // nodes [0, n_lds) cached in LDS (breadth-first order: the top of the tree)
    const uint4* a_glb; const uint4* b_glb;                  // every node, in HBM
    lds_f32_ptr tris_lds; bool tris_cached;                  // every triangle slot cached in LDS, or not at all
    const float4* tris_glb;
    f3 qmin, qstep;
    __device__ __forceinline__ void node(uint32_t i, uint4& qa, uint4& qb) const {
        if (i < n_lds) {
            lds_u32_ptr pa = a_lds + 4u * i; lds_u32_ptr pb = b_lds + 4u * i;      // 16-byte aligned: one ds_read_b128 each
            qa = make_uint4(pa[0], pa[1], pa[2], pa[3]); qb = make_uint4(pb[0], pb[1], pb[2], pb[3]);
        } else { qa = a_glb[i]; qb = b_glb[i]; }
    }
    __device__ __forceinline__ void tri(uint32_t slot, float4& t0, float4& t1, float4& t2) const {
        if (tris_cached) {
            lds_f32_ptr p = tris_lds + 12u * slot;
            t0 = make_float4(p[0], p[1], p[2], p[3]); t1 = make_float4(p[4], p[5], p[6], p[7]); t2 = make_float4(p[8], p[9], p[10], p[11]);
        } else { const float4* p = tris_glb + (size_t)slot * 3u; t0 = p[0]; t1 = p[1]; t2 = p[2]; }
    }
};
struct ResidentCtx : PathCtx {
    LdsBvh qb;
    __device__ __forceinline__ ResidentCtx(const SceneView& s, const FrameView& f, uint32_t* st, uint32_t sd) : PathCtx(s, f, st, sd) {}
    __device__ __forceinline__ void closest(f3 o, f3 d, float tmin, float tmax, HitRec& h) { n_closest++; trace_q<false>(sc, qb, o, d, tmin, tmax, stk, stride, h); }
    __device__ __forceinline__ bool any(f3 o, f3 d, float tmin, float tmax) { HitRec h; n_any++; trace_q<true>(sc, qb, o, d, tmin, tmax, stk, stride, h); return h.tri != 0xFFFFFFFFu; }
};
// Carves the dynamic LDS block (stacks | nodes A | nodes B | triangles), fills the BVH cache. Ends with a barrier.
__device__ __forceinline__ void resident_setup(const SceneView& sc, const ResidentArgs& ra, uint4* s_dyn, LdsBvh& qb, uint32_t*& stack) {
    stack = reinterpret_cast<uint32_t*>(s_dyn);
    uint4* s_a = s_dyn + (kResStack * kResThreads) / 4;
    uint4* s_b = s_a + ra.n_lds;
    float4* s_tri = reinterpret_cast<float4*>(s_b + ra.n_lds);
    for (uint32_t i = threadIdx.x; i < ra.n_lds; i += (uint32_t)kResThreads) { s_a[i] = sc.qnode_a[i]; s_b[i] = sc.qnode_b[i]; }
    if (ra.tris_lds) for (uint32_t i = threadIdx.x; i < sc.num_tris * 3u; i += (uint32_t)kResThreads) s_tri[i] = sc.tris[i];
    qb.a_lds = (lds_u32_ptr)s_a; qb.b_lds = (lds_u32_ptr)s_b; qb.n_lds = ra.n_lds; qb.a_glb = sc.qnode_a; qb.b_glb = sc.qnode_b;
    qb.tris_lds = (lds_f32_ptr)s_tri; qb.tris_cached = ra.tris_lds != 0u; qb.tris_glb = sc.tris;
    qb.qmin = mk3(sc.qmin[0], sc.qmin[1], sc.qmin[2]); qb.qstep = mk3(sc.qstep[0], sc.qstep[1], sc.qstep[2]);
    __syncthreads();
}
__device__ __forceinline__ uint32_t wave_next(uint32_t* counter) {
    uint32_t v = 0u;
    if ((threadIdx.x & 63u) == 0u) v = atomicAdd(counter, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// The last workgroup of a resident launch re-arms the work counters (and, for a pixel launch, decides the next sweep direction).
__device__ __forceinline__ void resident_finish(const ResidentArgs& ra, const TileOrder* to) {
    if (threadIdx.x != 0u) return;
    if (atomicAdd(&ra.work[1], 1u) != gridDim.x - 1u) return;
    ra.work[0] = 0u; ra.work[1] = 0u;
    if (to && to->st) {
        unsigned long long* sums = reinterpret_cast<unsigned long long*>(to->st + 2);
        const unsigned long long top = atomicExch(&sums[0], 0ull), bottom = atomicExch(&sums[1], 0ull);
        to->st[0] = top <= bottom ? 1u : 0u;   // the sweep should END on the cheaper eighth of the image
    }
}

template <int STAGE>
__global__ void __launch_bounds__(kResThreads, 1) resident_pixel_kernel(SceneView sc, FrameView fv, ContQueue q, uint32_t cut, TileOrder to, uint32_t* zero_counts,
                                                                        ResidentArgs ra) {
    extern __shared__ uint4 s_dyn[];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    if (zero_counts && blockIdx.x == 0u && threadIdx.x <= (uint32_t)kMaxCuts) zero_counts[threadIdx.x] = 0u;
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    uint32_t* stack;
    LdsBvh qb;
    resident_setup(sc, ra, s_dyn, qb, stack);
    ResidentCtx c(sc, fv, stack + threadIdx.x, (uint32_t)kResThreads);
    c.qb = qb;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bx_n = (fv.W + 15u) / 16u, by_n = (fv.y1 - fv.y0 + 15u) / 16u, nblocks = bx_n * by_n;
    const bool flip = to.st && to.st[0] != 0u;
    const uint32_t k8 = by_n / 8u > 0u ? by_n / 8u : 1u;
    uint32_t cnt_closest = 0u, cnt_any = 0u;
    const uint32_t ntiles = nblocks * 4u;
    for (;;) {
        uint32_t t0 = 0u;
        if (lane == 0u) t0 = atomicAdd(&ra.work[0], ra.batch);
        t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0);
        if (t0 >= ntiles) break;
        const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
        const uint32_t blk = t0 >> 2;       // (a batch never crosses a block: batch is 1, 2 or 4 and fetches are batch-aligned)
        uint32_t by = blk / bx_n;
        const uint32_t bx = blk - by * bx_n;
        if (flip) by = by_n - 1u - by;
#pragma nounroll
        for (uint32_t sub = t0 & 3u; sub < (t0 & 3u) + ra.batch; ++sub) {
            const uint32_t px = bx * 16u + (sub & 1u) * 8u + (lane & 7u);
            const uint32_t py = fv.y0 + by * 16u + (sub >> 1) * 8u + (lane >> 3);
            const bool active = px < fv.W && py < fv.y1;
            if (__ballot(active) == 0ull) continue;
            const uint32_t pix = py * fv.W + px;
            const bool counted = active && py >= fv.own_y0 && py < fv.own_y1;
            LoopState s;
            s.alive = false;
            ReservoirView r = zero_reservoir();
            bool traced = false;
            c.n_closest = 0u; c.n_any = 0u;
            if (active) {
                uint32_t seed = 0u;
                if (STAGE == 1) {
                    if (!(fv.gpos[pix].w < 0.0f)) { seed = temporal_seed(fv, pix); traced = true; }
                } else if (spatial_neighbors(c, pix, r)) { seed = r.y; traced = true; }
                if (traced) path_head<VARIANT>(c, pix, seed, s);
            }
            const bool parked = run_segment_and_park<VARIANT>(c, s, 1u, cut < fv.max_depth ? cut : fv.max_depth, q, pix, counted, STAGE == 2 ? &r : nullptr);
            if (traced && !parked) finish_path<STAGE>(c, pix, r, s);
            if (counted) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
        }
        if (to.st && lane == 0u) {      // this batch's time, for the next launch's sweep direction (see report_tile_cost)
            const unsigned long long cost = (__builtin_amdgcn_s_memtime() - t_begin) >> 8;
            unsigned long long* sums = reinterpret_cast<unsigned long long*>(to.st + 2);
            if (by < k8) atomicAdd(&sums[0], cost);
            if (by + k8 >= by_n) atomicAdd(&sums[1], cost);
        }
    }
    flush_ray_counters(fv, cnt_closest, cnt_any, s_cnt);   // (contains a barrier preceded by s_waitcnt vmcnt(0): this workgroup's atomics are performed)
    resident_finish(ra, &to);
}

template <int STAGE>
__global__ void __launch_bounds__(kResThreads, 1) resident_continue_kernel(SceneView sc, FrameView fv, ContQueue qin, ContQueue qout, uint32_t d0, uint32_t d1, ResidentArgs ra) {
    extern __shared__ uint4 s_dyn[];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    const uint32_t filled = *qin.count;
    const uint32_t n = filled < qin.capacity ? filled : qin.capacity;
    uint32_t* stack;
    LdsBvh qb;
    resident_setup(sc, ra, s_dyn, qb, stack);
    ResidentCtx c(sc, fv, stack + threadIdx.x, (uint32_t)kResThreads);
    c.qb = qb;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t cnt_closest = 0u, cnt_any = 0u;
    for (;;) {
        const uint32_t chunk = wave_next(&ra.work[0]);
        if ((unsigned long long)chunk * 64ull >= (unsigned long long)n) break;
        const uint32_t slot_in = chunk * 64u + lane;
        LoopState s;
        s.alive = false;
        ReservoirView r = zero_reservoir();
        uint32_t pix = 0u;
        bool owned = false;
        c.n_closest = 0u; c.n_any = 0u;
        const bool have = slot_in < n;
        if (have) cont_load(qin, slot_in, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
        const bool parked = run_segment_and_park<VARIANT>(c, s, d0, d1, qout, pix, owned, STAGE == 2 ? &r : nullptr);
        if (have && !parked) finish_path<STAGE>(c, pix, r, s);
        if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
    }
    flush_ray_counters(fv, cnt_closest, cnt_any, s_cnt);
    resident_finish(ra, nullptr);
}


// ---- launches ---------------------------------------------------------------------------------------------------------------------------
static dim3 grid_for(const FrameView& fv);
static bool empty_rows(const FrameView& fv);
static ContQueue queue_of(const TraceLaunch& L, uint32_t k);
static uint32_t first_cut(const TraceLaunch& L, const FrameView& fv);
hipError_t launch_compact(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream) {
    if (stage != 1 && stage != 2) return hipErrorInvalidValue;
    if (empty_rows(fv)) return hipSuccess;
    dim3 cgrid((fv.W + 31u) / 32u, (fv.y1 - fv.y0 + 15u) / 16u, 1u), cblock(kBlockC);
    if (stage == 1) hipLaunchKernelGGL(compact_kernel<1>, cgrid, cblock, 0, stream, sc, fv);
    else hipLaunchKernelGGL(compact_kernel<2>, cgrid, cblock, 0, stream, sc, fv);
    return hipGetLastError();
}
static uint32_t resident_lds_bytes(const SceneView& sc, const TraceLaunch& L) {
    return (uint32_t)(kResStack * kResThreads * 4) + L.res_nodes * 32u + (L.res_tris ? sc.num_tris * 48u : 0u);
}
template <class K>
static hipError_t allow_lds(K kernel, uint32_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static hipError_t exp_launch_resident_pixels(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const TraceLaunch& L) {
    const dim3 grid = grid_for(fv);
    {
        TileOrder to{L.tile_state, grid.y};
        const uint32_t lds = resident_lds_bytes(sc, L), ntiles = grid.x * grid.y * 4u, waves = L.num_cus * (uint32_t)(kResThreads / 64);
        uint32_t batch = L.res_batch ? L.res_batch : (ntiles >= 6u * waves ? 2u : 1u);
        if (batch != 1u && batch != 2u && batch != 4u) batch = 1u;
        ResidentArgs ra{L.res_nodes, L.res_tris ? 1u : 0u, L.work, batch};
        const uint32_t wgs = std::max(1u, std::min(L.num_cus, (ntiles / batch + 15u) / 16u));
        hipError_t e = stage == 1 ? allow_lds(resident_pixel_kernel<1>, lds) : allow_lds(resident_pixel_kernel<2>, lds);
        if (e != hipSuccess) return e;
        if (stage == 1) hipLaunchKernelGGL(resident_pixel_kernel<1>, dim3(wgs), dim3(kResThreads), lds, stream, sc, fv, queue_of(L, 0), first_cut(L, fv), to, L.zero_counts, ra);
        else hipLaunchKernelGGL(resident_pixel_kernel<2>, dim3(wgs), dim3(kResThreads), lds, stream, sc, fv, queue_of(L, 0), first_cut(L, fv), to, L.zero_counts, ra);
        return hipGetLastError();
    }
    return hipErrorUnknown;
}

// The continuation launches of a traced stage in one of the experimental forms; false: none applies (the plain continue_kernel launches follow).
static bool exp_launch_continuations(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const TraceLaunch& L, hipError_t& result) {
    auto inner = [&]() -> hipError_t {
    if (L.wavefront && L.ncuts == 1 && L.cuts[0] < fv.max_depth) {
            // counter block (zeroed by the caller before the pixel launch): records of pass k per region at [8 k .. 8 k + 8), items at 512 + 8 k
            const uint32_t rblocks = std::max(1u, (L.capacity / (uint32_t)kWfSub + (uint32_t)kBlock - 1u) / (uint32_t)kBlock);
            const uint32_t sgrid = rblocks * (uint32_t)kWfSub;
            const uint32_t tgrid = std::max(1u, std::min(L.num_cus * 8u, (2u * L.capacity + (uint32_t)kBlock - 1u) / (uint32_t)kBlock));
            for (uint32_t d = L.cuts[0]; d <= fv.max_depth; ++d) {
                const uint32_t k = d - L.cuts[0];
                WfPass io;
                io.qin = L.wf_words[k & 1u]; io.n_in = L.counts + 8u * k; io.qout = L.wf_words[(k + 1u) & 1u]; io.n_out = L.counts + 8u * (k + 1u);
                io.items_in = k == 0u ? nullptr : L.wf_items[k & 1u]; io.n_items_in = L.counts + 512 + 8u * k;
                io.items_out = L.wf_items[(k + 1u) & 1u]; io.n_items_out = L.counts + 512 + 8u * (k + 1u);
                io.hits = L.wf_hits; io.capacity = L.capacity; io.overflow = L.overflow;
                if (sc.bvh_depth <= 17u) hipLaunchKernelGGL(wf_trace_kernel<16>, dim3(tgrid), dim3(kBlock), 0, stream, sc, fv, io, L.refill_min, (int)L.slice);
                else hipLaunchKernelGGL(wf_trace_kernel<32>, dim3(tgrid), dim3(kBlock), 0, stream, sc, fv, io, L.refill_min, (int)L.slice);
                if (stage == 1) hipLaunchKernelGGL(wf_shade_kernel<1>, dim3(sgrid), dim3(kBlock), 0, stream, sc, fv, io, d);
                else hipLaunchKernelGGL(wf_shade_kernel<2>, dim3(sgrid), dim3(kBlock), 0, stream, sc, fv, io, d);
            }
            return hipGetLastError();
        }
        if (L.stream && L.ncuts == 1 && L.cuts[0] < fv.max_depth) {
            const uint32_t wgs = std::max(1u, std::min(L.num_cus * 4u, (L.capacity + (uint32_t)kBlock - 1u) / (uint32_t)kBlock));
            if (stage == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(wgs), dim3(kBlock), 0, stream, sc, fv, queue_of(L, 0), L.counts + 1, L.cuts[0], L.refill_min, L.shade_min, (int)L.slice);
            else hipLaunchKernelGGL(stream_kernel<2>, dim3(wgs), dim3(kBlock), 0, stream, sc, fv, queue_of(L, 0), L.counts + 1, L.cuts[0], L.refill_min, L.shade_min, (int)L.slice);
            return hipGetLastError();
        }
        if (L.refill && L.ncuts == 1 && L.cuts[0] < fv.max_depth) {
            // one bounce kernel with lane refill instead of the continuation launches: persistent waves, 4 workgroups per CU
            const uint32_t wgs = std::max(1u, std::min(L.num_cus * 4u, (L.capacity + (uint32_t)kBlock - 1u) / (uint32_t)kBlock));
            if (stage == 1) hipLaunchKernelGGL(bounce_kernel<1>, dim3(wgs), dim3(kBlock), 0, stream, sc, fv, queue_of(L, 0), L.counts + 1, L.cuts[0], L.refill_min);
            else hipLaunchKernelGGL(bounce_kernel<2>, dim3(wgs), dim3(kBlock), 0, stream, sc, fv, queue_of(L, 0), L.counts + 1, L.cuts[0], L.refill_min);
            return hipGetLastError();
        }
        if (L.resident) {
            for (uint32_t k = 0; k < L.ncuts && L.cuts[k] < fv.max_depth; ++k) {
                const uint32_t d0 = L.cuts[k], d1 = (k + 1 < L.ncuts && L.cuts[k + 1] < fv.max_depth) ? L.cuts[k + 1] : fv.max_depth;
            ResidentArgs ra{L.res_nodes, L.res_tris ? 1u : 0u, L.work + 2u * (3u + k), 1u};      // behind the three pixel-launch pairs (L.work = the stage's slot 0)
                const uint32_t lds = resident_lds_bytes(sc, L);
                const uint32_t wgs = std::max(1u, std::min(L.num_cus, (L.capacity + 1023u) / 1024u));
                hipError_t e = stage == 1 ? allow_lds(resident_continue_kernel<1>, lds) : allow_lds(resident_continue_kernel<2>, lds);
                if (e != hipSuccess) return e;
                if (stage == 1) hipLaunchKernelGGL(resident_continue_kernel<1>, dim3(wgs), dim3(kResThreads), lds, stream, sc, fv, queue_of(L, k), queue_of(L, k + 1), d0, d1, ra);
                else hipLaunchKernelGGL(resident_continue_kernel<2>, dim3(wgs), dim3(kResThreads), lds, stream, sc, fv, queue_of(L, k), queue_of(L, k + 1), d0, d1, ra);
            }
            return hipGetLastError();
        }
        return hipErrorNotReady;      // (marker: nothing applied)
    };
    result = inner();
    return result != hipErrorNotReady;
}
// How much of the scene's BVH a resident launch can keep in LDS next to its stacks: (nodes, all triangles?). (0, false): not resident.
void resident_plan(const SceneView& sc, uint32_t& nodes, bool& tris) {
    nodes = 0u; tris = false;
    if (sc.bvh_depth > (uint32_t)kResStack + 1u || !sc.qnode_a) return;
    const uint32_t budget = 163840u - 256u - (uint32_t)(kResStack * kResThreads * 4);
    if (sc.num_nodes * 32u + sc.num_tris * 48u <= budget) { nodes = sc.num_nodes; tris = true; return; }
    nodes = std::min(sc.num_nodes, budget / 32u);
}

