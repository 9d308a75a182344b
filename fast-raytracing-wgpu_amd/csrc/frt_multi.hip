// frt_multi.hip — N GPUs behind ONE call: frt_multi_renderer_* (include/frt.h).
//
// In the reference one call renders one frame (Renderer::render, src/renderer.rs:349-518, called from State::render, src/state.rs:192-204).
// This is that call for a node with several GPUs: ONE process, `ndev` strip renderers (frt_renderer with a row range, two-stream schedule),
// the scene replicated on every device, the frame cut into horizontal strips of equal WORK, and the three per-frame halo exchanges of
// DESIGN.md §8 as peer copies (hipMemcpyPeerAsync over xGMI; a plain device-to-device copy when two strips share a device):
//   "post"  1 row (K + 1 with a motion halo) of the previous frame's accumulation  -> the neighbour's halo rows, a whole frame ahead of its use
//   "pre"   K rows of the previous frame's spatial reservoirs, before T-merge (moving camera only)
//   "mid"   12 rows of this frame's temporal reservoirs, between T-merge and the spatial stage's EDGE rows (overlaps the interior rows)
// The same six steps, in the same order and against the same streams, as frt/dist.py::render_strip_frame (the torch.distributed / RCCL form
// of the loop); here every rank is a device of this process and the orderings are events:
//   a copy runs on the DESTINATION strip's copy stream, behind (a) the event the SOURCE strip recorded after the kernel that produced the
//   rows and (b) an event of the destination's own main stream that lies behind the last reader of the halo rows it overwrites;
//   the consumer stream (main stream, or the edge stream for "mid") then waits for the copy's event;
//   (c) the SOURCE strip's next writer of the rows waits for the copy's event too (write-after-read): T-merge(f+1) rewrites the temporal
//   reservoirs the neighbours' "mid" copies of frame f read, the spatial stage of frame f rewrites the spatial reservoirs their "pre" copies of
//   frame f read, post(f+1) rewrites the accumulation slot their "post" copies read. With balanced strips on healthy devices the copies are long
//   done by then; a device that lags (shared, throttled) would otherwise have its neighbour's rows overwritten under the copy.
// Reads gather the strips' rows. Images are bit-identical to a single renderer's (tests: every logical device mapped to ordinal 0).
// Host side: ONE WORKER THREAD PER STRIP enqueues that strip's launches, copies and event operations (about 85 us of host time per strip and
// frame: eight strips from one thread would take 0.68 ms per frame against the 0.36 ms a 1/8 strip of a 1080p frame needs on its GPU —
// tools/host_enqueue.py). A frame is two steps with a host barrier between them, so that every event is recorded before a neighbour's
// thread makes a stream wait for it (hipStreamWaitEvent on an event that has not been recorded yet would not wait at all):
//   step A  incoming "post" / "pre" copies (their source events were recorded in the previous frame) | G-buffer + T-trace + T-merge | record ev_tm
//   step B  incoming "mid" copies | spatial interior | edge stream waits for the copies | spatial edge + continuations | post | records | end of frame
// frt_multi_renderer_render returns when every strip's frame is ENQUEUED (the GPUs run behind, as with frt_renderer_render).
#include "frt_scene.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace frt { int set_error(int code, const std::string& msg); }
using frt::set_error;

#define HIPM_TRY(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return set_error(FRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static const uint32_t kHaloReservoir = 12;   // spatial radius 10 + 2 rows of redundant spatial work (frt_renderer.hip: kHaloGbuffer)
static const uint32_t kHaloHistory = 1;      // post reads the previous accumulation within +-1 row (post.wgsl:196-199)

struct Strip {
    int device = 0;
    uint32_t rb = 0, re = 0;
    frt_renderer* r = nullptr;
    hipStream_t copy = nullptr;                       // this strip's incoming halo rows
    hipEvent_t ev_tm = nullptr, ev_spatial = nullptr, ev_post = nullptr;      // recorded on the main stream behind T-merge / spatial / post of the current frame
    hipEvent_t ev_copy_pre = nullptr, ev_copy_mid = nullptr, ev_copy_post = nullptr;
    hipEvent_t ev_src = nullptr, ev_gather = nullptr; // gather: behind the strip's producers / behind the copy of its rows into the gathered frame
    bool gather_pending = false;                      // a gather copy of this strip's rows may still be reading them: the next writers wait for ev_gather
    void* p_res[2] = {nullptr, nullptr};              // device addresses of reservoir_buffers[0 / 1] and accumulation[0 / 1]: fixed for the renderer's life, so a
    void* p_acc[2] = {nullptr, nullptr};              // neighbour's worker thread reads them here instead of through the renderer handle
    std::thread worker;
    int status = FRT_OK; std::string message;         // of the last step this strip's worker ran
};

struct frt_multi_renderer {
    uint32_t W = 0, H = 0, motion_halo = 0;
    std::vector<Strip> strips;
    std::vector<uint32_t> bounds;
    uint32_t frame = 0;              // frames rendered since create / reset (== every strip's frame_count)
    uint64_t serial = 0;             // frames rendered since create (never reset)
    // worker threads: `job` counts the steps posted so far (two per frame); a worker runs step `job` when it sees the counter move
    std::mutex mu; std::condition_variable cv_post, cv_done;
    uint64_t job = 0; uint32_t done = 0; bool stop = false;
    int step = 0;                    // the step the posted job runs (0 = A, 1 = B): stated, not derived from the job counter's parity, so that a frame
                                     // that ended after a failed step A cannot shift every later frame by one step
    bool failed = false;             // a strip's step failed: half-enqueued frames on the other strips; FRT_ERR_STATE until frt_multi_renderer_clear
    int inject_strip = -1, inject_step = -1;      // frt_multi_renderer_inject_failure (testing)
    float jitter[2] = {0.0f, 0.0f};
    uint32_t peer_pairs = 0, peer_enabled = 0;
    void* gather_buf = nullptr; size_t gather_bytes = 0;      // read_display / read_buffer: the gathered frame on the first strip's device
    frt_camera_uniform cam{};
};

namespace {
struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Strip boundaries of equal WORK (the ceiling strip of a Cornell Box is far cheaper than the floor strip): a quarter-resolution probe
// rendered band by band, exact device ray counters per band, cost = rays + 4 per pixel (frt/dist.py::balanced_boundaries, same numbers).
int balanced_boundaries(const frt_scene* scene, uint32_t W, uint32_t H, uint32_t world, uint32_t max_depth, int device, uint32_t min_rows, std::vector<uint32_t>& out) {
    out.assign(1, 0u);
    if (world == 1) { out.push_back(H); return FRT_OK; }
    const uint32_t bands = 36, probe_frames = 2;
    const uint32_t pw = std::max(W / 4u, 16u), ph = std::max(H / 4u, bands);
    std::vector<double> cost(bands, 0.0);
    uint32_t counts[8];
    if (frt_scene_counts(scene, counts) != FRT_OK) return FRT_ERR_INVALID_ARG;
    for (uint32_t b = 0; b < bands; ++b) {
        const uint32_t y0 = ph * b / bands, y1 = ph * (b + 1) / bands;
        frt_render_opts o{};
        o.max_depth = max_depth; o.device = device; o.row_begin = y0; o.row_end = y1;
        frt_renderer* r = frt_renderer_create(scene, pw, ph, &o);
        if (!r) return FRT_ERR_HIP;
        for (uint32_t f = 0; f < probe_frames; ++f) {
            frt_camera_uniform cam;
            frt_camera_default((float)W / (float)H, f, counts[3], &cam);
            const int rc = frt_renderer_render(r, &cam);
            if (rc) { frt_renderer_destroy(r); return rc; }
        }
        frt_stats st;
        const int rc = frt_renderer_stats(r, &st);
        frt_renderer_destroy(r);
        if (rc) return rc;
        cost[b] = (double)(st.rays_closest + st.rays_any) + 4.0 * pw * (y1 - y0) * probe_frames;
    }
    std::vector<double> cum(H + 1, 0.0);
    for (uint32_t b = 0; b < bands; ++b) {
        const uint32_t y0 = H * b / bands, y1 = H * (b + 1) / bands;
        for (uint32_t y = y0; y < y1; ++y) cum[y + 1] = cost[b] / std::max(y1 - y0, 1u);
    }
    for (uint32_t y = 0; y < H; ++y) cum[y + 1] += cum[y];
    for (uint32_t k = 1; k < world; ++k) {
        const double target = cum[H] * k / world;
        uint32_t y = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        y = std::max(y, out.back() + min_rows);                         // every strip at least as tall as the halo
        y = std::min(y, H - min_rows * (world - k));
        out.push_back(y);
    }
    out.push_back(H);
    return FRT_OK;
}

uint32_t bpp_of_buf(int buf) {
    switch (buf) {
    case FRT_BUF_GPOS: case FRT_BUF_GNORMAL: case FRT_BUF_ACCUM: case FRT_BUF_CANDIDATE: return 16u;
    case FRT_BUF_GALBEDO: case FRT_BUF_DISPLAY: return 4u;
    case FRT_BUF_GMOTION: case FRT_BUF_RAW: return 8u;
    case FRT_BUF_RESERVOIR: return 32u;
    }
    return 0u;
}

// rows [y0, y1) of `buf`[index]: from strip `src` into the same image rows of strip `dst`, on dst's copy stream
int copy_rows(frt_multi_renderer* m, Strip& src, Strip& dst, int buf, int index, uint32_t y0, uint32_t y1) {
    const uint32_t bpp = bpp_of_buf(buf);
    const void* ps = buf == FRT_BUF_RESERVOIR ? src.p_res[index & 1] : src.p_acc[index & 1];
    void* pd = buf == FRT_BUF_RESERVOIR ? dst.p_res[index & 1] : dst.p_acc[index & 1];
    const size_t pitch = (size_t)m->W * bpp, off = pitch * y0, bytes = pitch * (y1 - y0);
    if (src.device == dst.device) HIPM_TRY(hipMemcpyAsync((uint8_t*)pd + off, (const uint8_t*)ps + off, bytes, hipMemcpyDeviceToDevice, dst.copy));
    else HIPM_TRY(hipMemcpyPeerAsync((uint8_t*)pd + off, dst.device, (const uint8_t*)ps + off, src.device, bytes, dst.copy));
    return FRT_OK;
}

// One exchange with both vertical neighbours: strip k receives `rows` rows from each neighbour into the rows just outside its own.
// src_ev(strip): the event behind the kernel that produced the rows; dst_ev(strip): an event of the receiver's main stream behind the last
// reader of the halo rows; done(strip): the event recorded behind the strip's incoming copies.
enum Which { PRE, MID, POST };
// Strip k's INCOMING rows of one exchange (run by strip k's worker).
int exchange_into(frt_multi_renderer* m, size_t k, Which which, int buf, int index, uint32_t rows) {
    const size_t n = m->strips.size();
    Strip& d = m->strips[k];
    DevGuard g(d.device);
    hipEvent_t own = which == MID ? d.ev_tm : (which == PRE ? d.ev_spatial : d.ev_post);
    hipEvent_t done = which == MID ? d.ev_copy_mid : (which == PRE ? d.ev_copy_pre : d.ev_copy_post);
    HIPM_TRY(hipStreamWaitEvent(d.copy, own, 0));
    for (int side = 0; side < 2; ++side) {
        if ((side == 0 && k == 0) || (side == 1 && k + 1 == n)) continue;
        Strip& s = m->strips[side == 0 ? k - 1 : k + 1];
        hipEvent_t produced = which == MID ? s.ev_tm : (which == PRE ? s.ev_spatial : s.ev_post);
        HIPM_TRY(hipStreamWaitEvent(d.copy, produced, 0));
        // the upper neighbour's last `rows` rows land in [rb - rows, rb); the lower neighbour's first `rows` rows in [re, re + rows)
        const uint32_t y0 = side == 0 ? d.rb - rows : d.re, y1 = side == 0 ? d.rb : d.re + rows;
        const int rc = copy_rows(m, s, d, buf, index, y0, y1);
        if (rc) return rc;
    }
    HIPM_TRY(hipEventRecord(done, d.copy));
    return FRT_OK;
}

// The two steps of a frame for strip k (see the head of this file). m->frame / m->serial are those of the frame being enqueued.
int strip_step(frt_multi_renderer* m, size_t k, int step) {
    Strip& s = m->strips[k];
    const size_t n = m->strips.size();
    const frt_camera_uniform* cam = &m->cam;
    const uint32_t K = m->motion_halo;
    // A previous frame's spatial reservoirs exist once ANY frame has been rendered, whatever frame_count says: the reference's host resets
    // frame_count to 0 on every frame the camera moves (state.rs:152) and T-merge reprojects into the previous reservoirs all the same
    // (restir.wgsl:846-900 uses frame_count for the seed only). Post ignores its history at frame_count 0 (post.wgsl:187).
    const bool prev_pre = m->serial > 0, prev_post = m->serial > 0 && m->frame > 0;
    if ((int)k == m->inject_strip && step == m->inject_step) return set_error(FRT_ERR_HIP, "injected failure (frt_multi_renderer_inject_failure)");
    DevGuard g(s.device);
    hipStream_t q = (hipStream_t)frt_renderer_stream(s.r, 0);
    hipStream_t qe = (hipStream_t)frt_renderer_stream(s.r, 2);
    // rule (c): this strip's next writer of rows a neighbour copies waits for that neighbour's copy event
    auto wait_neighbours = [&](hipStream_t st, hipEvent_t Strip::*ev) -> int {
        if (k > 0) HIPM_TRY(hipStreamWaitEvent(st, m->strips[k - 1].*ev, 0));
        if (k + 1 < n) HIPM_TRY(hipStreamWaitEvent(st, m->strips[k + 1].*ev, 0));
        return FRT_OK;
    };
    int rc;
    // rows a gather may still be copying (frt_multi_renderer_gather with a caller's stream): this frame's first writers wait for the copy
    auto wait_gather = [&]() -> int {
        if (!s.gather_pending) return FRT_OK;
        HIPM_TRY(hipStreamWaitEvent(q, s.ev_gather, 0));
        if (qe != q) HIPM_TRY(hipStreamWaitEvent(qe, s.ev_gather, 0));
        s.gather_pending = false;
        return FRT_OK;
    };
    if (step == 0) {
        if ((rc = wait_gather())) return rc;      // (T-merge rewrites the temporal reservoirs)
        if (prev_post) {
            rc = exchange_into(m, k, POST, FRT_BUF_ACCUM, (int)((m->frame - 1u) & 1u), K ? K + kHaloHistory : kHaloHistory);      // a whole frame of slack
            if (rc) return rc;
        }
        if (prev_pre && K) {
            rc = exchange_into(m, k, PRE, FRT_BUF_RESERVOIR, 1, K);
            if (rc) return rc;
            HIPM_TRY(hipStreamWaitEvent(q, s.ev_copy_pre, 0));
        }
        // T-merge(f) rewrites reservoir_buffers[0]: behind the neighbours' "mid" copies of frame f-1 (recorded in that frame's step B)
        if (m->serial > 0 && (rc = wait_neighbours(q, &Strip::ev_copy_mid))) return rc;
        rc = frt_renderer_render_phases(s.r, cam, FRT_PHASE_GBUFFER | FRT_PHASE_TEMPORAL);      // T-merge (G-buffer + T-trace normally ran ahead of the frame)
        if (rc) return rc;
        HIPM_TRY(hipEventRecord(s.ev_tm, q));
        return FRT_OK;
    }
    rc = exchange_into(m, k, MID, FRT_BUF_RESERVOIR, 0, kHaloReservoir);      // behind this strip's and its neighbours' T-merge
    if (rc) return rc;
    // the spatial stage rewrites reservoir_buffers[1] (interior launch: main stream; edge launches: the edge stream, whose second edge stream is
    // ordered behind it): behind the neighbours' "pre" copies of THIS frame (enqueued in step A; the host barrier between the steps recorded them)
    if (prev_pre && K) {
        if ((rc = wait_neighbours(q, &Strip::ev_copy_pre))) return rc;
        if (qe != q && (rc = wait_neighbours(qe, &Strip::ev_copy_pre))) return rc;
    }
    if ((rc = wait_gather())) return rc;      // ... and raw / spatial reservoirs / display rows a gather may still be copying
    rc = frt_renderer_render_phases(s.r, cam, FRT_PHASE_SPATIAL_INNER);       // interior rows: need nothing from a neighbour
    if (rc) return rc;
    HIPM_TRY(hipStreamWaitEvent(qe, s.ev_copy_mid, 0));      // the edge rows' stream waits for the neighbours' reservoirs
    rc = frt_renderer_render_phases(s.r, cam, FRT_PHASE_SPATIAL_EDGE);
    if (rc) return rc;
    HIPM_TRY(hipEventRecord(s.ev_spatial, q));
    if (prev_post) {
        HIPM_TRY(hipStreamWaitEvent(q, s.ev_copy_post, 0));
        // post(f) rewrites the accumulation slot the neighbours' "post" copies of frame f-1 read; their copies of THIS frame (step A, same copy
        // streams, behind those) are what the events now stand for: waiting for them covers both
        if ((rc = wait_neighbours(q, &Strip::ev_copy_post))) return rc;
    }
    rc = frt_renderer_render_phases(s.r, cam, FRT_PHASE_POST);
    if (rc) return rc;
    HIPM_TRY(hipEventRecord(s.ev_post, q));
    return frt_renderer_end_frame(s.r);
}

void worker_main(frt_multi_renderer* m, size_t k) {
    uint64_t seen = 0;
    for (;;) {
        int step;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_post.wait(lk, [&] { return m->stop || m->job != seen; });
            if (m->stop) return;
            seen = m->job;
            step = m->step;
        }
        const int rc = strip_step(m, k, step);
        Strip& s = m->strips[k];
        s.status = rc;
        if (rc) s.message = frt_last_error();      // (thread-local in the library: copy it out of this thread)
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (++m->done == m->strips.size()) m->cv_done.notify_all();
        }
    }
}

// Run one step on every strip's worker and wait until all have enqueued it.
int run_step(frt_multi_renderer* m, int step) {
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->done = 0; m->step = step; m->job += 1;
    }
    m->cv_post.notify_all();
    {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv_done.wait(lk, [&] { return m->done == m->strips.size(); });
    }
    for (Strip& s : m->strips) if (s.status) return set_error(s.status, s.message);
    return FRT_OK;
}
}   // namespace

extern "C" {

frt_multi_renderer* frt_multi_renderer_create(const frt_scene* scene, uint32_t width, uint32_t height, uint32_t ndev, const int32_t* devices, const frt_render_opts* opts) {
    if (!scene || width == 0 || height == 0 || ndev == 0) { set_error(FRT_ERR_INVALID_ARG, "multi_renderer_create: bad arguments"); return nullptr; }
    const int have = frt_device_count();
    if (have < 1) { set_error(FRT_ERR_NO_DEVICE, "no HIP device: this library has no CPU rendering path"); return nullptr; }
    std::vector<int> dev(ndev);
    for (uint32_t k = 0; k < ndev; ++k) {
        dev[k] = devices ? devices[k] : (int)k;
        if (dev[k] < 0 || dev[k] >= have) { set_error(FRT_ERR_INVALID_ARG, "multi_renderer_create: device ordinal out of range"); return nullptr; }
    }
    const uint32_t max_depth = (opts && opts->max_depth) ? opts->max_depth : 8u;
    const uint32_t K = opts ? opts->motion_halo_rows : 0u;
    const uint32_t need = std::max(kHaloReservoir, K + kHaloHistory);
    if (ndev > 1 && height / ndev < need) { set_error(FRT_ERR_INVALID_ARG, "multi_renderer_create: strips would be thinner than the halo"); return nullptr; }
    frt_multi_renderer* m = new frt_multi_renderer();
    m->W = width; m->H = height; m->motion_halo = K;
    if (balanced_boundaries(scene, width, height, ndev, max_depth, dev[0], need, m->bounds) != FRT_OK) {
        m->bounds.clear();      // equal strips are always valid; balancing is an optimisation
        for (uint32_t k = 0; k <= ndev; ++k) m->bounds.push_back((uint32_t)((uint64_t)height * k / ndev));
    }
    m->strips.resize(ndev);
    for (uint32_t k = 0; k < ndev; ++k) {
        Strip& s = m->strips[k];
        s.device = dev[k]; s.rb = m->bounds[k]; s.re = m->bounds[k + 1];
        frt_render_opts o{};
        o.max_depth = max_depth; o.device = dev[k]; o.flags = FRT_FLAG_PIPELINE | (opts ? (opts->flags & FRT_FLAG_TIMING) : 0u);
        o.queue_capacity = opts ? opts->queue_capacity : 0u;
        if (ndev > 1) { o.row_begin = s.rb; o.row_end = s.re; o.motion_halo_rows = K; }
        s.r = frt_renderer_create(scene, width, height, &o);
        if (!s.r) { frt_multi_renderer_destroy(m); return nullptr; }
        DevGuard g(s.device);
        bool ok = hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking) == hipSuccess;
        for (hipEvent_t* e : {&s.ev_tm, &s.ev_spatial, &s.ev_post, &s.ev_copy_pre, &s.ev_copy_mid, &s.ev_copy_post, &s.ev_src, &s.ev_gather})
            ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        if (!ok) { set_error(FRT_ERR_HIP, "multi_renderer_create: stream / event creation failed"); frt_multi_renderer_destroy(m); return nullptr; }
        uint32_t bpp = 0;
        for (int i = 0; i < 2; ++i)
            if (frt_renderer_buffer_info(s.r, FRT_BUF_RESERVOIR, i, &s.p_res[i], &bpp) != FRT_OK || frt_renderer_buffer_info(s.r, FRT_BUF_ACCUM, i, &s.p_acc[i], &bpp) != FRT_OK) {
                frt_multi_renderer_destroy(m); return nullptr;
            }
    }
    if (ndev > 1) for (uint32_t k = 0; k < ndev; ++k) m->strips[k].worker = std::thread(worker_main, m, (size_t)k);
    // peer access between neighbouring strips on different devices (hipMemcpyPeerAsync works without it, through a staging buffer; with it
    // the rows go straight over xGMI)
    for (uint32_t k = 0; k + 1 < ndev; ++k) {
        const int a = dev[k], b = dev[k + 1];
        if (a == b) continue;
        m->peer_pairs += 1;
        int can = 0, both = 0;
        auto enable = [&](int from, int to) {
            if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) return;
            DevGuard g(from);
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) both += 1;
        };
        enable(a, b); enable(b, a);
        if (both == 2) m->peer_enabled += 1;
        (void)hipGetLastError();      // "already enabled" is not an error here
    }
    return m;
}

void frt_multi_renderer_destroy(frt_multi_renderer* m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->stop = true;
    }
    m->cv_post.notify_all();
    for (Strip& s : m->strips) if (s.worker.joinable()) s.worker.join();
    for (Strip& s : m->strips) {      // every strip idle before any strip's buffers go away (a neighbour's incoming copy reads them)
        DevGuard g(s.device);
        if (s.r) (void)frt_renderer_sync(s.r);
        if (s.copy) (void)hipStreamSynchronize(s.copy);
    }
    for (Strip& s : m->strips) {
        DevGuard g(s.device);
        if (s.copy) (void)hipStreamDestroy(s.copy);
        for (hipEvent_t e : {s.ev_tm, s.ev_spatial, s.ev_post, s.ev_copy_pre, s.ev_copy_mid, s.ev_copy_post, s.ev_src, s.ev_gather}) if (e) (void)hipEventDestroy(e);
        if (s.r) frt_renderer_destroy(s.r);
    }
    if (m->gather_buf && !m->strips.empty()) { DevGuard g(m->strips[0].device); (void)hipFree(m->gather_buf); }
    delete m;
}

// Renderer::render, src/renderer.rs:349 — ONE call, one frame, on every device. Asynchronous on the GPUs (returns when the frame is enqueued).
int frt_multi_renderer_render(frt_multi_renderer* m, const frt_camera_uniform* cam) {
    if (!m || !cam) return set_error(FRT_ERR_INVALID_ARG, "multi render: null");
    if (m->failed) return set_error(FRT_ERR_STATE, "multi render: an earlier frame failed on one strip in the middle of its steps; call frt_multi_renderer_clear");
    if (m->strips.size() == 1) {
        Strip& s0 = m->strips[0];
        if (s0.gather_pending) {      // a gather on the caller's stream may still be copying the rows this frame rewrites
            DevGuard g(s0.device);
            HIPM_TRY(hipStreamWaitEvent((hipStream_t)frt_renderer_stream(s0.r, 0), s0.ev_gather, 0));
            s0.gather_pending = false;
        }
        int rc = m->inject_strip == 0 ? set_error(FRT_ERR_HIP, "injected failure (frt_multi_renderer_inject_failure)") : frt_renderer_render(s0.r, cam);
        m->inject_strip = m->inject_step = -1;
        if (rc == FRT_OK) { m->frame += 1; m->serial += 1; }
        else if (rc == FRT_ERR_HIP) m->failed = true;
        return rc;
    }
    m->cam = *cam;
    // Step A on every strip; the return is the host barrier: every ev_tm is recorded. A strip that fails leaves the others with an open, half-enqueued
    // frame (their T-merge is in flight, step B will never run): the handle is FAILED until frt_multi_renderer_clear.
    int rc = run_step(m, 0);
    if (rc == FRT_OK) rc = run_step(m, 1);
    m->inject_strip = m->inject_step = -1;
    if (rc) { m->failed = true; return rc; }
    m->frame += 1; m->serial += 1;
    return FRT_OK;
}

int frt_multi_renderer_sync(frt_multi_renderer* m) {
    if (!m) return set_error(FRT_ERR_INVALID_ARG, "multi sync: null");
    for (Strip& s : m->strips) {
        int rc = frt_renderer_sync(s.r);
        if (rc) return rc;
        DevGuard g(s.device);
        HIPM_TRY(hipStreamSynchronize(s.copy));
    }
    return FRT_OK;
}

uint32_t frt_multi_renderer_frame_count(const frt_multi_renderer* m) { return m ? m->frame : 0u; }

int frt_multi_renderer_reset(frt_multi_renderer* m) {      // frame_count = 0 (state.rs:152): buffers keep their contents
    if (!m) return set_error(FRT_ERR_INVALID_ARG, "multi reset: null");
    if (m->failed) return set_error(FRT_ERR_STATE, "multi reset: the handle is failed; call frt_multi_renderer_clear");
    for (Strip& s : m->strips) { int rc = frt_renderer_reset(s.r); if (rc) return rc; }
    m->frame = 0;
    return FRT_OK;
}

// Back to the state right after create, on every strip; the way out of the failed state. Waits for the devices first: no copy or kernel of a
// half-enqueued frame is in flight afterwards, and every event that a later frame waits for has either completed or is recorded anew before.
int frt_multi_renderer_clear(frt_multi_renderer* m) {
    if (!m) return set_error(FRT_ERR_INVALID_ARG, "multi clear: null");
    int first = FRT_OK;
    for (Strip& s : m->strips) {
        DevGuard g(s.device);
        (void)hipStreamSynchronize(s.copy);
        const int rc = frt_renderer_clear(s.r);      // syncs the strip's streams, zeroes its targets and counters, closes an open frame, clears `failed`
        if (rc && !first) first = rc;
        (void)hipStreamSynchronize(s.copy);
        s.gather_pending = false;
        s.status = FRT_OK; s.message.clear();
    }
    if (first) return first;      // (a device that does not come back: the handle stays failed)
    m->frame = 0; m->serial = 0; m->failed = false; m->inject_strip = m->inject_step = -1;
    return FRT_OK;
}

int frt_multi_renderer_set_jitter(frt_multi_renderer* m, float jx, float jy) {
    if (!m) return set_error(FRT_ERR_INVALID_ARG, "multi set_jitter: null");
    if (m->strips.size() > 1 && (jx != 0.0f || jy != 0.0f))
        return set_error(FRT_ERR_INVALID_ARG, "multi set_jitter: a non-zero post jitter (bilinear taps with Repeat addressing read the opposite image edge) is not supported when the frame is cut into strips");
    m->jitter[0] = jx; m->jitter[1] = jy;
    return frt_renderer_set_jitter(m->strips[0].r, jx, jy);
}

int frt_multi_renderer_inject_failure(frt_multi_renderer* m, uint32_t strip, int step) {
    if (!m || strip >= m->strips.size() || (step != 0 && step != 1)) return set_error(FRT_ERR_INVALID_ARG, "multi inject_failure: bad arguments");
    m->inject_strip = (int)strip; m->inject_step = step;
    return FRT_OK;
}

int frt_multi_renderer_peer_access(const frt_multi_renderer* m, uint32_t out[2]) {
    if (!m || !out) return set_error(FRT_ERR_INVALID_ARG, "multi peer_access: null");
    out[0] = m->peer_pairs; out[1] = m->peer_enabled;
    return FRT_OK;
}

int frt_multi_renderer_boundaries(const frt_multi_renderer* m, uint32_t* out) {
    if (!m || !out) return set_error(FRT_ERR_INVALID_ARG, "multi boundaries: null");
    memcpy(out, m->bounds.data(), m->bounds.size() * sizeof(uint32_t));
    return (int)m->strips.size();
}

// Device-side gather: every strip's own rows of a target into ONE full-frame buffer in the memory of `device`. Each strip's rows travel on that
// strip's copy stream (peer copy over xGMI; a plain device-to-device copy on the same device), behind an event recorded on the strip's main stream
// after frt_renderer_fence (= behind everything the strip has enqueued, the ahead stream included). ev_gather of a strip stands for "my rows have
// been copied": the caller's stream waits for all of them, and so do the strip's next writers of those rows (strip_step, gather_pending).
int frt_multi_renderer_gather(frt_multi_renderer* m, int buf, int index, int32_t device, void* dst, void* stream) {
    const uint32_t bpp = bpp_of_buf(buf);
    if (!m || !dst || !bpp) return set_error(FRT_ERR_INVALID_ARG, "multi gather: bad arguments");
    if (device < 0 || device >= frt_device_count()) return set_error(FRT_ERR_INVALID_ARG, "multi gather: device ordinal out of range");
    if (m->failed) return set_error(FRT_ERR_STATE, "multi gather: the handle is failed; call frt_multi_renderer_clear");
    const size_t pitch = (size_t)m->W * bpp;
    for (Strip& s : m->strips) {
        void* src = nullptr;
        int rc = frt_renderer_buffer_info(s.r, buf, index, &src, nullptr);
        if (rc) return rc;
        rc = frt_renderer_fence(s.r);
        if (rc) return rc;
        DevGuard g(s.device);
        hipStream_t q = (hipStream_t)frt_renderer_stream(s.r, 0);
        HIPM_TRY(hipEventRecord(s.ev_src, q));
        HIPM_TRY(hipStreamWaitEvent(s.copy, s.ev_src, 0));
        const size_t off = pitch * s.rb, bytes = pitch * (s.re - s.rb);
        if (s.device == device) HIPM_TRY(hipMemcpyAsync((uint8_t*)dst + off, (const uint8_t*)src + off, bytes, hipMemcpyDeviceToDevice, s.copy));
        else HIPM_TRY(hipMemcpyPeerAsync((uint8_t*)dst + off, device, (const uint8_t*)src + off, s.device, bytes, s.copy));
        HIPM_TRY(hipEventRecord(s.ev_gather, s.copy));
        s.gather_pending = true;
    }
    DevGuard g(device);
    for (Strip& s : m->strips) {
        if (stream) HIPM_TRY(hipStreamWaitEvent((hipStream_t)stream, s.ev_gather, 0));
        else HIPM_TRY(hipEventSynchronize(s.ev_gather));
    }
    return FRT_OK;
}

// Full-frame read-back: the device-side gather above into a buffer on the first strip's device, then ONE device-to-host copy (the reference reads
// one texture per presented frame, state.rs:226-278; strip by strip through the host this was N synchronisations + N copies).
int frt_multi_renderer_read_buffer(frt_multi_renderer* m, int buf, int index, void* out) {
    const uint32_t bpp = bpp_of_buf(buf);
    if (!m || !out || !bpp) return set_error(FRT_ERR_INVALID_ARG, "multi read_buffer: bad arguments");
    const size_t bytes = (size_t)m->W * m->H * bpp;
    const int dev = m->strips[0].device;
    DevGuard g(dev);
    if (m->gather_bytes < bytes) {
        if (m->gather_buf) (void)hipFree(m->gather_buf);
        m->gather_buf = nullptr; m->gather_bytes = 0;
        HIPM_TRY(hipMalloc(&m->gather_buf, bytes));
        m->gather_bytes = bytes;
    }
    int rc = frt_multi_renderer_gather(m, buf, index, dev, m->gather_buf, nullptr);
    if (rc) return rc;
    HIPM_TRY(hipMemcpy(out, m->gather_buf, bytes, hipMemcpyDeviceToHost));
    return FRT_OK;
}
int frt_multi_renderer_read_display(frt_multi_renderer* m, uint8_t* rgba8) { return frt_multi_renderer_read_buffer(m, FRT_BUF_DISPLAY, 0, rgba8); }
int frt_multi_renderer_read_accum(frt_multi_renderer* m, float* rgba32f) {
    if (!m) return set_error(FRT_ERR_INVALID_ARG, "multi read_accum: null");
    return frt_multi_renderer_read_buffer(m, FRT_BUF_ACCUM, m->frame ? (int)((m->frame - 1u) & 1u) : 0, rgba32f);
}

// Sums over the strips (rays are counted only for rows a strip owns, so the totals are those of a single renderer).
int frt_multi_renderer_stats(frt_multi_renderer* m, frt_stats* out) {
    if (!m || !out) return set_error(FRT_ERR_INVALID_ARG, "multi stats: null");
    frt_stats t{};
    for (Strip& s : m->strips) {
        frt_stats st;
        int rc = frt_renderer_stats(s.r, &st);
        if (rc) return rc;
        t.rays_closest += st.rays_closest; t.rays_any += st.rays_any;
        t.frames = std::max(t.frames, st.frames);
        for (int i = 0; i < 4; ++i) { t.ms_stage[i] = std::max(t.ms_stage[i], st.ms_stage[i]); t.launches[i] += st.launches[i]; t.rays_stage[i][0] += st.rays_stage[i][0]; t.rays_stage[i][1] += st.rays_stage[i][1]; }
        t.halo_overflow += st.halo_overflow; t.ms_merge = std::max(t.ms_merge, st.ms_merge); t.queue_overflow += st.queue_overflow;
        t.queue_capacity = std::max(t.queue_capacity, st.queue_capacity); t.speculated_frames += st.speculated_frames;
        t.discarded_speculations += st.discarded_speculations; t.queue_bytes += st.queue_bytes;
    }
    *out = t;
    return FRT_OK;
}

}   // extern "C"
