"""Image-strip partition of the frame across ranks and the per-frame halo exchanges.

The reference is single-GPU; this is the multi-GPU form of its render loop. The framebuffer is cut into `world`
horizontal strips (contiguous in row-major, so strips gather in place); the scene is replicated. Stage dependencies:
  G-buffer  : pure function of the pixel            -> each rank also computes 12 halo rows itself, no exchange
  T-trace   : reads only its own pixel's G-buffer   -> no exchange (and it runs ahead of its frame, frt_renderer.hip)
  T-merge   : reads its own pixel's previous spatial reservoir (static camera, restir.wgsl:846-855)
  spatial   : reads temporal reservoirs within 10 px (restir_spatial.wgsl:902-921); this build runs it on 2 extra rows
              per side so that post's +-2-row radiance reads (post.wgsl:93) stay local  -> needs 12 reservoir halo rows
  post      : reads the previous accumulation within +-1 row (post.wgsl:196-199)            -> needs 1 history halo row
Two sets of rows per frame and vertical neighbour, on different dependency chains, travelling in ONE batch (render_strip_frame):
  "mid"  : 12 rows of temporal reservoirs (32 B/px) between T-merge and the spatial stage. This one sits on the frame-to-frame
           chain (T-merge -> spatial -> T-merge); it is overlapped with the spatial stage's INTERIOR rows, which need nothing from a
           neighbour (FRT_PHASE_SPATIAL_INNER), and only the edge rows wait for it.
  "post" : 1 row of the previous frame's accumulation (16 B/px) for post. Its source is the previous frame's post and its consumer this
           frame's post, so it may travel any time in between: it rides with the "mid" rows (a batch of its own at the start of the
           frame cost a thin strip a second RCCL launch on its chain, profiles/r4_experiments/rccl_strips.md).

Moving camera (StripPlan(motion_halo=K), Renderer(motion_halo=K)): T-merge reprojects into the PREVIOUS frame's spatial
reservoirs and G-buffer (restir.wgsl:846-900) and post fetches the previous accumulation bilinearly at the reprojected position
(post.wgsl:187-266), both up to K rows outside the strip as long as the camera moves less than that per frame. The previous G-buffer
is local (the G-buffer halo grows to K rows); a "pre" exchange before T-merge brings K rows of the previous spatial reservoirs, and
the "post" exchange grows to K + 1 rows. Reads beyond the halo are counted (stats()["halo_overflow"]); check_halo() raises on them.

`exchange_halos` is transport-agnostic: it moves `rows(...)` tensors with torch.distributed point-to-point ops
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" on CPU for tests). `render_strip_frame` is one frame of one rank:
the phases and the exchanges in the order described above (what bench.py --gpus N and the multi-rank tests run). Given a
`frt.rccl.Comm` it places the transfers as grouped RCCL launches directly in the renderer's streams instead (`render_strip_frame_direct`:
no torch stream, no event per operation — the form bench.py's ranks use on a GPU node).
"""
import numpy as np

HALO_RESERVOIR = 12   # spatial radius 10 + 2 rows of redundant spatial work (frt_renderer.hip kHaloGbuffer)
HALO_HISTORY = 1
BUF_RESERVOIR, BUF_ACCUM = 4, 7


class StripPlan:
    """Rows [row_begin, row_end) of an image of height H owned by `rank` of `world`.

    boundaries: optional list of world + 1 ascending row indices (0 ... H), e.g. from balanced_boundaries(); default = equal strips."""

    def __init__(self, height, world, rank, boundaries=None, motion_halo=0):
        assert 0 <= rank < world
        self.H, self.world, self.rank, self.motion_halo = height, world, rank, int(motion_halo)
        if boundaries is None:
            boundaries = [height * k // world for k in range(world + 1)]
        assert len(boundaries) == world + 1 and boundaries[0] == 0 and boundaries[-1] == height
        self.boundaries = list(boundaries)
        self.row_begin, self.row_end = boundaries[rank], boundaries[rank + 1]
        need = max(HALO_RESERVOIR, self.motion_halo + HALO_HISTORY)
        if world > 1 and min(b - a for a, b in zip(boundaries, boundaries[1:])) < need:
            raise ValueError(f"a strip is thinner than the {need}-row halo: {boundaries}")

    def _pairs(self, buf, index, rows):
        """Both neighbours: send my `rows` boundary rows, receive theirs into the rows just outside my strip."""
        out = []
        rb, re = self.row_begin, self.row_end
        if self.rank > 0:               # upper neighbour owns rows < rb
            out.append((self.rank - 1, buf, index, (rb, rb + rows), (rb - rows, rb)))
        if self.rank < self.world - 1:  # lower neighbour owns rows >= re
            out.append((self.rank + 1, buf, index, (re - rows, re), (re, re + rows)))
        return out

    def transfers(self, frame, when="mid", serial=None):
        """[(peer, buf, index, send_rows, recv_rows)] for the frame whose frame_count is `frame`.
        when="pre": before its T-merge (moving camera only); "mid": between T-merge and the spatial stage; "post": before its post stage.
        serial: frames rendered since the renderers were created, for hosts that RESET frame_count while the camera moves (state.rs:152): the
        previous frame's spatial reservoirs exist — and T-merge reprojects into them — whatever frame_count says, so "pre" is gated on
        `serial`, not on `frame` (default: serial = frame, a counter that never restarts). Post ignores its history at frame_count 0
        (post.wgsl:187): "post" is gated on `frame`."""
        hist = (frame - 1) % 2          # post.rs:209-224: history = the slot written by the previous frame
        if serial is None:
            serial = frame
        if when == "pre":
            if self.motion_halo == 0 or serial == 0:
                return []
            return self._pairs(BUF_RESERVOIR, 1, self.motion_halo)             # previous spatial reservoirs (reservoir_buffers[1])
        if when == "post":
            if frame == 0:
                return []
            return self._pairs(BUF_ACCUM, hist, (self.motion_halo + HALO_HISTORY) if self.motion_halo else HALO_HISTORY)
        assert when == "mid"
        return self._pairs(BUF_RESERVOIR, 0, HALO_RESERVOIR)


class ArenaRows:
    """Row access for a frt.Renderer whose per-pixel buffers live in a caller-owned torch uint8 arena: zero-copy views."""

    def __init__(self, renderer, arena, staging_device=None):
        self.r, self.arena = renderer, arena
        self.base = arena.data_ptr()
        self.staging = staging_device          # "cpu" when the process group cannot move device tensors (gloo)

    def _view(self, buf, index, y0, y1):
        p, bpp = self.r.buffer_info(buf, index)
        pitch = self.r.width * bpp
        o = p - self.base + y0 * pitch
        n = (y1 - y0) * pitch
        if not (0 <= o and o + n <= self.arena.numel()):      # e.g. a strip renderer's third G-buffer set lives outside the caller's arena
            raise ValueError(f"buffer {buf}[{index}] rows {y0}:{y1} do not lie inside the caller's arena (offset {o}, {n} bytes)")
        return self.arena[o:o + n]

    def rows(self, buf, index, y0, y1):
        v = self._view(buf, index, y0, y1)
        return v.to(self.staging) if self.staging else v

    def recv_buffer(self, buf, index, y0, y1):
        v = self._view(buf, index, y0, y1)
        import torch
        return torch.empty(v.shape, dtype=v.dtype, device=self.staging) if self.staging else v

    def store(self, buf, index, y0, y1, t):
        v = self._view(buf, index, y0, y1)
        if t.data_ptr() != v.data_ptr():
            v.copy_(t, non_blocking=True)


    def edge_stream(self):
        """The stream on which the renderer launches the edge rows of the spatial stage (frt_renderer_stream(r, 2)) as a torch stream."""
        import torch
        h = self.r.stream_handle(2)
        if not h or h == self.r.stream_handle(0):
            return torch.cuda.current_stream()
        if getattr(self, "_edge", None) is None or self._edge_handle != h:
            self._edge, self._edge_handle = torch.cuda.ExternalStream(h), h
        return self._edge


class _Exchange:
    """A batched point-to-point exchange in flight: start() posts the sends / receives, finish() waits and lands the received rows."""

    def __init__(self, access, works, recvs):
        self.access, self.works, self.recvs = access, works, recvs

    def finish(self):
        for w in self.works:
            w.wait()       # nccl: orders the CURRENT torch stream behind the transfer (no host wait); gloo: blocks the host
        for buf, index, rrows, rb in self.recvs:
            self.access.store(buf, index, *rrows, rb)
        self.works, self.recvs = [], []


def start_exchange(access, plan, frame, group=None, when="mid", serial=None):
    """Post one batched exchange with both vertical neighbours (torch.distributed; nccl = RCCL, or gloo). Returns an _Exchange or None.
    With the nccl backend the transfer is ordered behind the torch stream that is current HERE and finish() orders the stream that is
    current THERE behind it: callers pick the streams (render_strip_frame)."""
    import torch.distributed as dist
    tr = []
    for w in ((when,) if isinstance(when, str) else when):      # several exchanges in ONE batch (one RCCL launch instead of one each)
        tr += plan.transfers(frame, w, serial) if serial is not None else plan.transfers(frame, w)
    if not tr:
        return None
    ops, recvs = [], []
    for peer, buf, index, srows, rrows in tr:
        ops.append(dist.P2POp(dist.isend, access.rows(buf, index, *srows), peer, group))
        rb = access.recv_buffer(buf, index, *rrows)
        recvs.append((buf, index, rrows, rb))
        ops.append(dist.P2POp(dist.irecv, rb, peer, group))
    return _Exchange(access, dist.batch_isend_irecv(ops), recvs)


def exchange_direct(comm, access, plan, frame, when, stream_handle, serial=None):
    """The same exchange as ONE grouped RCCL launch (frt.rccl.Comm) IN the HIP stream `stream_handle`: the rows are read and written in place (zero-copy
    arena views), the transfer is ordered by that stream alone — whatever the stream holds before is done before the rows leave, whatever it gets next sees
    the rows that arrived. Returns True when something was posted."""
    tr = []
    for w in ((when,) if isinstance(when, str) else when):
        tr += plan.transfers(frame, w, serial) if serial is not None else plan.transfers(frame, w)
    if not tr:
        return False
    sends, recvs = [], []
    for peer, buf, index, srows, rrows in tr:
        sv, rv = access._view(buf, index, *srows), access._view(buf, index, *rrows)
        sends.append((sv.data_ptr(), sv.numel(), peer))
        recvs.append((rv.data_ptr(), rv.numel(), peer))
    comm.exchange(sends, recvs, stream_handle)
    return True


def exchange_halos(access, plan, frame, group=None, when="mid"):
    """start_exchange + finish: the blocking form (CPU oracle strips; host-staged transports)."""
    ex = start_exchange(access, plan, frame, group, when)
    if ex:
        ex.finish()


def render_strip_frame(r, access, plan, cam, frame, frt, group=None, serial=None, comm=None):
    """One frame of one rank's strip renderer `r` (frt.Renderer on torch's current stream, buffers in `access`'s arena).

    comm: a frt.rccl.Comm -> the exchanges are grouped RCCL launches placed directly IN the renderer's streams (render_strip_frame_direct below);
    None -> torch.distributed's batched point-to-point operations, as described here.

    Everything a transfer touches (reservoirs, accumulation) is produced on the renderer's main stream = torch's current stream, so the
    transfers are ordered by that stream alone. ONE batch per frame (static camera): the "mid" rows (this frame's temporal reservoirs) and
    the "post" rows (the previous frame's accumulation, needed only by this frame's post stage) are posted together behind T-merge and
    overlap the interior rows of the spatial stage; only its edge rows wait. (Two batches — "post" at the start of the frame, "mid" behind
    T-merge — put two RCCL launches with ~50 us of launch latency each on a thin strip's chain: 0.503 -> 0.43 ms per frame for a 1/8 strip
    of the 1080p frame, tools/rccl_strip_time.py.)
    With a host-staged transport (gloo rehearsal, `access.staging`) start_exchange blocks in the device-to-host copy; same order."""
    if comm is not None:
        return render_strip_frame_direct(r, access, plan, cam, frame, frt, comm, serial)
    pre = start_exchange(access, plan, frame, group, when="pre", serial=serial)        # behind spatial(f-1) (moving camera only)
    if pre:
        pre.finish()
    r.render_phases(cam, frt.PHASE_GBUFFER | frt.PHASE_TEMPORAL)        # T-merge(f) (G-buffer + T-trace ran ahead of the frame)
    mid = start_exchange(access, plan, frame, group, when=("mid", "post"))      # behind T-merge(f) — and so behind post(f-1)
    r.render_phases(cam, frt.PHASE_SPATIAL_INNER)                       # interior rows: need nothing from a neighbour
    if mid:
        import torch
        with torch.cuda.stream(access.edge_stream()):
            mid.finish()                                                # only the stream of the edge rows waits for the neighbours' rows
    r.render_phases(cam, frt.PHASE_SPATIAL_EDGE)                        # edge rows (beside the interior ones) + continuations: the main stream joins the edge stream
    r.render_phases(cam, frt.PHASE_POST)                                # ... and is thereby behind the arrival of the accumulation rows
    r.end_frame()


def render_strip_frame_direct(r, access, plan, cam, frame, frt, comm, serial=None):
    """render_strip_frame with RCCL called directly on the renderer's streams (frt.rccl): no torch stream, no events per transfer.
      "pre" rows (moving camera): a grouped launch in the MAIN stream, right before T-merge, which consumes them;
      "mid" + "post" rows: ONE grouped launch in the EDGE stream (frt_renderer_stream(r, 2)) — ordered behind T-merge by one event — followed in that
      stream by the edge rows' launches; the main stream joins the edge stream before the continuation launches and post (frt_renderer.hip), and is
      thereby behind the arrival of the accumulation rows and behind the departure of everything T-merge(f+1) and post(f+1) overwrite.
    Measured on one GPU (send-to-self, tools/rccl_strip_time.py): see profiles/r4_experiments/rccl_strips.md."""
    import torch
    exchange_direct(comm, access, plan, frame, "pre", r.stream_handle(0), serial)
    r.render_phases(cam, frt.PHASE_GBUFFER | frt.PHASE_TEMPORAL)        # T-merge(f)
    h_edge = r.stream_handle(2)
    if h_edge != r.stream_handle(0):
        if hasattr(r, "order_edge_stream"):
            r.order_edge_stream()                                       # the edge stream behind T-merge, with the event the renderer recorded there anyway
        else:
            access.edge_stream().wait_stream(torch.cuda.current_stream())   # (a wrapper without the call: one more event on the main stream)
    exchange_direct(comm, access, plan, frame, ("mid", "post"), h_edge)
    r.render_phases(cam, frt.PHASE_SPATIAL_INNER)                       # interior rows beside the transfer
    r.render_phases(cam, frt.PHASE_SPATIAL_EDGE)                        # edge rows: in the edge stream behind the transfer
    r.render_phases(cam, frt.PHASE_POST)
    r.end_frame()


def check_halo(renderer):
    """Raise if the strip read previous-frame state beyond its motion halo (the frame then differs from a 1-GPU frame)."""
    n = renderer.stats()["halo_overflow"]
    if n:
        raise RuntimeError(f"{n} previous-frame reads fell outside the strip's motion halo: raise motion_halo (camera moves too fast for it)")


def exchange_halos_host(renderers, plans, frame, when="mid", serial=None):
    """Same exchange between strip renderers living in ONE process (tests on a single GPU): rows go through host memory
    (read_rows / write_rows wait for everything the renderers have enqueued)."""
    by_rank = {p.rank: r for r, p in zip(renderers, plans)}
    for r, p in zip(renderers, plans):
        for peer, buf, index, srows, _ in (p.transfers(frame, when, serial) if serial is not None else p.transfers(frame, when)):
            data = r.read_rows(buf, index, *srows)
            by_rank[peer].write_rows(buf, index, *srows, data)     # the sender's rows land at the same image rows of the peer


def gather_strips(local_rows_tensor, plan, group=None):
    """All-gather of the strips ([rows, ...] tensors, possibly of different heights) into the full frame."""
    import torch
    import torch.distributed as dist
    heights = [b - a for a, b in zip(plan.boundaries, plan.boundaries[1:])]
    hmax = max(heights)
    pad = torch.zeros((hmax,) + tuple(local_rows_tensor.shape[1:]), dtype=local_rows_tensor.dtype, device=local_rows_tensor.device)
    pad[:local_rows_tensor.shape[0]] = local_rows_tensor
    parts = [torch.empty_like(pad) for _ in range(plan.world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:h] for p, h in zip(parts, heights)], dim=0)


def balanced_boundaries(frt, scene, width, height, world, max_depth=8, bands=36, probe_frames=2, device=0):
    """Strip boundaries that equalise work instead of rows (the ceiling strip of a Cornell Box is far cheaper than the floor strip).

    Every rank runs the same probe: a quarter-resolution render, band by band, reading the exact device ray counters of each band
    (integers, so all ranks compute identical boundaries without talking to each other). Cost model: rays + 4 per pixel.
    The image does not depend on the partition (tests: strips == whole image, bit for bit), only the time does."""
    if world == 1:
        return [0, height]
    pw, ph = max(width // 4, 16), max(height // 4, bands)
    cost = []
    for b in range(bands):
        y0, y1 = ph * b // bands, ph * (b + 1) // bands
        r = frt.Renderer(scene, pw, ph, max_depth=max_depth, device=device, rows=(y0, y1))
        cam = frt.CameraController()
        for _ in range(probe_frames):
            r.render(cam.build_uniform(width / height, r.frame_count, scene.num_lights))
        st = r.stats()
        cost.append(st["rays_closest"] + st["rays_any"] + 4 * pw * (y1 - y0) * probe_frames)
        del r
    # cumulative cost over full-resolution rows (piecewise linear inside a band)
    row_cost = np.zeros(height)
    for b in range(bands):
        y0, y1 = height * b // bands, height * (b + 1) // bands
        row_cost[y0:y1] = cost[b] / max(y1 - y0, 1)
    cum = np.concatenate([[0.0], np.cumsum(row_cost)])
    bounds = [0]
    for k in range(1, world):
        y = int(np.searchsorted(cum, cum[-1] * k / world))
        y = max(y, bounds[-1] + HALO_RESERVOIR)                       # every strip at least as tall as the halo
        y = min(y, height - HALO_RESERVOIR * (world - k))
        bounds.append(y)
    bounds.append(height)
    return bounds
