"""Image-strip partition of the frame across ranks and the per-frame halo exchange.

The reference is single-GPU; this is the multi-GPU form of its render loop. The framebuffer is cut into `world`
horizontal strips (contiguous in row-major, so strips gather in place); the scene is replicated. Stage dependencies:
  G-buffer  : pure function of the pixel            -> each rank also computes 12 halo rows itself, no exchange
  temporal  : reads only its own pixel (static camera, restir.wgsl:846-855)
  spatial   : reads temporal reservoirs within 10 px (restir_spatial.wgsl:902-921); this build runs it on 2 extra rows
              per side so that post's +-2-row radiance reads (post.wgsl:93) stay local  -> needs 12 reservoir halo rows
  post      : reads the previous accumulation within +-1 row (post.wgsl:196-199)            -> needs 1 history halo row
So ONE exchange per frame, between the temporal and spatial stages: 12 rows of reservoirs (32 B/px) and 1 row of the
previous frame's accumulation (16 B/px) with each vertical neighbour. Everything else is local.

`exchange_halos` is transport-agnostic: it moves `rows(...)` tensors with torch.distributed point-to-point ops
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" on CPU for tests).
"""
import numpy as np

HALO_RESERVOIR = 12   # spatial radius 10 + 2 rows of redundant spatial work (frt_renderer.hip kHaloGbuffer)
HALO_HISTORY = 1
BUF_RESERVOIR, BUF_ACCUM = 4, 7


class StripPlan:
    """Rows [row_begin, row_end) of an image of height H owned by `rank` of `world`."""

    def __init__(self, height, world, rank):
        assert 0 <= rank < world
        self.H, self.world, self.rank = height, world, rank
        self.row_begin = height * rank // world
        self.row_end = height * (rank + 1) // world
        if world > 1 and self.row_end - self.row_begin < HALO_RESERVOIR:
            raise ValueError(f"strip of {self.row_end - self.row_begin} rows is thinner than the {HALO_RESERVOIR}-row halo")

    def transfers(self, frame):
        """[(peer, buf, index, send_rows, recv_rows)] for the exchange of frame `frame` (before its spatial stage)."""
        out = []
        rb, re = self.row_begin, self.row_end
        hist = (frame - 1) % 2          # post.rs:209-224: history = the slot written by the previous frame
        if self.rank > 0:               # upper neighbour owns rows < rb
            out.append((self.rank - 1, BUF_RESERVOIR, 0, (rb, rb + HALO_RESERVOIR), (rb - HALO_RESERVOIR, rb)))
            if frame > 0:
                out.append((self.rank - 1, BUF_ACCUM, hist, (rb, rb + HALO_HISTORY), (rb - HALO_HISTORY, rb)))
        if self.rank < self.world - 1:  # lower neighbour owns rows >= re
            out.append((self.rank + 1, BUF_RESERVOIR, 0, (re - HALO_RESERVOIR, re), (re, re + HALO_RESERVOIR)))
            if frame > 0:
                out.append((self.rank + 1, BUF_ACCUM, hist, (re - HALO_HISTORY, re), (re, re + HALO_HISTORY)))
        return out


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
        return self.arena[o:o + (y1 - y0) * pitch]

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


def exchange_halos(access, plan, frame, group=None):
    """One batched point-to-point exchange with both vertical neighbours (torch.distributed; nccl = RCCL, or gloo)."""
    import torch.distributed as dist
    tr = plan.transfers(frame)
    if not tr:
        return
    ops, recvs = [], []
    for peer, buf, index, srows, rrows in tr:
        ops.append(dist.P2POp(dist.isend, access.rows(buf, index, *srows), peer, group))
        rb = access.recv_buffer(buf, index, *rrows)
        recvs.append((buf, index, rrows, rb))
        ops.append(dist.P2POp(dist.irecv, rb, peer, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for buf, index, rrows, rb in recvs:
        access.store(buf, index, *rrows, rb)


def exchange_halos_host(renderers, plans, frame):
    """Same exchange between strip renderers living in ONE process (tests on a single GPU): rows go through host memory."""
    by_rank = {p.rank: r for r, p in zip(renderers, plans)}
    for r, p in zip(renderers, plans):
        for peer, buf, index, srows, _ in p.transfers(frame):
            data = r.read_rows(buf, index, *srows)
            by_rank[peer].write_rows(buf, index, *srows, data)     # the sender's rows land at the same image rows of the peer


def gather_strips(local_rows_tensor, plan, group=None):
    """All-gather of equally sized strip tensors into the full frame (in-place layout: strip k at rows [k*h, (k+1)*h))."""
    import torch
    import torch.distributed as dist
    parts = [torch.empty_like(local_rows_tensor) for _ in range(plan.world)]
    dist.all_gather(parts, local_rows_tensor, group=group)
    return torch.cat(parts, dim=0)
