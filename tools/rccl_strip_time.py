"""What the RCCL halo exchanges cost a thin strip, measured on ONE GPU: the N > 1 frame loop of bench.py (frt.dist.render_strip_frame) for strip 4 of 8 of the
1080p frame over the real "nccl" backend with a world of one rank — every transfer a send-to-self of the rows a neighbour would send (same bytes, same
stream orderings; tests/_nccl_selftest.py checks the pixels of this set-up) — against the same loop without transfers. The HIP runtime maps a process's
streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): a strip renderer under the pipeline uses four streams itself, RCCL adds its own.
    RCCL_STRIP_DIRECT=1 python tools/rccl_strip_time.py [rank world] [4k]      # what bench.py's ranks run: RCCL called directly in the renderer's edge stream (frt.rccl)
    python tools/rccl_strip_time.py [rank world] [4k]                          # the fallback: torch.distributed's batched point-to-point operations
Knobs that reproduce the states profiles/r4_experiments/rccl_strips.md walks through:
    RCCL_STRIP_TORCH_STREAM=1 [RCCL_STRIP_OWN_STREAM=1]   the renderer is handed torch's current stream as its main stream (bench.py before round 4's fix)
    RCCL_STRIP_COLD_P2P=1                                 torch's RCCL point-to-point stream is created by the first frame's transfer, not before the renderer
    RCCL_STRIP_FAKE=record | wait | copy | nofinish       the exchange replaced by parts of its stream choreography (diagnostics)
    RCCL_STRIP_DUMMY_STREAMS=k                            k streams created first (shifts the stream -> hardware queue mapping)
    RCCL_STRIP_TRACE=1                                    a short run for rocprofv3 --kernel-trace (tools/timeline_all.py prints the timeline)
    GPU_MAX_HW_QUEUES=n, NCCL_NCHANNELS_PER_PEER=n        the runtime's / RCCL's own knobs"""
import os, sys, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1")
import torch
import torch.distributed as dist
import frt
from frt.dist import StripPlan, ArenaRows, render_strip_frame, HALO_RESERVOIR, BUF_RESERVOIR, BUF_ACCUM
args = [x for x in sys.argv[1:] if x != "4k"]
rank = int(args[0]) if args else 4
world = int(args[1]) if len(args) > 1 else 8
W, H = (3840, 2160) if "4k" in sys.argv[1:] else (1920, 1080)
rb, re = H * rank // world, H * (rank + 1) // world
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


class Loopback(StripPlan):
    def __init__(self, quiet):
        self.H, self.world, self.rank, self.motion_halo = H, 1, 0, 0
        self.boundaries = [0, H]; self.row_begin, self.row_end = rb, re
        self.quiet = quiet

    def transfers(self, frame, when="mid"):
        if self.quiet:
            return []
        if when == "mid":      # both neighbours: 12 rows each way
            return [(0, BUF_RESERVOIR, 0, (rb, rb + HALO_RESERVOIR), (re, re + HALO_RESERVOIR)), (0, BUF_RESERVOIR, 0, (re - HALO_RESERVOIR, re), (rb - HALO_RESERVOIR, rb))]
        if when == "post" and frame > 0:
            return [(0, BUF_ACCUM, (frame - 1) % 2, (rb, rb + 1), (re, re + 1)), (0, BUF_ACCUM, (frame - 1) % 2, (re - 1, re), (rb - 1, rb))]
        return []


if os.environ.get("RCCL_STRIP_OWN_STREAM"):      # main stream = a stream of torch's pool instead of the legacy default stream
    torch.cuda.set_stream(torch.cuda.Stream())
if not os.environ.get("RCCL_STRIP_COLD_P2P"):      # create RCCL's point-to-point communicator and stream BEFORE the renderer's streams
    a_, b_ = torch.zeros(256, device="cuda:0"), torch.zeros(256, device="cuda:0")
    for w_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, a_, 0), dist.P2POp(dist.irecv, b_, 0)]): w_.wait()
    torch.cuda.synchronize()
def frame_nofinish(r, access, plan, cam, frame):
    """Diagnostic: the real RCCL batch is posted, but nothing waits for it (wrong pixels): what does POSTING it cost the main stream?"""
    from frt.dist import start_exchange
    r.render_phases(cam, frt.PHASE_GBUFFER | frt.PHASE_TEMPORAL)
    ex = start_exchange(access, plan, frame, None, when=("mid", "post"))
    r.render_phases(cam, frt.PHASE_SPATIAL_INNER)
    r.render_phases(cam, frt.PHASE_SPATIAL_EDGE)
    r.render_phases(cam, frt.PHASE_POST)
    r.end_frame()
    _keep.append(ex)
    if len(_keep) > 8: _keep.pop(0)
_keep = []


_side = {}
def frame_fake(r, access, plan, cam, frame):
    """Diagnostic: the frame loop with the exchange replaced by the stream choreography alone (RCCL_STRIP_FAKE = record: an event recorded on the main stream
    behind T-merge; wait: ... and a side stream made to wait for it; copy: ... and a 737 KB device copy on the side stream that the edge stream then waits for)."""
    mode = os.environ["RCCL_STRIP_FAKE"]
    r.render_phases(cam, frt.PHASE_GBUFFER | frt.PHASE_TEMPORAL)
    if not plan.quiet:
        if "s" not in _side:
            _side["s"] = torch.cuda.Stream(); _side["a"] = torch.zeros(737280, dtype=torch.uint8, device="cuda:0"); _side["b"] = torch.zeros_like(_side["a"])
        ev = torch.cuda.Event()
        ev.record()
        if mode in ("wait", "copy"):
            _side["s"].wait_event(ev)
        if mode == "copy":
            with torch.cuda.stream(_side["s"]):
                _side["b"].copy_(_side["a"], non_blocking=True)
                ev2 = torch.cuda.Event(); ev2.record()
            access.edge_stream().wait_event(ev2)
    r.render_phases(cam, frt.PHASE_SPATIAL_INNER)
    r.render_phases(cam, frt.PHASE_SPATIAL_EDGE)
    r.render_phases(cam, frt.PHASE_POST)
    r.end_frame()


scene = frt.scenes.create_cornell_box()
nbytes = frt.Renderer.arena_bytes(W, H)
arena = torch.zeros(nbytes + 256, dtype=torch.uint8, device="cuda:0")
off = (-arena.data_ptr()) % 256
_dummies = [frt.Renderer(scene, 16, 16, device=0) for _ in range(int(os.environ.get("RCCL_STRIP_DUMMY_STREAMS", "0")))]      # diagnostic: shift the stream -> hardware queue mapping (one stream each)
if not os.environ.get("RCCL_STRIP_TORCH_STREAM"):      # (bench.py since round 4) the renderer creates its main stream itself, next to its other streams; torch is told to use it
    torch.cuda.synchronize()
    r = frt.Renderer(scene, W, H, device=0, rows=(rb, re), arena=arena.data_ptr() + off, arena_bytes=nbytes, flags=frt.FLAG_PIPELINE)
    torch.cuda.set_stream(torch.cuda.ExternalStream(r.stream_handle(0)))
else:
    r = frt.Renderer(scene, W, H, device=0, stream=torch.cuda.current_stream().cuda_stream, rows=(rb, re), arena=arena.data_ptr() + off, arena_bytes=nbytes, flags=frt.FLAG_PIPELINE)
rows = ArenaRows(r, arena)
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(80)]
res, host = {}, {}
DUM = os.environ.get("RCCL_STRIP_DUMMY_STREAMS")
WHAT = 'the RCCL exchanges (send-to-self), RCCL called directly on the edge stream' if os.environ.get('RCCL_STRIP_DIRECT') else 'the RCCL exchanges (send-to-self)' if not os.environ.get('RCCL_STRIP_FAKE') else 'the fake exchange (' + os.environ['RCCL_STRIP_FAKE'] + ')'
_comm = None
if os.environ.get('RCCL_STRIP_DIRECT'):      # RCCL called directly on the renderer's streams (frt.rccl), as bench.py's ranks do
    import frt.rccl
    _comm = frt.rccl.Comm.create(0, 1, 0)
frame = frame_nofinish if os.environ.get('RCCL_STRIP_FAKE') == 'nofinish' else frame_fake if os.environ.get('RCCL_STRIP_FAKE') else (lambda r_, a_, p_, c_, f_: render_strip_frame(r_, a_, p_, c_, f_, frt, comm=_comm))
for rnd in range(1 if os.environ.get('RCCL_STRIP_TRACE') else 3):
    for quiet in ((False,) if os.environ.get('RCCL_STRIP_TRACE') else (True, False)):
        plan = Loopback(quiet)
        r.clear()
        for f in range(8): frame(r, rows, plan, cams[f], f)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in range(8, 72): frame(r, rows, plan, cams[f], f)
        th = (time.perf_counter() - t0) / 64 * 1e3       # host time to ENQUEUE a frame (the GPU runs behind)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 64 * 1e3
        res[quiet] = min(res.get(quiet, 1e9), t)
        host[quiet] = min(host.get(quiet, 1e9), th)
if True not in res: res[True] = float("nan")
print((f"{DUM} dummy streams first, " if DUM else "") + ("" if not os.environ.get("RCCL_STRIP_COLD_P2P") else "RCCL's P2P stream created by the first frame's transfer, ") + ("main stream = torch's current stream (" + ("a pool stream" if os.environ.get("RCCL_STRIP_OWN_STREAM") else "the legacy default stream") + "), " if os.environ.get("RCCL_STRIP_TORCH_STREAM") else "the renderer's own main stream, ") + f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: strip {rank} of {world} of {W}x{H}: {res[True]:.3f} ms per frame without transfers, {res[False]:.3f} with {WHAT}: exposed {res[False] - res[True]:+.3f} ms; host enqueue time per frame {host.get(True, float('nan')):.3f} / {host[False]:.3f} ms", flush=True)
dist.destroy_process_group()
