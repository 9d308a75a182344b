"""RCCL rehearsal on ONE GPU: the N > 1 frame loop of bench.py (frt.dist.render_strip_frame: arena views as send / receive buffers, the "mid"
rows and the "post" rows in one batch posted behind T-merge and finished on the edge-row stream) over the real "nccl" backend with a
world of one rank, every transfer a send-to-self inside one batch. The image means nothing (the strip's "neighbour" is itself); what is
checked is that the communicator comes up on cuda:0, that device-to-device point-to-point transfers of arena rows complete under the
stream ordering the loop sets up, and that the rows arrive bit for bit. Prints one JSON line."""
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))


def main():
    port = sys.argv[1] if len(sys.argv) > 1 else "29533"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    import numpy as np
    import torch
    import torch.distributed as dist
    import frt
    from frt.dist import StripPlan, ArenaRows, render_strip_frame, HALO_RESERVOIR, BUF_RESERVOIR, BUF_ACCUM
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

    class Loopback(StripPlan):
        """A strip in the middle of the image whose upper AND lower neighbour is this rank: its top boundary rows go to the halo below it."""
        def __init__(self, H, rb, re):
            self.H, self.world, self.rank, self.motion_halo = H, 1, 0, 0
            self.boundaries = [0, H]; self.row_begin, self.row_end = rb, re

        def transfers(self, frame, when="mid"):
            rb, re = self.row_begin, self.row_end
            if when == "mid":
                return [(0, BUF_RESERVOIR, 0, (rb, rb + HALO_RESERVOIR), (re, re + HALO_RESERVOIR))]
            if when == "post" and frame > 0:
                return [(0, BUF_ACCUM, (frame - 1) % 2, (rb, rb + 1), (re, re + 1))]
            if when == "pre" and frame > 0:      # (a moving camera's exchange: the previous frame's spatial reservoirs, consumed by T-merge on the MAIN stream)
                return [(0, BUF_RESERVOIR, 1, (rb, rb + 4), (re, re + 4))]
            return []

    scene = frt.scenes.create_cornell_box()

    class Stepwise:
        """The renderer with a host wait around every phase: the reference the asynchronous loop must match bit for bit."""
        def __init__(self, r): self.r = r
        def render_phases(self, cam, phases): torch.cuda.synchronize(); self.r.render_phases(cam, phases); torch.cuda.synchronize()
        def end_frame(self): self.r.end_frame()
        def stream_handle(self, which): return self.r.stream_handle(which)

    import frt.rccl
    direct = frt.rccl.Comm.create(0, 1, 0)      # RCCL called directly on the renderer's streams (frt.dist.render_strip_frame_direct): what bench.py's ranks use

    def run(W, H, rb, re, N, stepwise, comm=None):
        nbytes = frt.Renderer.arena_bytes(W, H)
        arena = torch.zeros(nbytes + 256, dtype=torch.uint8, device="cuda:0")
        off = (-arena.data_ptr()) % 256
        r = frt.Renderer(scene, W, H, device=0, stream=torch.cuda.current_stream().cuda_stream, rows=(rb, re),
                         arena=arena.data_ptr() + off, arena_bytes=nbytes, flags=frt.FLAG_PIPELINE)
        rows = ArenaRows(r, arena)
        plan = Loopback(H, rb, re)
        for f in range(N):
            render_strip_frame(Stepwise(r) if stepwise else r, rows, plan, frt.CameraController().build_uniform(W / H, f, 2), f, frt, comm=comm)
            if stepwise:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res = r.read_buffer(frt.BUF_RESERVOIR, 0)
        hist = r.read_buffer(frt.BUF_ACCUM, (N - 2) % 2)       # the slot the last frame's "post" exchange moved a row of
        out = {"res0": res, "res1": r.read_buffer(frt.BUF_RESERVOIR, 1), "raw": r.read_buffer(frt.BUF_RAW, 0), "acc": r.read_buffer(frt.BUF_ACCUM, (N - 1) % 2)}
        ok = bool(res[rb:rb + HALO_RESERVOIR].any()) and res[re:re + HALO_RESERVOIR].tobytes() == res[rb:rb + HALO_RESERVOIR].tobytes()
        ok &= bool(hist[rb].any()) and hist[re].tobytes() == hist[rb].tobytes()
        return ok, out, r.stats(), r.phase_rows()

    report = {"nccl": True}
    ok = True
    # (a) a strip with interior rows; (b) a strip too thin to have any (24 rows: every spatial row needs halo rows) — there the edge launches
    # are the whole stage and must still run on the stream the caller ordered behind the transfer (frt_renderer_stream(r, 2)).
    for name, (W, H, rb, re) in {"interior": (640, 360, 96, 240), "thin": (640, 360, 160, 184)}.items():
        N = 4
        ok_rows, got, st, _ = run(W, H, rb, re, N, stepwise=False)
        _, want, _, _ = run(W, H, rb, re, N, stepwise=True)
        same = all(got[k][rb:re].tobytes() == want[k][rb:re].tobytes() for k in got)      # the strip's own rows of every stage's output
        same &= got["res1"][re:re + 4].tobytes() == want["res1"][re:re + 4].tobytes() and bool(want["res1"][re:re + 4].any())      # the "pre" rows landed, and when they should
        report[name] = {"rows_arrive": bool(ok_rows), "async_equals_stepwise": bool(same), "speculated_frames": st["speculated_frames"]}
        ok &= bool(ok_rows) and bool(same)
        # the same frames with the transfers as grouped RCCL launches IN the renderer's edge stream: same rows, same pixels as the stepwise reference
        ok_rows_d, got_d, _, _ = run(W, H, rb, re, N, stepwise=False, comm=direct)
        same_d = all(got_d[k][rb:re].tobytes() == want[k][rb:re].tobytes() for k in got_d) and got_d["res1"][re:re + 4].tobytes() == want["res1"][re:re + 4].tobytes()
        report[name]["direct_rows_arrive"] = bool(ok_rows_d); report[name]["direct_equals_stepwise"] = bool(same_d)
        ok &= bool(ok_rows_d) and bool(same_d)
    report["ok"] = bool(ok)
    direct.destroy()
    print(json.dumps(report))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
