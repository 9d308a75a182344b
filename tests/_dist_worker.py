"""Worker for the multi-rank strip tests. mode=oracle: CPU oracle strips (gloo); mode=gpu: product strips on cuda:0 (gloo, host-staged).
Rank 0 compares the gathered image with a single-rank render and writes {"ok": bool, ...} to --out."""
import argparse
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleRows:
    def __init__(self, r):
        self.r = r

    def rows(self, buf, index, y0, y1):
        import torch
        return torch.from_numpy(self.r.read_rows(buf, index, y0, y1).reshape(-1).copy())

    def recv_buffer(self, buf, index, y0, y1):
        import torch
        return torch.empty((y1 - y0) * self.r.w * {4: 32, 7: 16}[buf], dtype=torch.uint8)

    def store(self, buf, index, y0, y1, t):
        self.r.write_rows(buf, index, y0, y1, t.numpy())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="oracle"); ap.add_argument("--rank", type=int); ap.add_argument("--world", type=int)
    ap.add_argument("--port", type=int); ap.add_argument("--out"); ap.add_argument("--W", type=int, default=96)
    ap.add_argument("--H", type=int, default=64); ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--balanced", type=int, default=0)
    ap.add_argument("--moving", type=int, default=0, help="motion halo rows K > 0: moving camera (tests/_scenes.py), two exchanges per frame")
    ap.add_argument("--flags", type=int, default=0, help="frt.Renderer flags of the gpu mode (8 = the side-stream schedule bench.py uses)")
    a = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
    import torch
    import torch.distributed as dist
    import frt
    from frt.dist import StripPlan, exchange_halos, gather_strips
    from _oracle import Oracle
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    W, H, N = a.W, a.H, a.frames
    fs = frt.scenes.create_cornell_box()
    bounds = None
    if a.balanced:
        from frt.dist import balanced_boundaries
        bounds = balanced_boundaries(frt, fs, W, H, a.world, bands=8)
    K = a.moving
    plan = StripPlan(H, a.world, a.rank, bounds, motion_halo=K)
    if K:
        import _scenes
        cams = _scenes.moving_camera_uniforms(frt, W / H, 2, N)
    else:
        cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(N)]
    orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
    osc = orc.cornell(); osc.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    rb, re = plan.row_begin, plan.row_end
    if a.mode == "oracle":
        r = osc.renderer(W, H, 8, True, 2)
        acc = OracleRows(r)
        hg = max(12, K)
        for f in range(N):
            exchange_halos(acc, plan, f, when="pre")
            r.render_phases(cams[f], 1, max(rb - hg, 0) if a.world > 1 else 0, min(re + hg, H) if a.world > 1 else H)
            r.render_phases(cams[f], 2, rb, re)
            exchange_halos(acc, plan, f, when=("mid", "post"))      # ONE batch per frame, as frt.dist.render_strip_frame posts it: the "post" rows ride with the "mid" rows
            r.render_phases(cams[f], 4, max(rb - 2, 0), min(re + 2, H))
            r.render_phases(cams[f], 8, rb, re)
            r.end_frame()
        mine = torch.from_numpy(r.read_rows(7, (N - 1) % 2, rb, re).copy())
        rays = None
    else:
        torch.cuda.set_device(0)
        from frt.dist import ArenaRows
        nbytes = frt.Renderer.arena_bytes(W, H)
        arena = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda:0")
        off = (-arena.data_ptr()) % 256
        r = frt.Renderer(fs, W, H, rows=(rb, re), arena=arena.data_ptr() + off, arena_bytes=nbytes,
                         stream=torch.cuda.current_stream().cuda_stream, motion_halo=K, flags=a.flags)
        acc = ArenaRows(r, arena, staging_device="cpu")
        from frt.dist import render_strip_frame
        for f in range(N):
            render_strip_frame(r, acc, plan, cams[f], f, frt)      # the loop bench.py --gpus N runs (host-staged transport here)
        torch.cuda.synchronize()
        mine = torch.from_numpy(r.read_rows(7, (N - 1) % 2, rb, re).copy())
        st = r.stats(); rays = st["rays_closest"] + st["rays_any"]
        assert st["halo_overflow"] == 0, st
    full = gather_strips(mine, plan).numpy()
    res = {"ok": True}
    if a.rank == 0:
        ref = osc.renderer(W, H, 8, True, 16 if W * H > 500000 else 4)
        for f in range(N):
            ref.render(cams[f])
        want = ref.read(7, (N - 1) % 2)
        res = {"bounds": plan.boundaries, "ok": bool(full.tobytes() == want.tobytes()), "mismatch_pixels": int((full.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())}
        if rays is not None:
            so = ref.stats()["total"]; res["oracle_rays"] = so["closest"] + so["any"]
    if rays is not None:
        t = torch.tensor([rays], dtype=torch.int64); dist.all_reduce(t); res["rays_all_ranks"] = int(t.item())
    dist.barrier()
    if a.rank == 0:
        json.dump(res, open(a.out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
