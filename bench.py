#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1]): Cornell Box (src/scene/scenes.rs:9-130), 1920x1080, MAX_DEPTH = 8 (restir.wgsl:5),
one candidate path per pixel per frame with the reference's four stages (G-buffer -> ReSTIR temporal -> ReSTIR spatial + shade
-> post/accumulate), static camera at the initial pose. One "step" = one frame. Synthetic data: the scene is procedural.
Metric (BASELINE.json): Mrays/s = (closest-hit + any-hit rays issued, counted on the device) / wall seconds, and ms/frame.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU, the frame is cut into N horizontal strips, two halo exchanges per frame over RCCL (frt.dist);
total work is fixed ("strong" scaling). Rank 0 prints ONE JSON line.
Extra objects: "roofline" (dominant kernel, HBM bound, algorithmic bytes per SURVEY.md §8d / DESIGN.md §6) and, at N = 1,
"cpu_baseline" (the scalar C++ oracle over the same BVH on the host cores — a reported baseline, never the target).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs on this pool (task environment notes)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, MAX_DEPTH = 1920, 1080, 8
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
STAGES = ("gbuffer", "temporal", "spatial", "post")
# Compulsory per-pixel stream bytes of each kernel in this design: G-buffer write 44; temporal = T-trace (read 36, write the
# 16-byte candidate) + T-merge (read candidate 16, G-buffer 36 + previous 36 + motion 8, previous spatial reservoir 32, write 32) = 212;
# spatial read 36+32, write 32+8; post read 68, write 20.
B_PX = {"gbuffer": 44, "temporal": 212, "spatial": 108, "post": 88}
PMC_JSON = os.path.join(ROOT, "profiles", "r2_pmc.json")     # written by tools/pmc_to_json.py on the GPU box, committed


def source_hash():
    """sha256 over the kernel sources the PMC passes were measured on (csrc/*.hip, *.hpp, the Makefile): a profile of other code is stale."""
    import glob, hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "fast-raytracing-wgpu_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "fast-raytracing-wgpu_amd", "Makefile")]:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())       # (the Makefile: compiler flags change the kernels too)
    return h.hexdigest()[:16]


def load_pmc():
    """Per-kernel counters of the committed rocprofv3 --pmc passes, or None when they were measured on different kernel sources."""
    try:
        d = json.load(open(PMC_JSON))
    except (OSError, ValueError):
        return None
    return d if d.get("source_hash") == source_hash() else None


def cpu_share():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box gives a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(scene, cams, n_frames):
    """Oracle (kind "port": scalar C++ restatement walking the product-built BVH2) on the host cores. Bounded sample."""
    from _oracle import Oracle
    import numpy as np
    orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
    osc = orc.cornell()
    osc.set_bvh(scene.get("bvh2_nodes"), scene.get("bvh2_tri_index"))
    # threads actually used: the detected share, capped at 16 (one GPU's host share on the test pool) unless FRT_CPU_THREADS is set
    cores = int(os.environ.get("FRT_CPU_THREADS", min(cpu_share(), 16)))
    r = osc.renderer(W, H, MAX_DEPTH, True, cores)
    r.render(cams[0])                                     # warm-up frame (also frame 0 of the sequence)
    s0 = r.stats()
    blob = np.concatenate([np.frombuffer(bytes(c), np.uint8) for c in cams[1:1 + n_frames]])
    secs = r.time_frames(blob)
    s1 = r.stats()
    rays = (s1["total"]["closest"] + s1["total"]["any"]) - (s0["total"]["closest"] + s0["total"]["any"])
    per_stage = {}
    for st in STAGES[:3]:
        n = (s1[st]["closest"] + s1[st]["any"]) - (s0[st]["closest"] + s0[st]["any"])
        per_stage[st] = {"nodes_per_ray": (s1[st]["nodes"] - s0[st]["nodes"]) / max(n, 1), "tris_per_ray": (s1[st]["tris"] - s0[st]["tris"]) / max(n, 1)}
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"frames 1..{n_frames} of the same 1920x1080 8-bounce workload after 1 warm-up frame, {secs:.1f} s",
            "ms_per_frame": secs / n_frames * 1e3}, per_stage


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--cpu-frames", type=int, default=12, help="frames of the CPU baseline sample (N = 1 only; 0 disables)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1. nccl (= RCCL over xGMI) is the real thing; gloo stages the halo rows through the "
                         "host and lets several ranks share ONE GPU (set FRT_BENCH_ONE_GPU=1) to rehearse the N > 1 code path on a 1-GPU box")
    a = ap.parse_args()

    import torch
    import frt
    if frt.lib().frt_device_count() < 1:
        raise SystemExit("bench.py: no HIP device — the product has no CPU path")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch N > 1 with torch.distributed.run")
    if os.environ.get("FRT_BENCH_ONE_GPU") == "1":
        local_rank = 0                        # rehearsal: every rank renders on cuda:0 (gloo only)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = f"cuda:{local_rank}" if a.backend == "nccl" else "cpu"

    from frt.dist import StripPlan, ArenaRows, render_strip_frame, balanced_boundaries
    scene = frt.scenes.create_cornell_box()
    nl = scene.num_lights
    total = a.warmup + a.steps
    cam_ctl = frt.CameraController()
    cams = [cam_ctl.build_uniform(W / H, f, nl) for f in range(max(total + 17 + max(8, a.steps), a.cpu_frames + 1))]

    # strips of equal WORK (probe render, identical on every rank), not equal height
    bounds = None
    if world > 1:
        try:
            bounds = balanced_boundaries(frt, scene, W, H, world, max_depth=MAX_DEPTH, device=local_rank)
        except Exception as e:      # equal strips are always valid; balancing is an optimisation
            print(f"[rank {rank}] work-balanced strips unavailable ({e}); using equal strips", file=sys.stderr)
    try:
        plan = StripPlan(H, world, rank, bounds)
    except ValueError as e:     # (cannot happen with balanced_boundaries' own guards; equal strips are always valid)
        print(f"[rank {rank}] {e}; using equal strips", file=sys.stderr)
        bounds = None
        plan = StripPlan(H, world, rank, None)
    nbytes = frt.Renderer.arena_bytes(W, H)
    arena = torch.zeros(nbytes + 256, dtype=torch.uint8, device=f"cuda:{local_rank}")
    off = (-arena.data_ptr()) % 256
    stream = torch.cuda.current_stream()
    # Same instrumentation at every N: the two-stream schedule, NO per-stage events inside the timed region (two event records per stage and
    # frame cost a thin strip 0.04 of its 0.43 ms); the per-stage times of the JSON line come from a short instrumented pass afterwards.
    r = frt.Renderer(scene, W, H, max_depth=MAX_DEPTH, device=local_rank, stream=stream.cuda_stream,
                     rows=(plan.row_begin, plan.row_end) if world > 1 else None,
                     arena=arena.data_ptr() + off, arena_bytes=nbytes, flags=frt.FLAG_PIPELINE)
    rows = ArenaRows(r, arena, staging_device=None if a.backend == "nccl" else "cpu")

    def frame(f):
        if world == 1:
            r.render(cams[f])
        else:
            render_strip_frame(r, rows, plan, cams[f], f, frt)

    for f in range(a.warmup):
        frame(f)
    torch.cuda.synchronize()
    s0 = r.stats()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(a.warmup, total):
        frame(f)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    s1 = r.stats()
    # instrumented pass (outside the timed region): the same frame loop with the per-stage HIP events on
    n_inst = min(16, a.steps)
    r.set_timing(True)
    si0 = r.stats()
    for f in range(total, total + n_inst):
        frame(f)
    torch.cuda.synchronize()
    si1 = r.stats()
    r.set_timing(False)
    first_extra = total + n_inst

    rays = (s1["rays_closest"] + s1["rays_any"]) - (s0["rays_closest"] + s0["rays_any"])
    exposed_ms = None
    if dist:
        # What the halo transfers cost per frame: the same frame loop with the transfers switched off (the pixels of those frames are
        # wrong — neighbours' rows are stale — but the work is the same), timed the same way; the difference is the exposed transfer time.
        class _NoTransfers(StripPlan):
            def transfers(self, frame, when="mid"):
                return []
        quiet = _NoTransfers(H, world, rank, bounds)
        nq = max(8, a.steps // 2)
        dist.barrier(); torch.cuda.synchronize()
        tq = time.perf_counter()
        for f in range(first_extra, first_extra + nq):
            render_strip_frame(r, rows, quiet, cams[f], f, frt)
        torch.cuda.synchronize(); dist.barrier()
        quiet_ms = (time.perf_counter() - tq) / nq * 1e3
        tqm = torch.tensor([quiet_ms], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tqm, op=dist.ReduceOp.MAX)
        exposed_ms = elapsed / a.steps * 1e3 - float(tqm.item())
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n = torch.tensor([rays], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        rays = int(n.item())

    if rank == 0:
        K = a.steps
        ms = [(b - c) / n_inst for b, c in zip(si1["ms_stage"], si0["ms_stage"])]
        ms_merge = (si1["ms_merge"] - si0["ms_merge"]) / n_inst
        stage_rays = [(s1["rays_stage"][i][0] + s1["rays_stage"][i][1] - s0["rays_stage"][i][0] - s0["rays_stage"][i][1]) / K for i in range(4)]
        cpu, per_stage = (None, None)
        if world == 1 and a.cpu_frames > 0:
            cpu, per_stage = cpu_baseline(scene, cams, a.cpu_frames)
        # Algorithmic bytes: rays * (32 B * nodes/ray + 48 B * tris/ray) + pixels * B_px per stage. nodes/ray and tris/ray
        # are per-ray means on the canonical BVH2 measured by the oracle for that stage (same run at N = 1; the committed figures of
        # DESIGN.md §6 otherwise).
        defaults = {"gbuffer": (10.1, 1.7), "temporal": (18.8, 2.2), "spatial": (16.5, 2.1), "post": (0.0, 0.0)}
        px = W * (plan.row_end - plan.row_begin)
        stages = {}
        for i, name in enumerate(STAGES):
            npr, tpr = (per_stage[name]["nodes_per_ray"], per_stage[name]["tris_per_ray"]) if per_stage and name in per_stage else defaults[name]
            algo = stage_rays[i] * (32.0 * npr + 48.0 * tpr) + px * B_PX[name]
            stages[name] = {"event_ms": ms[i], "rays": stage_rays[i], "bytes_per_ray": 32.0 * npr + 48.0 * tpr, "algorithmic_bytes": algo,
                            "logical_GBs_over_event_ms": algo / (ms[i] * 1e-3) / 1e9 if ms[i] > 0 else 0.0}
        stages["temporal"]["merge_event_ms"] = ms_merge
        # The stages of a frame are co-scheduled on two streams (G-buffer + T-trace of frame f+1 and post(f) run beside spatial(f)), so a
        # stage's event time includes the time it shares the chip: the roofline is quoted for the frame's kernel set as a whole, over
        # the wall time of the timed region; the per-stage event times are listed beside it. (Round 1 quoted the spatial stage alone:
        # 7.17 GB / 1.33 ms = 0.68; its whole-frame figure was 11.9 GB / 2.356 ms = 0.63.)
        frame_ms = elapsed / K * 1e3
        algo_frame = sum(v["algorithmic_bytes"] for v in stages.values())
        achieved = algo_frame / (frame_ms * 1e-3) / 1e9
        pmc = load_pmc() if world == 1 else None
        roof = {"bound": "hbm", "kernel": "frame = gbuffer_kernel + pixel_kernel<1> + continue_kernel<1> + merge_kernel + pixel_kernel<2> + continue_kernel<2> + post_kernel, co-scheduled on two streams",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc["hbm_bytes_per_frame"] if pmc else None, "avg_launch_ms": frame_ms, "algorithmic_bytes_per_launch": algo_frame,
                "stages": stages,
                "note": "logical BVH + stream bytes per SURVEY 8(d); the scene (91 KB) is L2-resident, so HBM carries only the per-pixel streams (traffic << algorithmic bytes) and the binding resource is the vector ALU / L2 latency: see hbm_actual and valu_issue"}
        if pmc:
            # HBM-actual: the PMC bytes over this run's frame time. VALU issue: wave-instructions of one frame (deterministic) x the calibrated
            # SIMD cycles per wave-instruction (tools/valu_calib.hip) over the SIMD-cycles the frame lasted.
            roof["hbm_actual"] = {"GBs": pmc["hbm_bytes_per_frame"] / (frame_ms * 1e-3) / 1e9, "frac": pmc["hbm_bytes_per_frame"] / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            simd_cycles = frame_ms * 1e-3 * pmc["shader_clock_ghz"] * 1e9 * pmc["simds"]
            roof["valu_issue"] = {"wave_insts_per_frame": pmc["valu_insts_per_frame"], "cycles_per_wave_inst": pmc["cycles_per_valu_inst"],
                                  "frac": pmc["valu_insts_per_frame"] * pmc["cycles_per_valu_inst"] / simd_cycles,
                                  "lane_utilisation": pmc.get("lane_utilisation")}
            roof["pmc_source"] = {"file": "profiles/r2_pmc.json", "source_hash": pmc["source_hash"], "git_head": pmc.get("git_head")}
        elif world == 1:
            roof["pmc_source"] = "profiles/r2_pmc.json is absent or was measured on other kernel sources: traffic / hbm_actual / valu_issue withheld"
        par = "1 GPU, two-stream schedule"
        if world > 1:
            par = (f"{world} work-balanced image strips {bounds}; per frame 2 halo exchanges per neighbour ({'RCCL' if a.backend == 'nccl' else 'gloo rehearsal'}): "
                   "12 reservoir rows overlapped with the spatial stage's interior rows, 1 accumulation row posted a frame early; "
                   f"exposed transfer time {exposed_ms:.3f} ms/frame (frame loop with vs without the transfers)")
        out = {
            "metric": "Mrays/sec, 1920x1080 8-bounce Cornell Box", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": frame_ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Cornell Box 1920x1080, MAX_DEPTH 8, 1 candidate path/pixel/frame, 4-stage ReSTIR-PT frame (BASELINE.json configs[1])",
                       "rays_per_frame": rays / a.steps, "parallelism": par,
                       "speculated_frames": s1["speculated_frames"] - s0["speculated_frames"], "queue_overflow": s1["queue_overflow"],
                       "exchange_exposed_ms": exposed_ms},
            "roofline": roof,
            "stage_ms": dict(zip(STAGES, ms)),
            "stage_ms_note": f"per-stage HIP event times of a separate instrumented pass of {n_inst} frames after the timed region (the timed region records no events)",
        }
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
