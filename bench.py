#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1]): Cornell Box (src/scene/scenes.rs:9-130), 1920x1080, MAX_DEPTH = 8 (restir.wgsl:5),
one candidate path per pixel per frame with the reference's four stages (G-buffer -> ReSTIR temporal -> ReSTIR spatial + shade
-> post/accumulate), static camera at the initial pose. One "step" = one frame. Synthetic data: the scene is procedural.
Metric (BASELINE.json): Mrays/s = (closest-hit + any-hit rays issued, counted on the device) / wall seconds, and ms/frame.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU, the frame is cut into N horizontal strips, one batched halo exchange per frame and neighbour over RCCL (frt.dist);
total work is fixed ("strong" scaling). Rank 0 prints ONE JSON line. Either launch form works: with RANK / WORLD_SIZE in the
environment this process IS a rank; without them `--gpus N` starts its own N rank processes (fresh children, before this
process has touched the GPU), relays rank 0's line and exits non-zero if any rank fails.
`--native`: ONE process drives the N GPUs through the C ABI's frt_multi_renderer (include/frt.h), no torch.distributed.
Extra objects: "roofline" (what binds the frame, from the committed counters) and, at N = 1,
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
W4K, H4K = 3840, 2160          # BASELINE.json configs[2]
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
L2_PEAK_GBS = 34500.0          # same guide, §L2: ~34.5 TB/s aggregate over the 8 XCDs
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12      # 256 CUs x 4 SIMD-32 x 2.4 GHz: 78.6 T lane-operations/s (= 157.3 TFLOP/s of fma)
STAGES = ("gbuffer", "temporal", "spatial", "post")
# Compulsory per-pixel stream bytes of each kernel in this design: G-buffer write 44; temporal = T-trace (read 36, write the
# 16-byte candidate) + T-merge (read candidate 16, G-buffer 36 + previous 36 + motion 8, previous spatial reservoir 32, write 32) = 212;
# spatial read 36+32, write 32+8; post read 68, write 20.
B_PX = {"gbuffer": 44, "temporal": 212, "spatial": 108, "post": 88}
def _latest_pmc_json():
    """profiles/r<N>_pmc.json of the highest round N (written by tools/pmc_to_json.py from the --pmc passes of tools/profile_all.sh, committed)."""
    import glob, re
    best = (-1, os.path.join(ROOT, "profiles", "r0_pmc.json"))
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")):
        m = re.fullmatch(r"r(\d+)_pmc\.json", os.path.basename(f))
        if m:
            best = max(best, (int(m.group(1)), f))
    return best[1]


PMC_JSON = _latest_pmc_json()


def device_sources():
    """The files that decide the DEVICE code of lib/libfrt.so: csrc/frt_kernels.hip (the only translation unit with __global__ functions) and
    what it includes, transitively. csrc/experiments/ is compiled into lib/libfrt_exp.so only (FRT_EXPERIMENTS) and host-only files
    (frt_bvh_opt.hpp, frt_scene.cpp, frt_loader.cpp, frt_renderer.hip ...) are not device code: editing them does not stale a profile."""
    import re
    csrc = os.path.join(ROOT, "fast-raytracing-wgpu_amd", "csrc")
    seen, todo = [], ["frt_kernels.hip"]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.append(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(os.path.join(csrc, f)).read(), re.M):
            inc = os.path.normpath(os.path.join(os.path.dirname(f), inc))
            if not inc.startswith("experiments") and os.path.exists(os.path.join(csrc, inc)):
                todo.append(inc)
    return sorted(seen)


def source_hash():
    """sha256 over the device sources (device_sources()) and the compiler flags of the Makefile: what the committed counters were measured on."""
    import hashlib, re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "fast-raytracing-wgpu_amd", "csrc")
    for f in device_sources():
        h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    mk = open(os.path.join(ROOT, "fast-raytracing-wgpu_amd", "Makefile")).read()
    for var in ("ARCH", "FLAGS"):      # (the flags change the kernels too; comments and rules of the Makefile do not)
        m = re.search(r"^%s\s*\??=\s*((?:.*\\\n)*.*)$" % var, mk, re.M)
        h.update((var + "=" + " ".join(m.group(1).replace("\\\n", " ").split())).encode() if m else b"?")
    return h.hexdigest()[:16]


def code_object_hash(path=None):
    """sha256 of the gfx950 code objects inside lib/libfrt.so (its .hip_fatbin section): host edits never change it, device edits always do.
    None when the library (or the section) is absent."""
    import hashlib, struct
    path = path or os.path.join(ROOT, "fast-raytracing-wgpu_amd", "lib", "libfrt.so")
    try:
        with open(path, "rb") as f:
            d = f.read()
    except OSError:
        return None
    if d[:4] != b"\x7fELF" or d[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", d, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", d, 0x3A)
    sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize)
    str_off = sec(shstrndx)[4]
    for i in range(shnum):
        name, _, _, _, off, size = sec(i)[:6]
        end = d.index(b"\0", str_off + name)
        if d[str_off + name:end] == b".hip_fatbin":
            return hashlib.sha256(d[off:off + size]).hexdigest()[:16]
    return None


def load_pmc():
    """Per-kernel counters of the committed rocprofv3 --pmc passes. d["stale"] is False when they were measured on these kernels: the device
    sources + flags hash alike, or the library's code objects do (either is proof; a host-only edit changes neither). A stale profile is
    still reported — with the flag — so that a record never loses its numbers; tests/test_bench_profile.py fails on the CPU box the moment a
    commit stales the profile."""
    try:
        d = json.load(open(PMC_JSON))
    except (OSError, ValueError):
        return None
    co = code_object_hash()
    d["stale"] = not (d.get("source_hash") == source_hash() or (co is not None and d.get("code_object_hash") == co))
    return d


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--cpu-frames", type=int, default=12, help="frames of the CPU baseline sample (N = 1 only; 0 disables)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1. nccl (= RCCL over xGMI) is the real thing; gloo stages the halo rows through the "
                         "host and lets several ranks share ONE GPU (set FRT_BENCH_ONE_GPU=1) to rehearse the N > 1 code path on a 1-GPU box")
    ap.add_argument("--native", action="store_true",
                    help="N > 1 in ONE process through frt_multi_renderer (C ABI): strips on N devices, peer copies for the halos; "
                         "FRT_BENCH_ONE_GPU=1 maps every logical device to ordinal 0")
    ap.add_argument("--no-4k", action="store_true", help="skip the extra configs[2] (3840x2160) measurement")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous plumbing only: every rank joins the process group and reports itself; nothing is rendered, no GPU is touched (CPU test)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------ self-launch
def self_launch(a, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU), relay rank 0's JSON line.
    Runs BEFORE torch / frt are imported: this process never initialises the GPU (and never execs), the children start clean."""
    import socket
    import subprocess
    import threading

    def launch_once():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        t_start = time.time()
        procs = []
        for rank in range(a.gpus):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FRT_BENCH_SELF_LAUNCHED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=subprocess.PIPE if rank == 0 else sys.stderr))
        deadline = time.time() + float(os.environ.get("FRT_BENCH_TIMEOUT", "1500"))
        failed = None
        box = {"out": b""}

        def drain():
            box["out"] = procs[0].stdout.read()
        t = threading.Thread(target=drain, daemon=True)
        t.start()
        while True:
            codes = [p.poll() for p in procs]
            bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                failed = "timeout"
                break
            time.sleep(0.05)
        if failed:
            for p in procs:            # the exact processes started above, nothing else
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        t.join(timeout=10)
        return failed, box["out"], port, time.time() - t_start

    # The free port is found by bind-then-close: another process can take it before rank 0 binds MASTER_PORT. A launch that dies within its
    # first seconds without a result line is tried ONCE more on a fresh port (a rank that fails later fails for its own reasons).
    failed, out0, port, took = launch_once()
    if failed and failed != "timeout" and took < 30.0 and not out0.strip():
        print(f"bench.py: {failed} {took:.1f} s after launch on 127.0.0.1:{port}; one more try on a fresh port", file=sys.stderr)
        failed, out0, port, took = launch_once()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    if failed:
        raise SystemExit(f"bench.py: {failed} (launched {a.gpus} ranks on 127.0.0.1:{port})")
    return 0


_RESULT_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout. Libraries loaded later write there too (gloo announces its peers on std::cout, the HIP runtime
    complains about a missing amdgpu.ids): from here on file descriptor 1 IS stderr, and emit_result() writes the line to the original stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit_result(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def rank_env():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))


def dry_run(a):
    """Rendezvous only (gloo, CPU): proves that the launcher started `world` ranks that can talk. One JSON line from rank 0."""
    world, rank, local_rank = rank_env()
    info = {"rank": rank, "local_rank": local_rank, "pid": os.getpid()}
    ranks = [info]
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks = [None] * world
        dist.all_gather_object(ranks, info)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_result({"dry_run": True, "n_gpus": world, "config": {"ranks": ranks, "backend": "gloo" if world > 1 else None,
                                                                   "self_launched": os.environ.get("FRT_BENCH_SELF_LAUNCHED") == "1"}})


# ------------------------------------------------------------------------------------------------------------ measurement
def timed_loop(frame, first, warmup, steps, sync, barrier, stats):
    """`warmup` untimed frames, then exactly `steps` frames between barrier + synchronize on both sides. Returns (seconds, stats before, stats after)."""
    for f in range(first, first + warmup):
        frame(f)
    sync()
    s0 = stats()
    barrier()
    sync()
    t0 = time.perf_counter()
    for f in range(first + warmup, first + warmup + steps):
        frame(f)
    sync()
    barrier()
    sync()
    return time.perf_counter() - t0, s0, stats()


def rays_of(s0, s1):
    return (s1["rays_closest"] + s1["rays_any"]) - (s0["rays_closest"] + s0["rays_any"])


def roofline_block(world, frame_ms, stages, px_rows):
    """What binds the frame, read from the committed counters (PMC_JSON; flagged `stale` when measured on other device code).
    The scene is L2-resident, so neither roofline the task names (hbm, mfma) binds: the frame is bound by the LATENCY of dependent vector
    instructions between L1 round trips at 4 waves per SIMD. `bound`/`achieved`/`peak`/`frac` therefore describe the vector ALU
    (wave-instructions x 64 lanes per second against 256 CUs x 4 SIMD-32 x 2.4 GHz); `useful_lane_frac` = that x the lane utilisation is the
    number to drive up. The §8(d) algorithmic-bytes figure is carried as `logical_*`, the PMC HBM bytes as `traffic` / `hbm_actual`."""
    algo_frame = sum(v["algorithmic_bytes"] for v in stages.values())
    logical = algo_frame / (frame_ms * 1e-3) / 1e9
    pmc = load_pmc() if world == 1 else None
    roof = {"bound": "valu-issue latency (vector ALU; not hbm, not mfma: the scene lives in L2)",
            "kernel": "frame = gbuffer_kernel + pixel_kernel<1> + continue_kernel<1> + merge_kernel + pixel_kernel<2> + continue_kernel<2> + post_kernel, co-scheduled on two streams",
            "achieved": None, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s (VALU wave-instructions x 64 lanes)", "frac": None, "traffic": None,
            "avg_launch_ms": frame_ms,
            "logical_GBs": logical, "logical_frac": logical / HBM_PEAK_GBS, "logical_frac_of_l2_peak": logical / L2_PEAK_GBS,
            "algorithmic_bytes_per_launch": algo_frame, "stages": stages,
            "note": "logical_* = SURVEY 8(d) algorithmic bytes (BVH2 nodes and triangles per ray as the oracle counts them + compulsory per-pixel streams) over the frame time, "
                    "against the HBM peak (logical_frac) and the aggregate L2 peak; served by the CUs' L1 / the XCDs' L2, so it is NOT an HBM roofline"}
    if pmc:
        lane_ops = pmc["valu_insts_per_frame"] * 64.0
        roof["achieved"] = lane_ops / (frame_ms * 1e-3) / 1e12
        roof["frac"] = roof["achieved"] / VALU_PEAK_TLANEOPS
        roof["traffic"] = pmc["hbm_bytes_per_frame"]
        hbm = pmc["hbm_bytes_per_frame"] / (frame_ms * 1e-3) / 1e9
        roof["hbm_actual"] = {"GBs": hbm, "frac": hbm / HBM_PEAK_GBS, "traffic_over_algorithmic": pmc["hbm_bytes_per_frame"] / algo_frame}
        simd_cycles = frame_ms * 1e-3 * pmc["shader_clock_ghz"] * 1e9 * pmc["simds"]
        issue = pmc["valu_insts_per_frame"] * pmc["cycles_per_valu_inst"] / simd_cycles
        lu = pmc.get("lane_utilisation") or {}
        # lane utilisation of the frame: VALU thread-cycles over 64 x VALU instruction-cycles, summed over the frame's launches
        num = sum(k.get("SQ_THREAD_CYCLES_VALU", 0.0) * k["launches_per_frame"] for k in pmc["kernels"].values())
        den = sum(64.0 * k.get("SQ_ACTIVE_INST_VALU", 0.0) * k["launches_per_frame"] for k in pmc["kernels"].values())
        frame_lu = num / den if den else None
        roof["valu_issue"] = {"wave_insts_per_frame": pmc["valu_insts_per_frame"], "cycles_per_wave_inst": pmc["cycles_per_valu_inst"],
                              "shader_clock_ghz": pmc["shader_clock_ghz"], "frac_at_measured_clock": issue,
                              "lane_utilisation": lu, "lane_utilisation_frame": frame_lu}
        roof["useful_lane_frac"] = issue * frame_lu if frame_lu else None
        if pmc.get("l1"):
            roof["l1"] = pmc["l1"]          # TCP hit rates / TA busy of the traced kernels (tools/pmc_ta.sh)
        roof["pmc_source"] = {"file": os.path.relpath(PMC_JSON, ROOT), "source_hash": pmc["source_hash"], "code_object_hash": pmc.get("code_object_hash"),
                              "git_head": pmc.get("git_head"), "stale": pmc["stale"],
                              "note": "counters of the committed rocprofv3 --pmc passes x this run's frame time; stale = measured on other device code than this library's "
                                      "(device sources + flags and code objects both hash differently)"}
    elif world == 1:
        roof["pmc_source"] = f"{os.path.relpath(PMC_JSON, ROOT)} is absent: achieved / frac / traffic / hbm_actual / valu_issue withheld"
    else:
        roof["pmc_source"] = "counters are collected at N = 1 only"
    return roof


def main():
    a = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and env_world is None and not a.native:
        return self_launch(a, sys.argv[1:])          # nothing GPU-related has been imported or called yet
    claim_stdout()                                   # (a rank, or the one process of N = 1 / --native: only the JSON line reaches stdout)
    if a.dry_run:
        return dry_run(a)
    if a.native:
        return main_native(a)

    import torch
    import frt
    world, rank, local_rank = rank_env()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if frt.lib().frt_device_count() < 1:
        raise SystemExit("bench.py: no HIP device — the product has no CPU path")
    one_gpu = os.environ.get("FRT_BENCH_ONE_GPU") == "1"
    if one_gpu:
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
    n_inst = min(16, a.steps)
    nq = max(8, a.steps // 2)
    cam_ctl = frt.CameraController()
    cams = [cam_ctl.build_uniform(W / H, f, nl) for f in range(max(total + n_inst + nq + 1, a.cpu_frames + 1))]

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
    stream = torch.cuda.current_stream()

    def make(width, height, pl):
        nbytes = frt.Renderer.arena_bytes(width, height)
        arena = torch.zeros(nbytes + 256, dtype=torch.uint8, device=f"cuda:{local_rank}")
        off = (-arena.data_ptr()) % 256
        # Same instrumentation at every N: the two-stream schedule, NO per-stage events inside the timed region (two event records per stage and
        # frame cost a thin strip 0.04 of its 0.43 ms); the per-stage times of the JSON line come from a short instrumented pass afterwards.
        if world == 1:
            r = frt.Renderer(scene, width, height, max_depth=MAX_DEPTH, device=local_rank, stream=stream.cuda_stream,
                             arena=arena.data_ptr() + off, arena_bytes=nbytes, flags=frt.FLAG_PIPELINE)
        else:
            # A strip renderer under the pipeline enqueues on FOUR streams (main, ahead, two edge streams) and the HIP runtime has four hardware queues per
            # process (GPU_MAX_HW_QUEUES): created one after the other they get a queue each. Handed torch's current stream as its main stream — created
            # long before, with torch's and RCCL's own streams in between — the main and the ahead stream landed on ONE queue and the two-stream schedule
            # ran in series: 0.56 instead of 0.34 ms per frame for a 1/8 strip (tools/rccl_strip_time.py, profiles/r4_experiments/rccl_strips.md).
            # So the renderer creates its main stream itself and torch (the RCCL transfers of frt.dist) is told to order its work on that stream.
            torch.cuda.synchronize()           # (the arena was zeroed on torch's current stream)
            r = frt.Renderer(scene, width, height, max_depth=MAX_DEPTH, device=local_rank, rows=(pl.row_begin, pl.row_end),
                             arena=arena.data_ptr() + off, arena_bytes=nbytes, flags=frt.FLAG_PIPELINE)
            torch.cuda.set_stream(torch.cuda.ExternalStream(r.stream_handle(0), device=f"cuda:{local_rank}"))
        return r, ArenaRows(r, arena, staging_device=None if a.backend == "nccl" else "cpu")

    # The halo rows travel as grouped RCCL send / recv launches placed directly IN the renderer's streams (frt.rccl over librccl.so; torch.distributed only
    # carries the communicator's id): torch's own point-to-point operations put eight event markers on the main stream, a hop to torch's RCCL stream and a
    # hop back between T-merge and the edge rows (one-GPU rehearsal, 1/8 strip: 0.437 -> 0.395 ms per frame, host time per frame 0.23 -> 0.11 ms;
    # profiles/r4_experiments/rccl_strips.md). Every rank votes: if the communicator cannot be had everywhere, every rank uses torch's operations.
    comm = None
    if dist and world > 1 and a.backend == "nccl" and os.environ.get("FRT_BENCH_TORCH_P2P") != "1":
        import frt.rccl
        ok = 1
        try:
            frt.rccl.lib()
        except Exception as e:      # noqa: BLE001
            ok = 0
            print(f"[rank {rank}] librccl.so cannot be loaded directly ({e}); torch.distributed point-to-point operations instead", file=sys.stderr)
        vote = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if int(vote.item()) == 1:      # every rank can call RCCL: the communicator's creation is collective from here on
            try:
                comm = frt.rccl.Comm.create(rank, world, local_rank)
            except Exception as e:      # noqa: BLE001
                ok = 0
                print(f"[rank {rank}] ncclCommInitRank failed ({e}); torch.distributed point-to-point operations instead", file=sys.stderr)
            vote = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)
            if int(vote.item()) != 1:
                if comm is not None:
                    comm.destroy()
                comm = None
    if dist and world > 1 and a.backend == "nccl" and comm is None:
        # RCCL creates its point-to-point communicator and stream at the first transfer. Do that NOW, before the renderer's streams exist: the HIP runtime
        # hands out its four hardware queues in stream-creation order, and RCCL's stream then shares one with the renderer's LAST stream (the second edge
        # stream) instead of with its main stream, where every transfer sat between T-merge and the interior launch (one-GPU rehearsal: 0.465 -> 0.436 ms per
        # 1/8-strip frame, profiles/r4_experiments/rccl_strips.md). It also keeps the communicator's set-up out of the warm-up frames. Same pattern as the
        # frame loop's exchange: one batch, upper neighbour first.
        warm_s, warm_r = torch.zeros(256, device=f"cuda:{local_rank}"), torch.zeros(2, 256, device=f"cuda:{local_rank}")
        ops = []
        for k, peer in enumerate(p for p in (rank - 1, rank + 1) if 0 <= p < world):
            ops += [dist.P2POp(dist.isend, warm_s, peer), dist.P2POp(dist.irecv, warm_r[k], peer)]
        if ops:
            for w_ in dist.batch_isend_irecv(ops):
                w_.wait()
        torch.cuda.synchronize()
    r, rows = make(W, H, plan)

    def frame_fn(rr, rws, pl, cam_list):
        if world == 1:
            return lambda f: rr.render(cam_list[f])
        return lambda f: render_strip_frame(rr, rws, pl, cam_list[f], f, frt, comm=comm)

    frame = frame_fn(r, rows, plan, cams)
    sync = torch.cuda.synchronize
    barrier = dist.barrier if dist else (lambda: None)
    elapsed, s0, s1 = timed_loop(frame, 0, a.warmup, a.steps, sync, barrier, r.stats)
    # instrumented pass (outside the timed region): the same frame loop with the per-stage HIP events on
    r.set_timing(True)
    si0 = r.stats()
    for f in range(total, total + n_inst):
        frame(f)
    torch.cuda.synchronize()
    si1 = r.stats()
    r.set_timing(False)
    first_extra = total + n_inst

    def reduce_time_rays(el, n_rays):
        if not dist:
            return el, n_rays
        t = torch.tensor([el], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([n_rays], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        return float(t.item()), int(n.item())

    elapsed, rays = reduce_time_rays(elapsed, rays_of(s0, s1))      # MAX over ranks of the timed region, SUM of the rays: from here on job-wide figures
    exposed_ms, gather = None, None
    if dist:
        # What the halo transfers cost per frame: the same frame loop with the transfers switched off (the pixels of those frames are
        # wrong — neighbours' rows are stale — but the work is the same), timed the same way and reduced the same way (MAX over ranks);
        # the difference of the two job-wide times is the exposed transfer time.
        class _NoTransfers(StripPlan):
            def transfers(self, frame, when="mid"):
                return []
        quiet = _NoTransfers(H, world, rank, bounds)
        dist.barrier(); torch.cuda.synchronize()
        tq = time.perf_counter()
        for f in range(first_extra, first_extra + nq):
            render_strip_frame(r, rows, quiet, cams[f], f, frt, comm=comm)
        torch.cuda.synchronize(); dist.barrier()
        quiet_s, _ = reduce_time_rays(time.perf_counter() - tq, 0)
        exposed_ms = elapsed / a.steps * 1e3 - quiet_s / nq * 1e3
        # "tiles gathered over RCCL/xGMI" (north_star; the reference reads ONE texture per presented frame, state.rs:226-278): one all-gather of
        # the display strips over the real backend, outside the timed region — a host that presents every frame pays this once per frame.
        from frt.dist import gather_strips
        rh = plan.row_end - plan.row_begin
        gms = []
        for _ in range(3):
            mine = rows.rows(frt.BUF_DISPLAY, 0, plan.row_begin, plan.row_end).view(rh, W * 4)      # (gloo rehearsal: a host copy of the rows)
            torch.cuda.synchronize(); dist.barrier()
            tg = time.perf_counter()
            full = gather_strips(mine, plan)
            torch.cuda.synchronize(); dist.barrier()
            gms.append(reduce_time_rays(time.perf_counter() - tg, 0)[0] * 1e3)
        gather = {"ms": min(gms), "ms_first": gms[0], "bytes": W * H * 4, "what": "all-gather of the RGBA8 display strips (frt.dist.gather_strips), "
                  + ("RCCL" if a.backend == "nccl" else "gloo rehearsal through the host"), "shape": list(full.shape)}
        del full, mine

    # who ran: every rank reports its device and rows; a sum over the process group proves that `world` ranks took part in a collective
    me = {"rank": rank, "device": local_rank, "device_name": torch.cuda.get_device_name(local_rank), "rows": [plan.row_begin, plan.row_end],
          "rays_per_frame": rays_of(s0, s1) / a.steps, "pid": os.getpid()}
    ranks, allreduce_ranks = [me], 1
    if dist:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        one = torch.ones(1, dtype=torch.int64, device=comm_dev)
        dist.all_reduce(one)
        allreduce_ranks = int(one.item())

    # configs[2]: the same scene at 3840x2160, the workload BASELINE.json tiles over 8 GPUs (extra key; the headline stays configs[1])
    extra4k = None
    if not a.no_4k:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())      # (the strip renderer's main stream goes away with the renderer)
        del r, rows, frame
        torch.cuda.empty_cache()
        plan4 = StripPlan(H4K, world, rank, [b * 2 for b in bounds] if bounds else None)
        k4, w4 = max(4, min(16, a.steps)), 4
        cams4 = [frt.CameraController().build_uniform(W4K / H4K, f, nl) for f in range(k4 + w4)]
        r4, rows4 = make(W4K, H4K, plan4)
        el4, q0, q1 = timed_loop(frame_fn(r4, rows4, plan4, cams4), 0, w4, k4, sync, barrier, r4.stats)
        el4, rays4 = reduce_time_rays(el4, rays_of(q0, q1))
        extra4k = {"workload": "Cornell Box 3840x2160, MAX_DEPTH 8 (BASELINE.json configs[2])", "value": rays4 / el4 / 1e6, "unit": "Mrays/s",
                   "ms_per_step": el4 / k4 * 1e3, "steps": k4, "warmup": w4, "rays_per_frame": rays4 / k4,
                   "rows": [plan4.row_begin, plan4.row_end] if world == 1 else plan4.boundaries}
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())
        del r4, rows4

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
            stages[name] = {"event_ms": ms[i], "rays": stage_rays[i], "bytes_per_ray": 32.0 * npr + 48.0 * tpr, "algorithmic_bytes": algo}
        stages["temporal"]["merge_event_ms"] = ms_merge
        frame_ms = elapsed / K * 1e3
        roof = roofline_block(world, frame_ms, stages, px)
        par = "1 GPU, two-stream schedule"
        if world > 1:
            par = (f"{world} work-balanced image strips {bounds}; per frame ONE batched halo exchange with each neighbour ({('RCCL, grouped send / recv placed in the renderer edge stream (frt.rccl)' if comm is not None else 'RCCL through torch.distributed point-to-point operations') if a.backend == 'nccl' else 'gloo rehearsal'}), behind T-merge and overlapped with the "
                   "spatial stage's interior rows: 12 reservoir rows + 1 row of the previous accumulation; "
                   f"exposed transfer time {exposed_ms:.3f} ms/frame (frame loop with vs without the transfers)")
        out = {
            "metric": "Mrays/sec, 1920x1080 8-bounce Cornell Box", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": frame_ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Cornell Box 1920x1080, MAX_DEPTH 8, 1 candidate path/pixel/frame, 4-stage ReSTIR-PT frame (BASELINE.json configs[1])",
                       "rays_per_frame": rays / a.steps, "parallelism": par,
                       "backend": (dist.get_backend() if dist else None), "process_group_world_size": (dist.get_world_size() if dist else 1),
                       "allreduce_of_ones": allreduce_ranks, "ranks": ranks, "one_gpu_rehearsal": one_gpu,
                       "self_launched": os.environ.get("FRT_BENCH_SELF_LAUNCHED") == "1",
                       "speculated_frames": s1["speculated_frames"] - s0["speculated_frames"], "queue_overflow": s1["queue_overflow"],
                       "exchange_exposed_ms": exposed_ms, "gather_ms": gather["ms"] if gather else None, "gather": gather,
                       "exchange": (None if not dist else "frt.rccl: grouped ncclSend / ncclRecv in the renderer's edge stream" if comm is not None
                                    else "torch.distributed batch_isend_irecv" + (" (gloo rehearsal)" if a.backend != "nccl" else ""))},
            "roofline": roof,
            "stage_ms": dict(zip(STAGES, ms)),
            "stage_ms_note": f"per-stage HIP event times of a separate instrumented pass of {n_inst} frames after the timed region (the timed region records no events)",
        }
        if extra4k:
            out["config2_4k"] = extra4k
        if cpu:
            out["cpu_baseline"] = cpu
        emit_result(out)
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main_native(a):
    """One process, N devices, through the C ABI's multi-device renderer (include/frt.h: frt_multi_renderer_*): what a Rust host calls."""
    import frt
    if frt.lib().frt_device_count() < 1:
        raise SystemExit("bench.py: no HIP device — the product has no CPU path")
    one_gpu = os.environ.get("FRT_BENCH_ONE_GPU") == "1"
    devices = [0] * a.gpus if one_gpu else list(range(a.gpus))
    scene = frt.scenes.create_cornell_box()
    nl = scene.num_lights

    def run(width, height, warmup, steps):
        mr = frt.MultiRenderer(scene, width, height, devices, max_depth=MAX_DEPTH)
        cams = [frt.CameraController().build_uniform(width / height, f, nl) for f in range(warmup + steps)]
        el, s0, s1 = timed_loop(lambda f: mr.render(cams[f]), 0, warmup, steps, mr.sync, lambda: None, mr.stats)
        return mr, el, rays_of(s0, s1)

    mr, elapsed, rays = run(W, H, a.warmup, a.steps)
    # the presented image: device-side gather of the strips' rows (peer copies over xGMI) on the first device + ONE device-to-host copy
    rd = []
    for _ in range(3):
        mr.sync()
        t0 = time.perf_counter()
        img = mr.read_display()
        rd.append((time.perf_counter() - t0) * 1e3)
    peer = mr.peer_access()
    out = {"metric": "Mrays/sec, 1920x1080 8-bounce Cornell Box", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
           "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "Cornell Box 1920x1080, MAX_DEPTH 8, 1 candidate path/pixel/frame, 4-stage ReSTIR-PT frame (BASELINE.json configs[1])",
                      "rays_per_frame": rays / a.steps,
                      "parallelism": f"frt_multi_renderer: ONE process, {a.gpus} strip renderers on devices {devices}, halo rows by hipMemcpyPeerAsync, boundaries {mr.boundaries()}",
                      "one_gpu_rehearsal": one_gpu, "native": True, "peer_access": peer,
                      "gather_ms": min(rd), "gather": {"ms": min(rd), "ms_first": rd[0], "bytes": int(img.nbytes),
                                                       "what": "frt_multi_renderer_read_display: device-side gather of the strips (peer copies) + one device-to-host copy"}}}
    del mr
    if not a.no_4k:
        k4 = max(4, min(16, a.steps))
        mr4, el4, rays4 = run(W4K, H4K, 4, k4)
        out["config2_4k"] = {"workload": "Cornell Box 3840x2160, MAX_DEPTH 8 (BASELINE.json configs[2])", "value": rays4 / el4 / 1e6, "unit": "Mrays/s",
                             "ms_per_step": el4 / k4 * 1e3, "steps": k4, "warmup": 4, "rays_per_frame": rays4 / k4, "rows": mr4.boundaries()}
    emit_result(out)


if __name__ == "__main__":
    sys.exit(main())
