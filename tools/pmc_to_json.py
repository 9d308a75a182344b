"""Turn the rocprofv3 --pmc passes of tools/profile_all.sh (gpurun_out/pmc_<tag>_*) and the VALU calibration (gpurun_out/valu_calib.txt)
into the JSON bench.py reads for its roofline block: per-kernel per-launch means, HBM bytes and VALU wave-instructions per frame, the
calibrated issue cost, and the hashes of the device sources / code objects they were measured on (bench.py flags the numbers `stale` when neither matches).
Run HERE (the repository with .git), after the gpurun call that produced the passes:
    python tools/pmc_to_json.py r4 > profiles/r4_pmc.json"""
import collections, csv, glob, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash, code_object_hash, device_sources     # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "run"
tab = collections.defaultdict(dict)
launches = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*"))):
    if not os.path.isdir(d):
        continue
    # gpurun MERGES a call's output into the local gpurun_out/: a tag used twice leaves two runs' files in one directory — only the newest counts
    for f in sorted(glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "frt::" in k:
                name = k.replace("frt::", "").split("(")[0].replace("void ", "")
                for c, x in v.items():
                    tab[name][c] = sum(x) / len(x)
                    launches[name] = len(x)
frames = launches.get("merge_kernel") or (max(launches.values()) if launches else 0)       # T-merge runs exactly once per frame
kern = {}
for k, v in tab.items():
    e = dict(v)
    lp = launches[k] / frames if frames else 0
    # one launch per frame (the last speculated G-buffer + T-trace have no frame of their own); the continuation kernels run once per cut
    e["launches_per_frame"] = float(round(lp)) if lp > 1.2 else min(1.0, lp)
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e["hbm_bytes"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024       # gfx950: FETCH_SIZE counts 2 x 32 B units per KiB reported (MI355X_MICROARCH.md §HBM)
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        e["lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
        e["wait_frac"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    kern[k] = e
# the per-CU load path of the one-stream frame (tools/pmc_ta.sh <tag>ta): vector L1 (TCP) hit rate, texture addresser / data return busy
l1 = {}
ta = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}ta_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "frt::" in k:
                name = k.replace("frt::", "").split("(")[0].replace("void ", "")
                for c, x in v.items():
                    ta[name][c] = sum(x) / len(x)
for k, v in ta.items():
    e = {}
    if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        e["l1_hit_rate"] = 1.0 - v.get("TCP_TCC_READ_REQ_sum", 0.0) / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if v.get("TCP_TCC_READ_REQ_sum"):
            e["l1_miss_latency_cycles"] = v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / v["TCP_TCC_READ_REQ_sum"]
    if v.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; TA_BUSY_avr is a per-TA average; TD_TD_BUSY_sum sums 256 CUs
        e["ta_busy_frac"] = v.get("TA_BUSY_avr", 0.0) / (v["GRBM_GUI_ACTIVE"] / 8.0)
        e["td_busy_frac"] = v.get("TD_TD_BUSY_sum", 0.0) / 256.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
    if e:
        l1[k] = {kk: round(vv, 4) for kk, vv in e.items()}
calib = {}
try:
    for line in open(os.path.join(ROOT, "gpurun_out", "valu_calib.txt")):
        if line.startswith("["):      # (the half-wave runs of tools/valu_calib.hip: not the calibration)
            continue
        m = re.search(r"(independent|dependent)\s+K=(\d).*shader clock ([\d.]+) GHz.*a wave needs ([\d.]+) cycles.*-> ([\d.]+) SIMD cycles", line)
        if m:
            calib[f"{m.group(1)}_K{m.group(2)}"] = {"shader_clock_ghz": float(m.group(3)), "own_cycles_per_inst": float(m.group(4)), "simd_cycles_per_wave_inst": float(m.group(5))}
except OSError:
    pass
sat = calib.get("independent_K8") or calib.get("independent_K4") or {"simd_cycles_per_wave_inst": 2.0, "shader_clock_ghz": 2.4}
out = {
    "source_hash": source_hash(),                 # device sources (bench.device_sources) + compiler flags
    "code_object_hash": code_object_hash(),       # .hip_fatbin of lib/libfrt.so as it stands in this tree (the library the passes ran)
    "device_sources": device_sources(),
    "git_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
    "workload": "bench.py --cpu-frames 0 --no-4k --steps 16 --warmup 4 (1920x1080, two-stream schedule), per-launch means",
    "kernels": kern,
    "hbm_bytes_per_frame": sum(e.get("hbm_bytes", 0) * e["launches_per_frame"] for e in kern.values()),
    "valu_insts_per_frame": sum(e.get("SQ_INSTS_VALU", 0) * e["launches_per_frame"] for e in kern.values()),
    "lane_utilisation": {k: round(e["lane_utilisation"], 3) for k, e in kern.items() if "lane_utilisation" in e},
    "cycles_per_valu_inst": sat["simd_cycles_per_wave_inst"],     # saturated SIMD (8 independent waves): the issue cost of one wave-instruction
    "shader_clock_ghz": sat["shader_clock_ghz"],
    "simds": 1024,
    "calibration": calib,
    "l1": l1,
}
print(json.dumps(out, indent=1))
