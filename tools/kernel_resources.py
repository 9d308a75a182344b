#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch budget of the gfx950 code object, as the compiler reports it
(-Rpass-analysis=kernel-resource-usage on csrc/frt_kernels.hip with the Makefile's flags; no GPU needed).

  python tools/kernel_resources.py [--extra "-DFOO=1"] [--filter pixel_kernel]

One line per kernel: VGPRs, AGPRs, SGPRs, spilled SGPRs / VGPRs, scratch bytes per lane, LDS bytes, waves per SIMD.
The committed copies live under profiles/ (rN_kernel_resources.txt)."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fast-raytracing-wgpu_amd")


def makefile_flags():
    txt = open(os.path.join(PKG, "Makefile")).read().replace("\\\n", " ")
    m = re.search(r"^FLAGS\s*=\s*(.*)$", txt, re.M)
    return m.group(1).replace("$(ARCH)", "gfx950").split()


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
        return [o.split("(")[0] for o in out[:len(names)]]
    except Exception:
        return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--extra", default="", help="extra compiler flags")
    ap.add_argument("--filter", default="", help="only kernels whose demangled name contains this")
    ap.add_argument("--src", default="csrc/frt_kernels.hip")
    a = ap.parse_args()
    cmd = ["/opt/rocm/bin/hipcc"] + [f for f in makefile_flags() if f not in ("-fPIC", "-Wall")] + a.extra.split() + \
          ["-x", "hip", a.src, "--offload-device-only", "-c", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, cwd=PKG, capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr)
        raise SystemExit(p.returncode)
    kernels, cur = [], None
    for line in p.stderr.split("\n"):
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            kernels.append(cur)
        elif cur is not None:
            cur[k] = v
    names = demangle([k["name"] for k in kernels])
    print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'LDS':>7s} {'waves/SIMD':>10s}")
    for k, n in sorted(zip(kernels, names), key=lambda t: t[1]):
        if a.filter and a.filter not in n:
            continue
        n = n.replace("frt::", "")
        print(f"{n:44s} {k.get('VGPRs', '?'):>5s} {k.get('AGPRs', '?'):>5s} {k.get('TotalSGPRs', '?'):>5s} {k.get('SGPRs Spill', '?'):>6s} "
              f"{k.get('VGPRs Spill', '?'):>6s} {k.get('ScratchSize [bytes/lane]', '?'):>7s} {k.get('LDS Size [bytes/block]', '?'):>7s} {k.get('Occupancy [waves/SIMD]', '?'):>10s}")


if __name__ == "__main__":
    main()
