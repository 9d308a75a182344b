"""tools/bvh_quality — the host model of the lockstep BVH walk (profiles/r3_experiments/traversal_in_situ.md).

What a ray hits must not depend on how a wave schedules its lanes' steps (while-while, voting), on idle lanes helping (a ray's root children
dealt to several lanes), or on the tree (the insertion-optimised BVH2 the product builds for larger scenes): the model reports a checksum of the hits, and the step counts it prints are
only meaningful if those agree. Also pins the headline numbers of the notes loosely (a bounce ray: ~3.6 node steps per lane, ~11 per wave)."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "_build", "bvh_quality")
LIB = os.path.join(ROOT, "fast-raytracing-wgpu_amd", "lib")


def _build():
    src = os.path.join(ROOT, "tools", "bvh_quality.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", src, "-I" + os.path.join(ROOT, "include"), "-L" + LIB, "-lfrt", "-Wl,-rpath," + LIB, "-o", EXE], check=True)


def _run(*args):
    p = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def _checksum(out):
    return re.search(r"hits checksum (\d+) (\d+) (\d+)", out).groups()


def test_hits_do_not_depend_on_schedule_helpers_or_tree(frt):
    _build()
    # scene, tiles, insertion passes, policy, vote weight, presence, helper lanes
    ref = _checksum(_run("cornell", 120, 0, 0, 16, 0.6, 0))
    assert _checksum(_run("cornell", 120, 0, 1, 24, 0.6, 0)) == ref        # the wave votes for its next step
    assert _checksum(_run("cornell", 120, 0, 2, 8, 0.6, 0)) == ref         # leaf step as soon as fewer than 8 lanes hold a node
    assert _checksum(_run("cornell", 120, 0, 0, 16, 0.6, 1)) == ref        # idle lanes help
    assert _checksum(_run("cornell", 120, 3, 0, 16, 0.6, 0)) == ref        # insertion-optimised tree
    assert int(ref[1]) > 0 and int(ref[2]) > 0


def test_model_numbers_of_the_notes(frt):
    _build()
    out = _run("cornell", 300)
    m = re.search(r"bounce 1\s+rays\s+\d+\s+per lane-ray: nodes\s+([\d.]+) tris\s+([\d.]+) \| per wave-ray: node steps\s+([\d.]+) leaf steps\s+([\d.]+)", out)
    lane_nodes, lane_tris, wave_nodes, wave_leaves = map(float, m.groups())
    assert 2.8 < lane_nodes < 4.5 and 1.5 < lane_tris < 3.0
    assert 8.0 < wave_nodes < 13.0 and 3.0 < wave_leaves < 5.0          # a wave executes about three times the node steps one of its lanes needs
    # the product's quad tree as the model decodes it (round 4: folded by the surface-area programme: 326 nodes instead of the greedy fold's 390)
    assert "stack need 24" in out and "326 quad nodes" in out
    # the 8-wide tree (csrc/frt_bvh8.hpp) walked by the model's trace8: the same hits, a third fewer steps per wave-ray, a stack of 5
    assert re.search(r"8-wide tree: 174 nodes .* stack need 5", out)
    q = re.search(r"hits checksum (\d+) (\d+) (\d+)", out).groups()
    w = re.search(r"hits checksum \(8-wide\) (\d+) (\d+) (\d+)", out).groups()
    assert q == w
    m8 = re.findall(r"bounce 1\s+rays\s+\d+\s+per lane-ray: nodes\s+([\d.]+) leaf steps\s+([\d.]+) tris\s+([\d.]+) \| per wave-ray: node steps\s+([\d.]+) leaf steps\s+([\d.]+)", out)
    lane8, _, tris8, wave8, leaves8 = map(float, m8[0])
    assert lane8 < 0.6 * lane_nodes and tris8 < 1.3 * lane_tris and wave8 + leaves8 < wave_nodes + wave_leaves
