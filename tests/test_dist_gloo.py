"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the strip partition + halo exchange (frt.dist), with the oracle as
the per-rank renderer, must reproduce the single-rank image bit for bit."""
import json
import os
import socket
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def run_ranks(mode, world, tmp_path, extra=()):
    port = _free_port()
    out = os.path.join(str(tmp_path), "res.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), "--mode", mode, "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--out", out, *extra], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return json.load(open(out))


@pytest.mark.parametrize("world,H", [(2, 64), (3, 72)])
def test_strip_partition_matches_single_rank(frt, orc, tmp_path, world, H):
    res = run_ranks("oracle", world, tmp_path, ("--H", str(H), "--W", "80", "--frames", "4"))
    assert res["ok"], res


@pytest.mark.parametrize("world,H", [(2, 64), (3, 96)])
def test_moving_camera_strips_match_single_rank(frt, orc, tmp_path, world, H):
    """with a moving camera the temporal stage reprojects into, and post fetches history from, rows of other strips;
    a motion halo of K rows of previous-frame state (second exchange, before the temporal stage) restores bit equality."""
    res = run_ranks("oracle", world, tmp_path, ("--H", str(H), "--W", "80", "--frames", "5", "--moving", "6"))
    assert res["ok"], res


def test_strip_plan_geometry(frt):
    from frt.dist import StripPlan
    plans = [StripPlan(1080, 8, k) for k in range(8)]
    assert plans[0].row_begin == 0 and plans[-1].row_end == 1080
    assert all(a.row_end == b.row_begin for a, b in zip(plans, plans[1:]))
    assert all(p.row_end - p.row_begin == 135 for p in plans)
    # every send has a matching receive of the same rows on the peer
    for f in (0, 3):
        for when in ("mid", "post"):
            sends = {(p.rank, peer, buf, idx, s) for p in plans for peer, buf, idx, s, r in p.transfers(f, when)}
            recvs = {(peer, p.rank, buf, idx, r) for p in plans for peer, buf, idx, s, r in p.transfers(f, when)}
            assert sends == recvs
    # "mid": 12 rows of temporal reservoirs per neighbour, every frame; "post": 1 row of the previous accumulation, from frame 1 on
    assert len(plans[0].transfers(0)) == 1 and len(plans[3].transfers(2)) == 2
    assert plans[3].transfers(0, "post") == [] and len(plans[3].transfers(2, "post")) == 2 and plans[3].transfers(2, "pre") == []
    assert {(buf, idx, s[1] - s[0]) for _, buf, idx, s, _ in plans[3].transfers(2, "post")} == {(7, 1, 1)}
    with pytest.raises(ValueError):
        StripPlan(64, 8, 0)
    # moving camera: a "pre" exchange of the previous spatial reservoirs before T-merge; the accumulation rows grow to K + 1
    mp = [StripPlan(1080, 8, k, motion_halo=16) for k in range(8)]
    assert mp[2].transfers(0, "pre") == [] and len(mp[2].transfers(3, "pre")) == 2 and len(mp[2].transfers(3)) == 2
    for when in ("pre", "mid", "post"):
        sends = {(q.rank, peer, buf, idx, s) for q in mp for peer, buf, idx, s, r in q.transfers(3, when)}
        recvs = {(peer, q.rank, buf, idx, r) for q in mp for peer, buf, idx, s, r in q.transfers(3, when)}
        assert sends == recvs
    assert {(buf, idx, s[1] - s[0]) for _, buf, idx, s, _ in mp[2].transfers(3, "pre")} == {(4, 1, 16)}
    assert {(buf, idx, s[1] - s[0]) for _, buf, idx, s, _ in mp[2].transfers(3, "post")} == {(7, 0, 17)}
    with pytest.raises(ValueError):
        StripPlan(1080, 8, 0, motion_halo=200)
    # unequal (work-balanced) strips
    p = [StripPlan(100, 3, k, [0, 50, 70, 100]) for k in range(3)]
    assert (p[1].row_begin, p[1].row_end) == (50, 70)
    sends = {(q.rank, peer, buf, idx, s) for q in p for peer, buf, idx, s, r in q.transfers(1)}
    recvs = {(peer, q.rank, buf, idx, r) for q in p for peer, buf, idx, s, r in q.transfers(1)}
    assert sends == recvs
    with pytest.raises(ValueError):
        StripPlan(100, 3, 0, [0, 50, 55, 100])


def test_pre_rows_are_gated_on_frames_rendered_not_on_frame_count(frt):
    """A host that resets frame_count while the camera moves (state.rs:152) passes frame = 0 every frame: the previous spatial reservoirs exist all
    the same and T-merge reprojects into them, so the "pre" transfer follows `serial` (frames rendered since creation); post ignores its history at
    frame_count 0 (post.wgsl:187), so "post" follows `frame`."""
    from frt.dist import StripPlan, BUF_RESERVOIR
    p = StripPlan(96, 3, 1, motion_halo=6)
    assert p.transfers(0, "pre") == [] and p.transfers(0, "pre", serial=0) == []
    pre = p.transfers(0, "pre", serial=3)
    assert len(pre) == 2 and all(t[1] == BUF_RESERVOIR and t[2] == 1 and t[3][1] - t[3][0] == 6 for t in pre)
    assert p.transfers(0, "post", serial=3) == [] and len(p.transfers(2, "post", serial=3)) == 2
    assert len(p.transfers(5, "pre")) == 2            # a counter that never restarts: serial defaults to frame
    assert StripPlan(96, 3, 1).transfers(4, "pre", serial=4) == []      # static camera: no motion halo, no "pre" rows
