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


def test_direct_rccl_binding_loads_and_orders_its_transfers():
    """frt/rccl.py (RCCL called directly on the renderer's streams; what bench.py's ranks use on a GPU node): the library torch ships exports what the
    binding calls, a unique id is 128 bytes, and frt.dist.exchange_direct posts every send and every receive of a frame's ("mid", "post") transfers in ONE
    group, sends and receives of a peer in the same order on both sides (a grouped ncclSend / ncclRecv pair matches by order per peer). No GPU needed:
    the communicator is a recorder."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-raytracing-wgpu_amd"))
    import torch
    import frt.rccl as R
    from frt.dist import StripPlan, exchange_direct, BUF_RESERVOIR, BUF_ACCUM, HALO_RESERVOIR
    L = R.lib()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(L, name), name
    assert len(R.unique_id()) == R.NCCL_UNIQUE_ID_BYTES == 128

    W, H, world = 64, 96, 3
    bpp = {BUF_RESERVOIR: 32, BUF_ACCUM: 16}

    class Rows:          # the arena as frt.dist.ArenaRows exposes it: one tensor per (buffer, index), full-frame pitch
        def __init__(self):
            self.t = {}
        def _view(self, buf, index, y0, y1):
            t = self.t.setdefault((buf, index), torch.zeros(H * W * bpp[buf], dtype=torch.uint8))
            return t[y0 * W * bpp[buf]:y1 * W * bpp[buf]]

    class Recorder:
        def __init__(self): self.calls = []
        def exchange(self, sends, recvs, stream): self.calls.append((sends, recvs, stream))

    posted = {}
    for rank in range(world):
        rec, rows = Recorder(), Rows()
        plan = StripPlan(H, world, rank)
        assert exchange_direct(rec, rows, plan, 0, "pre", 7) is False and rec.calls == []            # static camera: no "pre" rows
        assert exchange_direct(rec, rows, plan, 2, ("mid", "post"), 7) is True
        assert len(rec.calls) == 1 and rec.calls[0][2] == 7                                           # ONE group, in the stream it was given
        sends, recvs, _ = rec.calls[0]
        peers = [p for p in (rank - 1, rank + 1) if 0 <= p < world]
        assert [p for _, _, p in sends] == peers + peers == [p for _, _, p in recvs]                   # "mid" rows of every neighbour, then "post" rows
        n_mid, n_post = HALO_RESERVOIR * W * 32, 1 * W * 16
        assert [n for _, n, _ in sends] == [n_mid] * len(peers) + [n_post] * len(peers) == [n for _, n, _ in recvs]
        posted[rank] = (sends, recvs)
    for a in range(world - 1):       # what a sends to a + 1, a + 1 receives from a: same sizes in the same order
        to_b = [n for _, n, p in posted[a][0] if p == a + 1]
        from_a = [n for _, n, p in posted[a + 1][1] if p == a]
        assert to_b == from_a and len(to_b) == 2
