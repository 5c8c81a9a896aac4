"""Multi-process CPU tests of the N>1 path (gloo, world_size 2 and 3): interleaved row tiles ->
one gather -> de-interleave reproduces the single-rank frame byte for byte."""
import os
import socket
import subprocess
import sys

import pytest

import oracle_util as ou
import shard

ROOT = ou.ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_tile_plan_covers_every_row_once():
    for w, h, tr, world in [(64, 40, 16, 2), (64, 41, 8, 3), (32, 7, 8, 2), (32, 2160, 16, 8), (32, 100, 16, 8)]:
        plan = shard.TilePlan(w, h, tr, world)
        rows = sorted(r for g in range(world) for r in plan.rows_of(g))
        assert rows == list(range(h)), (w, h, tr, world)
        assert plan.band_rows * world >= h
        for g in range(world):
            assert plan.rt_tiles(g) == (tr, g, world, plan.tiles_per_rank)
        # RGB24 bands: same rows, three quarters of the bytes
        p3 = shard.TilePlan(w, h, tr, world, channels=3)
        assert p3.band_bytes * 4 == plan.band_bytes * 3 and p3.rows_of(0) == plan.rows_of(0)
    with pytest.raises(ValueError):
        shard.TilePlan(30, 8, 8, 2, channels=3)          # RGB24 needs w % 4 == 0


def test_deinterleave_rgb24_on_cpu_restores_alpha():
    import torch
    plan = shard.TilePlan(8, 21, 4, 3, channels=3)
    frame = torch.randint(0, 256, (plan.h, plan.w, 4), dtype=torch.uint8)
    frame[..., 3] = 255
    gathered = torch.zeros((plan.world, plan.band_rows, plan.w, 3), dtype=torch.uint8)
    for g in range(plan.world):
        for k, row in enumerate(plan.rows_of(g)):
            gathered[g, k] = frame[row, :, :3]
    out = torch.zeros_like(frame)
    shard.deinterleave(plan, gathered, out)
    assert torch.equal(out, frame)


@pytest.mark.parametrize("world,scene,w,h,tile_rows", [(2, "h8", 64, 48, 8), (2, "cfg2", 40, 37, 16), (3, "default14", 32, 26, 8)])
def test_gather_and_deinterleave_world(world, scene, w, h, tile_rows, built):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), scene, str(w), str(h), str(tile_rows)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "DIST_RESULT world=%d identical=1" % world in r.stdout          # gather-to-root form (rt_render's plan)
    assert "DIST_RESULT_A2A world=%d identical=1" % world in r.stdout      # batch + all-to-all form (bench.py's plan)
