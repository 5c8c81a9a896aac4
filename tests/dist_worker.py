"""world_size-N CPU worker for tests/test_shard_gloo.py (backend gloo).

Each rank produces its band of interleaved row tiles (here with the ORACLE's C restatement, because
there is no GPU in the build container — the sharding/gather/de-interleave code under test is
html5-canvas-raytracer_amd/shard.py, the same code bench.py runs over RCCL), the bands are gathered to
rank 0 and de-interleaved, and rank 0 compares the frame with the whole frame rendered at once.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as ou  # noqa: E402
import rt_host  # noqa: E402
import shard  # noqa: E402


def main():
    scene, w, h, tile_rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = shard.TilePlan(w, h, tile_rows, world)
    blob = rt_host.flatten_scene(rt_host.load_scene(scene))
    band = torch.zeros((plan.band_rows, w, 4), dtype=torch.uint8)
    rows = plan.rows_of(rank)
    if rows:
        data = np.frombuffer(ou.c_oracle_rows(blob, w, h, rows), dtype=np.uint8).reshape(len(rows), w, 4)
        band[:len(rows)] = torch.from_numpy(data.copy())
    gathered = torch.empty((world, plan.band_rows, w, 4), dtype=torch.uint8) if rank == 0 else None
    work = shard.gather_bands(band, gathered, dst=0, async_op=True)
    work.wait()
    ok = 1
    if rank == 0:
        frame = torch.empty((h, w, 4), dtype=torch.uint8)
        shard.deinterleave(plan, gathered, frame)
        whole = np.frombuffer(ou.c_oracle_render(blob, w, h), dtype=np.uint8).reshape(h, w, 4)
        ok = int(np.array_equal(frame.numpy(), whole))
        assert sum(plan.pixels_of(r) for r in range(world)) == w * h
        print("DIST_RESULT world=%d identical=%d" % (world, ok), flush=True)
    # ---- throughput form: a batch of `world` frames, one all-to-all, every rank reassembles one frame.
    # The frames differ (frame f = the scene at depth 1+f) so that a mix-up of frames cannot go unnoticed.
    # The bands travel as RGB24 (what bench.py does over RCCL); deinterleave() restores the alpha byte.
    plan = shard.TilePlan(w, h, tile_rows, world, channels=3)
    scene_d = rt_host.load_scene(scene)
    blobs = []
    for f in range(world):
        sc = dict(scene_d)
        sc["segs"] = 1 + (f % 3)
        blobs.append(rt_host.flatten_scene(sc))
    # as in bench.py, the bands of EVERY = 2 consecutive steps share one exchange: send[f, j] is this rank's band of the
    # frame that rank f owns in step j; step j's frames differ from step 0's (the scene with one more bounce)
    EVERY = 2
    blobs2 = []
    for f in range(world):
        sc = dict(scene_d)
        sc["segs"] = 2 + (f % 3)
        blobs2.append(rt_host.flatten_scene(sc))
    per_step = [blobs, blobs2]
    send = torch.zeros((world, EVERY, plan.band_rows, w, 3), dtype=torch.uint8)
    for j in range(EVERY):
        for f in range(world):
            if rows:
                data = np.frombuffer(ou.c_oracle_rows(per_step[j][f], w, h, rows), dtype=np.uint8).reshape(len(rows), w, 4)
                send[f, j, :len(rows)] = torch.from_numpy(data[..., :3].copy())
    recv = torch.empty_like(send)
    shard.exchange_bands(send, recv, async_op=True).wait()
    same = 1
    for j in range(EVERY):
        mine = torch.empty((h, w, 4), dtype=torch.uint8)
        shard.deinterleave(plan, recv[:, j], mine)                 # a strided view: the rank stride is EVERY bands
        whole = np.frombuffer(ou.c_oracle_render(per_step[j][rank], w, h), dtype=np.uint8).reshape(h, w, 4)
        same &= int(np.array_equal(mine.numpy(), whole))
    ok2 = torch.tensor([same])
    dist.all_reduce(ok2, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_RESULT_A2A world=%d identical=%d" % (world, int(ok2.item())), flush=True)
    ok = ok and int(ok2.item())
    t = torch.tensor([ok])
    dist.broadcast(t, src=0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(t.item()) == 1 else 1)


if __name__ == "__main__":
    main()
