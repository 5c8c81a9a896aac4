"""Row-tile sharding of one frame over the ranks of a node, and its reassembly on rank 0.

The reference walks the frame one scanline per macrotask (spanish(y), main.js:183-201); every pixel
is independent, so the frame shards with no exchange during rendering.  Here the frame is cut into
tiles of `tile_rows` rows dealt round-robin to the ranks (sky rows are several times cheaper than
floor rows, so contiguous bands would be badly balanced); each rank renders its tiles contiguously
into a band, ONE gather (RCCL when the tensors are on GPUs, gloo in the CPU tests) brings the bands to
rank 0, and one de-interleave pass puts the rows in frame order.

Throughput form (what bench.py times): a step is a BATCH of `world` frames.  Every rank renders its row
tiles of all `world` frames (one launch, rt_render_batch_device), then ONE all-to-all sends the band of
frame f to rank f, so each rank reassembles one whole frame per step.  On a point-to-point xGMI full mesh
that uses every directed link at once (each carries 1/world of a frame), where a gather to a single root
would funnel every band through the root's 7 inbound links and leave the other 49 idle.  The bands cross the
links as RGB24 (`channels=3`): the alpha byte is the constant 255 in the reference (main.js:198), so it is not
shipped; the de-interleave pass on the receiving rank restores it.

Used by bench.py (GPU, backend "nccl") and by tests/test_shard_gloo.py (CPU, backend "gloo",
world_size 2 and 3) — the same code path, only the band producer differs.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class TilePlan:
    w: int
    h: int
    tile_rows: int
    world: int
    channels: int = 4          # bytes per pixel of a band: 4 = RGBA8, 3 = RGB24 (RT_FLAG_RGB24; needs w % 4 == 0)

    def __post_init__(self):
        if self.channels not in (3, 4) or (self.channels == 3 and self.w % 4):
            raise ValueError("TilePlan: channels must be 4, or 3 with a frame width that is a multiple of 4")

    @property
    def n_tiles(self):
        return (self.h + self.tile_rows - 1) // self.tile_rows

    @property
    def tiles_per_rank(self):
        return (self.n_tiles + self.world - 1) // self.world

    @property
    def band_rows(self):
        return self.tiles_per_rank * self.tile_rows

    @property
    def band_bytes(self):
        return self.band_rows * self.w * self.channels

    def rt_tiles(self, rank):
        """(tile_rows, tile_first, tile_stride, n_tiles) for rt_render_tiles_device on `rank`."""
        return (self.tile_rows, rank, self.world, self.tiles_per_rank)

    def rows_of(self, rank):
        """Frame rows rank `rank` owns, in band order (rows past the frame's end are left out)."""
        rows = []
        for i in range(self.tiles_per_rank):
            t = rank + i * self.world
            rows.extend(range(t * self.tile_rows, min(self.h, (t + 1) * self.tile_rows)))
        return rows

    def pixels_of(self, rank):
        return len(self.rows_of(rank)) * self.w


def gather_bands(band, gathered, dst=0, async_op=False):
    """One gather of every rank's band ([band_rows, w, channels] uint8) into `gathered` ([world, band_rows, w, channels],
    rank `dst` only; None elsewhere).  Returns the work handle when async_op."""
    import torch.distributed as dist
    recv = list(gathered.unbind(0)) if dist.get_rank() == dst else None
    return dist.gather(band, recv, dst=dst, async_op=async_op)


def exchange_bands(send, recv, async_op=False):
    """One all-to-all over a batch of frames: send[f] holds this rank's band(s) for rank f (shape
    [world, band_rows, w, channels], or [world, B, band_rows, w, channels] when B steps share one exchange);
    afterwards recv[g] holds rank g's band(s) of the frame(s) this rank owns - the layout deinterleave() expects
    (recv[:, j] for the j-th frame)."""
    import torch.distributed as dist
    return dist.all_to_all_single(recv.view(-1), send.view(-1), async_op=async_op)


def deinterleave(plan, gathered, frame, lib=None, device_index=0, stream=0):
    """gathered [world, band_rows, w, channels] -> frame [h, w, 4] (RGBA8, alpha 255) in row order.
    `gathered` may be a view with any stride between ranks (e.g. buf[:, j] of a [world, B, band_rows, w, channels]
    exchange buffer that holds B frames per rank); each rank's band itself must be contiguous.
    GPU tensors: the library's HBM->HBM kernel (rt_deinterleave[_rgb24]_device) on `stream`.
    CPU tensors (tests): the same permutation expressed with torch views."""
    if tuple(gathered.shape) != (plan.world, plan.band_rows, plan.w, plan.channels) or not gathered[0].is_contiguous():
        raise ValueError("deinterleave: expected [world, band_rows, w, channels] with contiguous bands")
    if gathered.is_cuda:
        fn = lib.rt_deinterleave_device if plan.channels == 4 else lib.rt_deinterleave_rgb24_device
        rank_stride = gathered.stride(0) * gathered.element_size() if plan.world > 1 else plan.band_bytes
        rc = fn(device_index, gathered.data_ptr(), frame.data_ptr(), plan.w, plan.h, plan.tile_rows, plan.world, rank_stride, stream)
        if rc != 0:
            raise RuntimeError("rt_deinterleave_device: " + lib.rt_last_error().decode())
        return frame
    v = gathered.reshape(plan.world, plan.tiles_per_rank, plan.tile_rows, plan.w, plan.channels).permute(1, 0, 2, 3, 4)
    rows = v.reshape(-1, plan.w, plan.channels)[:plan.h]
    if plan.channels == 4:
        frame.copy_(rows)
    else:
        frame[..., :3] = rows
        frame[..., 3] = 255
    return frame
