"""Multi-GPU frame partition: interleaved image tiles, one process per GPU.

Tile t of the frame (row-major over ceil(W/ts) x ceil(H/ts) tiles) belongs to
rank t % world.  Every rank renders its tiles into a compact buffer (its k-th
tile at [k*ts*ts*3, ...), rows top-down), the buffers are gathered to rank 0
(the only exchange step of the path: W*H*12/world bytes per rank) and scattered
into the Screen-layout frame.  The scene is replicated on every GPU.

The numpy functions below state the layout on the CPU (used by the gloo tests
and as documentation); on the GPU the same layout is written by
rtx_render_tiles_device and undone by rtx_detile_device.
"""
import numpy as np


def tile_grid(width, height, tile_size):
    return (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size


def tiles_per_rank(width, height, tile_size, world):
    tx, ty = tile_grid(width, height, tile_size)
    return (tx * ty + world - 1) // world


def extract_tiles(frame, tile_size, rank, world):
    """This rank's compact tile buffer cut from a full Screen-layout frame
    (h, w, 3); padding outside the frame is 0."""
    h, w, _ = frame.shape
    tx, ty = tile_grid(w, h, tile_size)
    out = np.zeros((tiles_per_rank(w, h, tile_size, world), tile_size, tile_size, 3), frame.dtype)
    for k, t in enumerate(range(rank, tx * ty, world)):
        x0, y0 = (t % tx) * tile_size, (t // tx) * tile_size
        blk = frame[y0:y0 + tile_size, x0:x0 + tile_size]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def assemble_tiles(all_tiles, width, height, tile_size, world):
    """Inverse of extract_tiles over the rank-major stack of every rank's buffer."""
    tx, ty = tile_grid(width, height, tile_size)
    per = tiles_per_rank(width, height, tile_size, world)
    all_tiles = np.asarray(all_tiles).reshape(world, per, tile_size, tile_size, 3)
    frame = np.zeros((height, width, 3), all_tiles.dtype)
    for t in range(tx * ty):
        x0, y0 = (t % tx) * tile_size, (t // tx) * tile_size
        blk = all_tiles[t % world, t // world]
        frame[y0:y0 + tile_size, x0:x0 + tile_size] = blk[:height - y0, :width - x0]
    return frame


def gather_tiles(local_tiles, dst=0, out=None):
    """torch.distributed gather of every rank's compact tile buffer to `dst`
    (RCCL over xGMI for GPU tensors under the nccl backend, gloo for CPU
    tensors).  `out` is an optional preallocated (world, *local.shape) tensor on
    dst.  Returns the rank-major stack on dst and None elsewhere.  With a
    CPU-only backend and GPU tensors the tiles are staged through the host."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    staged = local_tiles
    if local_tiles.is_cuda and dist.get_backend() == "gloo":
        staged = local_tiles.cpu()
    if rank == dst:
        if out is None or out.device != staged.device:
            buf = torch.empty((world,) + tuple(staged.shape), dtype=staged.dtype, device=staged.device)
        else:
            buf = out
        dist.gather(staged, list(buf.unbind(0)), dst=dst)
        if out is not None and buf is not out:
            out.copy_(buf)
            return out
        return buf
    dist.gather(staged, None, dst=dst)
    return None


class FramePipeline:
    """Pipelined frame assembly: the gather of frame i runs on the collective's
    own stream while frame i + 1 is being rendered, and frame i is scattered into
    the Screen-layout image once frame i + 1 has been queued.

        pipe = FramePipeline(make_buffer, render, assemble)
        for _ in range(n): pipe.submit()
        pipe.drain()

    `make_buffer()` returns a fresh compact tile tensor; `render(tiles)` queues
    this rank's tiles of the next frame into it; `assemble(stack)` (called on dst
    only) scatters the gathered rank-major stack.  Every frame submitted is
    gathered and assembled, in order; drain() finishes the last one.  With a
    backend that cannot take the tensors as they are (gloo and GPU tensors) the
    frames are gathered one after the other through gather_tiles.

    `slots` > 1 keeps that many frames in flight on the GPU: frame i is rendered,
    gathered and assembled in slot i % slots, and `enter(slot)` must return a
    context manager that makes the slot current for the calls inside it (the
    library's frame slot, rtx_select_slot, and the stream the collective orders
    itself against).  Frames of different slots run on the GPU at the same time;
    within a slot they follow one another."""

    def __init__(self, make_buffer, render, assemble, dst=0, slots=1, enter=None):
        import contextlib
        import torch
        import torch.distributed as dist
        self.render, self.assemble, self.dst = render, assemble, dst
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.slots = max(1, int(slots))
        self.enter = enter if enter is not None else (lambda slot: contextlib.nullcontext())
        self.nbuf = max(2, self.slots)
        self.tiles = [make_buffer() for _ in range(self.nbuf)]
        t = self.tiles[0]
        self.overlapped = not (t.is_cuda and dist.get_backend() == "gloo")
        self.stacks = [torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device) if self.rank == dst else None for _ in range(self.nbuf)]
        self.pending = None   # (work, buffer index, slot) of the gather in flight
        self.frames = 0

    def _finish(self):
        if self.pending is None:
            return
        work, k, slot = self.pending
        self.pending = None
        with self.enter(slot):
            if work is not None:
                work.wait()       # a device-side wait under the nccl backend: the host does not block
            if self.rank == self.dst:
                self.assemble(self.stacks[k])

    def submit(self):
        import torch.distributed as dist
        k, slot = self.frames % self.nbuf, self.frames % self.slots
        self.frames += 1
        with self.enter(slot):
            self.render(self.tiles[k])
            if not self.overlapped:
                gather_tiles(self.tiles[k], dst=self.dst, out=self.stacks[k])
                work = None
            else:
                outs = list(self.stacks[k].unbind(0)) if self.rank == self.dst else None
                work = dist.gather(self.tiles[k], outs, dst=self.dst, async_op=True)
        if not self.overlapped:
            self.pending = (None, k, slot)
            self._finish()
            return
        self._finish()        # the frame before this one: its gather has had this frame's render to hide behind
        self.pending = (work, k, slot)

    def drain(self):
        self._finish()
