"""The N > 1 path on the CPU: two processes over gloo each cut their interleaved
tiles out of a known frame (the layout rtx_render_tiles_device writes), gather
them to rank 0 with the same torch.distributed call the GPU path uses, and rank
0 reassembles the Screen-layout frame (the layout rtx_detile_device undoes).
"""
import os
import socket

import numpy as np
import pytest

from raytracer_amd.distributed import assemble_tiles, extract_tiles, tiles_per_rank


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, tile, out_path):
    import torch
    import torch.distributed as dist
    from raytracer_amd.distributed import gather_tiles
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    frame = rng.random((height, width, 3)).astype(np.float32)      # same frame on every rank
    mine = torch.from_numpy(extract_tiles(frame, tile, rank, world))
    stacked = gather_tiles(mine, dst=0)
    if rank == 0:
        rebuilt = assemble_tiles(stacked.numpy(), width, height, tile, world)
        np.save(out_path, np.array([np.array_equal(rebuilt, frame), stacked.shape[0] == world]))
    else:
        assert stacked is None
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, width, height, tile, out_path, slots=1):
    """Five different frames through the pipeline: every one must come out assembled, in
    order, although each gather overlaps the next render.  With `slots` frames in flight
    frame i must be rendered and assembled inside slot i % slots."""
    import contextlib
    import torch
    import torch.distributed as dist
    from raytracer_amd.distributed import FramePipeline
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    frames = [rng.random((height, width, 3)).astype(np.float32) for _ in range(5)]
    per = tiles_per_rank(width, height, tile, world)
    state = {"next": 0}
    rebuilt = []

    current, rendered_in, assembled_in = [None], [], []

    @contextlib.contextmanager
    def enter(slot):
        assert current[0] is None
        current[0] = slot
        yield
        current[0] = None

    def render(tiles):
        rendered_in.append(current[0])
        tiles.copy_(torch.from_numpy(extract_tiles(frames[state["next"]], tile, rank, world)))
        state["next"] += 1

    def assemble(stack):
        assembled_in.append(current[0])
        rebuilt.append(assemble_tiles(stack.numpy().copy(), width, height, tile, world))

    pipe = FramePipeline(lambda: torch.zeros((per, tile, tile, 3), dtype=torch.float32), render, assemble, slots=slots, enter=enter)
    for _ in frames:
        pipe.submit()
    pipe.drain()
    want_slots = [i % slots for i in range(len(frames))]
    assert rendered_in == want_slots, rendered_in
    if rank == 0:
        assert assembled_in == want_slots, assembled_in
        np.save(out_path, np.array([len(rebuilt) == len(frames)] + [np.array_equal(a, b) for a, b in zip(rebuilt, frames)]))
    else:
        assert not rebuilt
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("slots", [1, 2, 3])
def test_two_rank_frame_pipeline_assembles_every_frame_in_order(tmp_path, slots):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), 100, 70, 16, out, slots), nprocs=2, join=True)
    ok = np.load(out)
    assert len(ok) == 6 and ok.all()


@pytest.mark.parametrize("width,height,tile", [(100, 70, 16), (64, 64, 32), (33, 9, 8)])
def test_two_rank_tile_gather_rebuilds_the_frame(tmp_path, width, height, tile):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), width, height, tile, out), nprocs=2, join=True)
    assert np.load(out).all()


def test_tile_layout_round_trip_for_many_world_sizes():
    rng = np.random.default_rng(0)
    frame = rng.random((70, 100, 3)).astype(np.float32)
    for world in (1, 2, 3, 4, 8):
        for tile in (8, 16, 32):
            stack = np.stack([extract_tiles(frame, tile, r, world) for r in range(world)])
            assert stack.shape[1] == tiles_per_rank(100, 70, tile, world)
            assert np.array_equal(assemble_tiles(stack, 100, 70, tile, world), frame)
    # every tile belongs to exactly one rank
    owners = np.zeros((70, 100), np.int32)
    for r in range(3):
        marked = assemble_tiles(np.stack([np.full_like(extract_tiles(frame, 16, q, 3), 1.0 if q == r else 0.0) for q in range(3)]), 100, 70, 16, 3)
        owners += (marked[..., 0] == 1.0)
    assert (owners == 1).all()
