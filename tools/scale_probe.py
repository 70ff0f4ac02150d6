#!/usr/bin/env python3
"""What one rank of an N-rank frame costs, measured on ONE GPU: for N in 1,2,4,8 and
every rank r < N, the device time per frame of rtx_render_tiles_device (tile t ->
rank t % N), the host time to enqueue it, and the time of rtx_detile_device on
rank 0.  The gather between them needs N GPUs and is not measured here.

    python tools/scale_probe.py [--steps 20]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--workload", default="headline", choices=["headline", "winding", "area_glossy", "big_scene"], help="bench.py's workloads")
    ap.add_argument("--slots", type=int, default=1, help="frames in flight (rtx_select_slot): frame i is queued in slot i %% slots on that slot's stream")
    args = ap.parse_args()
    import torch
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import dragon_scene, make_camera, make_params

    hip = engine.hip_lib()
    import bench
    wargs = argparse.Namespace(workload=args.workload, width=1920, height=1080, spp=16, obj_npz=None, obj=None)
    if args.workload == "big_scene":
        wargs.width, wargs.height = 3840, 2160
    wl = bench.workload(args.workload, wargs)
    W, H = wargs.width, wargs.height
    f = wl["features"]
    gpu = engine.GpuEngine(wl["scene"], f, device_build=args.workload == "big_scene")
    cam = engine.device_camera(make_camera(), W, H)
    dev = torch.device("cuda", 0)
    streams = engine.concurrent_streams(args.slots, dev)
    args.slots = len(streams)
    for k in reversed(range(args.slots)):
        assert hip.rtx_select_slot(k) == 0
        assert hip.rtx_set_stream(C.c_void_p(streams[k].cuda_stream)) == 0
    stream = streams[0]
    torch.cuda.set_stream(stream)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    for world in [int(w) for w in args.worlds.split(",")]:
        worst = 0.0
        for rank in range(world):
            rp = make_params(W, H, tile_size=args.tile, tile_rank=rank, tile_world=world, **wl["params"])
            per = hip.rtx_tiles_per_rank(C.byref(rp))
            tiles = [torch.zeros((per, args.tile, args.tile, 3), dtype=torch.float32, device=dev) for _ in range(args.slots)]
            n = [0]

            def step():
                k = n[0] % args.slots
                n[0] += 1
                assert hip.rtx_select_slot(k) == 0
                rc = hip.rtx_render_tiles_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(tiles[k].data_ptr()), None)
                assert rc == 0, hip.rtx_last_error()
            for _ in range(3 * args.slots):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(args.steps):
                step()
            for k in range(1, args.slots):
                e = torch.cuda.Event()
                e.record(streams[k])
                stream.wait_event(e)
            e1.record(stream)
            hip.rtx_select_slot(0)
            host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dev_ms = e0.elapsed_time(e1) / args.steps
            worst = max(worst, dev_ms)
            print(f"world {world} rank {rank}: device {dev_ms*1e3:7.1f} us/frame, host enqueue {host/args.steps*1e6:6.1f} us/frame", flush=True)
        rp = make_params(W, H, tile_size=args.tile, tile_rank=0, tile_world=world, **wl["params"])
        per = hip.rtx_tiles_per_rank(C.byref(rp))
        allt = torch.zeros((world, per, args.tile, args.tile, 3), dtype=torch.float32, device=dev)
        for _ in range(3):
            hip.rtx_detile_device(C.c_void_p(allt.data_ptr()), C.byref(rp), C.c_void_p(frame.data_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            hip.rtx_detile_device(C.c_void_p(allt.data_ptr()), C.byref(rp), C.c_void_p(frame.data_ptr()))
        e1.record(stream)
        torch.cuda.synchronize()
        det = e0.elapsed_time(e1) / args.steps
        print(f"world {world}: slowest rank {worst*1e3:.1f} us, detile {det*1e3:.1f} us", flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
