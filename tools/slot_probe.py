#!/usr/bin/env python3
"""Which pairs of HIP streams let two frames share the GPU: one rank's share of an 8-rank
headline frame (a frame that ends on a few long work units) queued alternately in frame
slots 0 and 1, with slot 1 on each of several streams in turn.  Prints the time per frame
for every pairing.  Needs a GPU.

    python tools/slot_probe.py [--streams 8] [--steps 40]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--world", type=int, default=8)
    args = ap.parse_args()
    import torch
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import dragon_scene, make_camera, make_params

    hip = engine.hip_lib()
    W, H = 1920, 1080
    f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
    gpu = engine.GpuEngine(dragon_scene(), f)
    cam = engine.device_camera(make_camera(), W, H)
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    rp = make_params(W, H, driver=abi.DRIVER_MRAYS, samples=16, tile_size=32, tile_rank=4 % args.world, tile_world=args.world)
    per = hip.rtx_tiles_per_rank(C.byref(rp))
    tiles = [torch.zeros((per, 32, 32, 3), dtype=torch.float32, device=dev) for _ in range(2)]

    def run(pair):
        for k in (1, 0):
            assert hip.rtx_select_slot(k) == 0
            assert hip.rtx_set_stream(C.c_void_p(streams[pair[k]].cuda_stream)) == 0
        n = [0]

        def step():
            k = n[0] % 2 if pair[0] != pair[1] else 0
            n[0] += 1
            assert hip.rtx_select_slot(k) == 0
            assert hip.rtx_render_tiles_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(tiles[k].data_ptr()), None) == 0
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        e0, e1, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        s0, s1 = streams[pair[0]], streams[pair[1]]
        e0.record(s0)
        for _ in range(args.steps):
            step()
        e.record(s1)
        s0.wait_event(e)
        e1.record(s0)
        torch.cuda.synchronize()
        hip.rtx_select_slot(0)
        return e0.elapsed_time(e1) / args.steps * 1e3

    print(f"one stream: {run((0, 0)):7.1f} us/frame", flush=True)
    for j in range(1, args.streams):
        print(f"streams 0 and {j}: {run((0, j)):7.1f} us/frame", flush=True)
    for i in range(1, min(4, args.streams)):
        for j in range(i + 1, min(5, args.streams)):
            print(f"streams {i} and {j}: {run((i, j)):7.1f} us/frame", flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
