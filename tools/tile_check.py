#!/usr/bin/env python3
"""Renders every rank's tiles of a bench.py workload on ONE GPU, assembles them and compares the
result with the frame rendered whole, bit for bit.  Needs a GPU.

    python tools/tile_check.py [--workload area_glossy] [--world 4]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="area_glossy")
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--repeat", type=int, default=4, help="times each rank's tiles are rendered (later frames run on grown pools and measured block costs)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    import torch
    import bench
    from raytracer_amd import engine
    from raytracer_amd.scene import make_camera, make_params
    hip = engine.hip_lib()
    wargs = argparse.Namespace(workload=args.workload, width=args.width, height=args.height, spp=16, obj_npz=None)
    wl = bench.workload(args.workload, wargs)
    W, H, f = args.width, args.height, wl["features"]
    gpu = engine.GpuEngine(wl["scene"], f)
    cam = engine.device_camera(make_camera(), W, H)
    dev = torch.device("cuda", 0)
    whole = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    again = torch.zeros_like(whole)
    one = make_params(W, H, tile_size=args.tile, **wl["params"])
    for out in (whole, again):
        assert hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(one), C.c_void_p(out.data_ptr()), None) == 0, hip.rtx_last_error()
    assert hip.rtx_synchronize() == 0
    print("two whole frames equal:", torch.equal(whole.view(torch.int32), again.view(torch.int32)), flush=True)
    rp0 = make_params(W, H, tile_size=args.tile, tile_rank=0, tile_world=args.world, **wl["params"])
    per = hip.rtx_tiles_per_rank(C.byref(rp0))
    stack = torch.zeros((args.world, per, args.tile, args.tile, 3), dtype=torch.float32, device=dev)
    for rep in range(args.repeat):
        for rank in range(args.world):
            rp = make_params(W, H, tile_size=args.tile, tile_rank=rank, tile_world=args.world, **wl["params"])
            assert hip.rtx_render_tiles_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(stack[rank].data_ptr()), None) == 0, hip.rtx_last_error()
        out = torch.zeros_like(whole)
        assert hip.rtx_detile_device(C.c_void_p(stack.data_ptr()), C.byref(rp0), C.c_void_p(out.data_ptr())) == 0
        assert hip.rtx_synchronize() == 0
        same = out.view(torch.int32) == whole.view(torch.int32)
        bad = (~same).any(dim=2)
        print(f"pass {rep}: assembled == whole: {bool(same.all())}; {int(bad.sum())} pixels differ", flush=True)
        if bad.any():
            ys, xs = torch.nonzero(bad, as_tuple=True)
            for y, x in list(zip(ys.tolist(), xs.tolist()))[:6]:
                print(f"  pixel ({x}, {y}) tile {(y // args.tile) * ((W + args.tile - 1) // args.tile) + x // args.tile}: whole {whole[y, x].tolist()} tiles {out[y, x].tolist()}")
    again2 = torch.zeros_like(whole)
    assert hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(one), C.c_void_p(again2.data_ptr()), None) == 0
    assert hip.rtx_synchronize() == 0
    print("whole frame after the tiles equals the first:", torch.equal(whole.view(torch.int32), again2.view(torch.int32)), flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
