#!/usr/bin/env python3
"""BASELINE configs[4] shape on one GPU: ~10 M triangles (instanced stand-in), 3840x2160,
the depth-of-field, motion-blur and transparency drivers.  Prints build / upload / kernel
times, and checks that two renders of a frame are identical.  Needs a GPU.

    python tools/big_scene_probe.py [--copies 122]
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
    ap.add_argument("--copies", type=int, default=122)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--device-build", action="store_true", help="the tree built on the GPU (rtx_scene_build_device)")
    args = ap.parse_args()
    import torch
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import instanced_scene, make_camera, make_params
    hip = engine.hip_lib()
    t0 = time.time()
    scene = instanced_scene(args.copies)
    print(f"scene: {scene.n_triangles} triangles in {len(scene.meshes)} meshes, made in {time.time() - t0:.1f} s", flush=True)
    f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
    t0 = time.time()
    gpu = engine.GpuEngine(scene, f, device_build=args.device_build)
    print(f"host BVH build + flatten + upload: {time.time() - t0:.1f} s; {gpu.levels()} levels, {gpu.leaves()} leaves", flush=True)
    W, H = args.width, args.height
    dev = torch.device("cuda", 0)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    cam = engine.device_camera(make_camera(), W, H)
    prev = engine.device_camera(make_camera(rotation_degrees=(22, 18, 0)), W, H)
    runs = [
        ("plain 1 spp", make_params(W, H)),
        ("16 spp jittered", make_params(W, H, driver=abi.DRIVER_MRAYS, samples=16)),
        ("depth of field, 4 lens samples", make_params(W, H, driver=abi.DRIVER_DOF, samples=4, aperture=0.1, focal_length=2.0)),
        ("motion blur, 7 time steps", make_params(W, H, driver=abi.DRIVER_MOTION_BLUR, samples=7, prev_camera=prev)),
        ("transparency", make_params(W, H, driver=abi.DRIVER_TRANSPARENCY)),
        # configs[4]'s combination: 4 lens samples x 7 time steps x 2 jittered positions, the transparency chain on every ray
        ("composed 4 x 7 x 2 + transparency", make_params(W, H, driver=abi.DRIVER_COMPOSED, samples=2, lens_samples=4, time_steps=7, transparency=1, aperture=0.1, focal_length=2.0, prev_camera=prev)),
        ("composed 4 x 7 x 2", make_params(W, H, driver=abi.DRIVER_COMPOSED, samples=2, lens_samples=4, time_steps=7, aperture=0.1, focal_length=2.0, prev_camera=prev)),
    ]
    for name, rp in runs:
        st = abi.Stats()
        best, first = 1e9, None
        for rep in range(3):
            rc = hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st))
            assert rc == 0, hip.rtx_last_error()
            best = min(best, st.kernel_ms)
            if first is None:
                first = frame.clone()
            else:
                assert torch.equal(first, frame), "two renders of the same frame differ"
        lit = float((first.sum(dim=2) > 0).float().mean())
        print(f"{name:32s} {best:9.3f} ms  rays {st.rays/1e6:8.2f} M  {st.rays/best/1e3:9.1f} Mrays/s  ({lit*100:4.1f} % of pixels lit)", flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
