#!/usr/bin/env python3
"""Kernel time of the batch intersect entry point (rtx_trace_device) on the primary
rays of the headline view (one ray per pixel, tiled to the requested count) and on
the same rays shuffled (incoherent).  Needs a GPU."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import dragon_scene, make_camera
    hip = engine.hip_lib()
    f = abi.features(accel_structure=1, bvh_sah_binning=1, normal_interp=1)
    gpu = engine.GpuEngine(dragon_scene(), f)
    dev = torch.device("cuda", 0)
    for tag, cam in (("headline view", make_camera()), ("close-up d=1.2", make_camera(distance=1.2))):
        rays = engine.camera_rays(cam, 1920, 1080)
        # pixel order -> 8x8 blocks, as a renderer would hand them over
        idx = np.arange(1920 * 1080).reshape(1080 // 8, 8, 1920 // 8, 8).transpose(0, 2, 1, 3).ravel()
        for order, r in (("blocked", rays[idx]), ("shuffled", rays[np.random.default_rng(1).permutation(len(rays))])):
            n = len(r)
            d_rays = torch.from_numpy(r.view(np.uint8).reshape(n, -1).copy()).to(dev)
            d_hits = torch.zeros((n, abi.HIT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
            st = abi.Stats()
            best = 1e9
            for _ in range(5):
                rc = hip.rtx_trace_device(gpu.device_scene(), C.c_void_p(d_rays.data_ptr()), C.c_size_t(n), C.c_void_p(d_hits.data_ptr()), C.byref(f), C.byref(st))
                assert rc == 0, hip.rtx_last_error()
                best = min(best, st.kernel_ms)
            hits = int((d_hits.cpu().numpy().view(abi.HIT_DTYPE)["hit"] != 0).sum())
            print(f"{tag:16s} {order:9s} {n/1e6:6.2f} M rays  {best:7.3f} ms  {n/best/1e3:9.1f} Mrays/s  ({hits/n*100:4.1f} % hit)", flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
