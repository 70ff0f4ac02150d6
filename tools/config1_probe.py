#!/usr/bin/env python3
"""BASELINE configs[0]: CornellBox-Mirror-Rotated 512x512, 1 spp, point light, Phong + hard
shadows + one mirror bounce (renderRayTracing, depth 1).  Kernel time, ray counts and the
call's wall time with a host framebuffer.  Needs a GPU and the committed scene fixture."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import cases
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import make_camera, make_params
    scene = cases.load_scene("cornell")
    f = abi.features(shading=1, hard_shadow=1, recursive=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
    t0 = time.perf_counter()
    gpu = engine.GpuEngine(scene, f)
    print(f"BvhInterface (build + flatten + upload): {(time.perf_counter() - t0) * 1e3:.1f} ms; {gpu.levels()} levels, {gpu.leaves()} leaves")
    cam, rp = make_camera(), make_params(512, 512)
    for rep in range(4):
        t0 = time.perf_counter()
        img = gpu.render(cam, f, rp)
        wall = (time.perf_counter() - t0) * 1e3
        s = gpu.stats
        print(f"frame {rep}: kernel {s.kernel_ms:.3f} ms, call {wall:.2f} ms (host framebuffer), rays {s.rays} = {s.rays / s.kernel_ms / 1e3:.0f} Mrays/s, sum {float(np.nansum(img)):.3f}")
    gpu.close()


if __name__ == "__main__":
    main()
