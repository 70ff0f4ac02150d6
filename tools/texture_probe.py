#!/usr/bin/env python3
"""BASELINE configs[3] shape on one GPU: textured geometry with nearest / bilinear /
trilinear filtering and an equirectangular environment map, 1920x1080, 1 spp.
Needs a GPU and the committed scene fixtures (tests/golden/scenes)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import cases
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import make_camera, make_params
    hip = engine.hip_lib()
    rng = np.random.default_rng(4)
    scene = cases.load_scene("cube_textured")
    # a 1024^2 seeded checker + noise texture instead of the 128^2 one
    yy, xx = np.mgrid[0:1024, 0:1024]
    tex = (((xx // 32 + yy // 32) % 2) * 0.6 + 0.2)[..., None] * np.ones(3) + rng.uniform(-0.1, 0.1, (1024, 1024, 3))
    for m in scene.meshes:
        if m.texture is not None:
            m.texture = np.ascontiguousarray(np.clip(tex, 0, 1), np.float32)
    env = (np.floor(rng.random((1024, 2048, 3)) * 256) / 255.0).astype(np.float32)
    scene.set_envmap(env, abi.ENV_SPHERICAL, 180.0, (0.5, 0.5, 0.5))
    base = dict(cases.BASE, texture_mapping=1, environment_mapping=1)
    dev = torch.device("cuda", 0)
    W, H = 1920, 1080
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    gpu = engine.GpuEngine(scene, abi.features(**base))
    cam = engine.device_camera(make_camera(distance=2.2), W, H)
    for name, extra, pchg in (("nearest", dict(), dict()),
                              ("bilinear", dict(bilinear_texture_filtering=1), dict()),
                              ("trilinear (mip chain)", dict(bilinear_texture_filtering=1, mipmap_texture_filtering=1), dict()),
                              ("trilinear + mirror depth 2", dict(bilinear_texture_filtering=1, mipmap_texture_filtering=1, recursive=1), dict(ray_depth=2)),
                              ("trilinear, 16 spp", dict(bilinear_texture_filtering=1, mipmap_texture_filtering=1), dict(driver=abi.DRIVER_MRAYS, samples=16))):
        f = abi.features(**dict(base, **extra))
        rp = make_params(W, H, **pchg)
        st = abi.Stats()
        best = 1e9
        for _ in range(4):
            rc = hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st))
            assert rc == 0, hip.rtx_last_error()
            best = min(best, st.kernel_ms)
        px = W * H * (rp.samples if rp.driver == abi.DRIVER_MRAYS else 1)
        print(f"{name:28s} {best:8.3f} ms  rays {st.rays/1e6:7.2f} M  {st.rays/best/1e3:8.1f} Mrays/s  {px/best/1e3:8.1f} Msamples/s", flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
