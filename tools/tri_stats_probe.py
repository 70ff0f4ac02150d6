#!/usr/bin/env python3
"""The wave-shared walk's triangle tests on the headline frame and on the golden scenes: how many pretests and whole
tests there are, how many have a lane that passes the plane stage, how many have a lane that hits, and how many lanes a
pretest dropped whose whole test would have hit (must be 0).  Needs a probe build (tools/dev_build.sh exp/probe.so -DRTX_PROBE);
run on the GPU box's scratch copy only."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, os.environ.get("RTX_PROBE_SO", "exp/probe.so")), os.path.join(ROOT, "raytracer_amd", "librtx_hip.so"))
import torch
from raytracer_amd import abi, engine
from raytracer_amd.scene import dragon_scene, make_camera, make_params
hip = engine.hip_lib()
from tests import cases
f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
out = (C.c_ulonglong * 8)()


def run(name, gpu, cam_kw, W, H, **kw):
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    cam = engine.device_camera(make_camera(**cam_kw), W, H)
    rp = make_params(W, H, **kw)
    st = abi.Stats()
    for _ in range(3):
        hip.rtx_debug_wave_walk_stats(out, 1)
        assert hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st)) == 0
    hip.rtx_debug_wave_walk_stats(out, 1)
    t, p, h, n, pre, wrong = out[0], out[1], out[2], out[3], out[4], out[5]
    print(f"{name}: {st.kernel_ms:.3f} ms; node visits {n/1e6:.2f} M; pretests {pre/1e6:.2f} M, whole tests {t/1e6:.2f} M, plane stage passed by some lane {p/1e6:.2f} M, "
          f"hit by some lane {h/1e6:.2f} M; dropped lanes that hit: {wrong}", flush=True)
    return wrong


bad = 0
gpu = engine.GpuEngine(dragon_scene(), f)
for name, cam_kw in (("headline", {}), ("close-up d=1.2", dict(distance=1.2)), ("inside d=0.3", dict(distance=0.3)), ("far d=12", dict(distance=12.0))):
    bad += run(name, gpu, cam_kw, 1920, 1080, driver=abi.DRIVER_MRAYS, samples=16 if name == "headline" else 4)
gpu.close()
for scene in ("cornell", "monkey", "teapot", "cube", "transparency", "custom"):
    gpu = engine.GpuEngine(cases.load_scene(scene), f)
    for cam_kw in ({}, dict(distance=0.7, rotation_degrees=(10, 200, 0)), dict(distance=6.0, rotation_degrees=(80, 45, 0)), dict(distance=0.05, look_at=(0.3, 0.2, 0.1))):
        bad += run(f"{scene} {cam_kw}", gpu, cam_kw, 512, 512, driver=abi.DRIVER_MRAYS, samples=4)
    gpu.close()
# triangle soups (slivers, zero-area triangles, lights and cameras inside), also shrunk and blown up
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fz
for seed in range(int(os.environ.get("RTX_PROBE_SOUPS", "40"))):
    for scale in (1.0, 0.004, 3000.0):
        scene, rng = fz.random_scene(seed)
        for m in scene.meshes:
            m.vertices[:, 0:3] *= np.float32(scale)
        scene.lights = [abi.point_light(tuple(float(np.float32(v) * np.float32(scale)) for v in l.p[0]), tuple(l.c[0])) for l in scene.lights]
        scene.spheres = []
        try:
            gpu = engine.GpuEngine(scene, f)
        except Exception as e:      # (a development build takes trees of at most 16 levels)
            print(f"soup {seed} x {scale}: {e}")
            continue
        for dist in (None, 0.15, 0.02):
            cam = fz.random_camera(rng, dist)
            kw = dict(fov_degrees=float(np.degrees(cam.fovy)), distance=float(cam.distance) * scale, look_at=tuple(float(v) * scale for v in cam.look_at),
                      rotation_degrees=tuple(float(np.degrees(v)) for v in cam.euler))
            bad += run(f"soup {seed} x {scale} d={dist}", gpu, kw, 256, 144, driver=abi.DRIVER_MRAYS, samples=4)
        gpu.close()
print("dropped lanes that hit, in all:", bad)
sys.exit(1 if bad else 0)
