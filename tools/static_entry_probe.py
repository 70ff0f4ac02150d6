#!/usr/bin/env python3
"""The headline frame through rtx_render_device and through rtx_render_device_static (the entry point that allocates
nothing): kernel time of each.  Needs a GPU."""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from raytracer_amd import abi, engine
from raytracer_amd.scene import dragon_scene, make_camera, make_params
hip = engine.hip_lib()
static = hip.rtx_render_device_static
static.argtypes = hip.rtx_render_device.argtypes
f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
gpu = engine.GpuEngine(dragon_scene(), f)
W, H = 1920, 1080
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
cam = engine.device_camera(make_camera(), W, H)
rp = make_params(W, H, driver=abi.DRIVER_MRAYS, samples=16)
for name, fn in (("rtx_render_device", hip.rtx_render_device), ("rtx_render_device_static", static)):
    st = abi.Stats(); best = 1e9
    for _ in range(8):
        assert fn(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st)) == 0
        best = min(best, st.kernel_ms)
    print(name, round(best, 3), "ms")
gpu.close()
