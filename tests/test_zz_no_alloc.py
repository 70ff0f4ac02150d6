"""rtx_render_device_static: the frame driver for callers that cannot allocate device
memory.  Its workspace is a fixed array inside the library instead of the buffers
rtx_init allocates; the frames must be rtx_render_device's, bit for bit (the bar for
frames is 1e-3 absolute; the two entry points share their kernels, so nothing less
than identity is expected), on calls in a row that share the array, the second smaller.
"""
import ctypes as C

import numpy as np
import pytest

import cases
from raytracer_amd import abi, engine
from raytracer_amd.engine import GpuEngine
from raytracer_amd.scene import make_camera, make_params

pytestmark = pytest.mark.gpu


def _frames(hip, entry, dsc, cam, f, rp, n=1):
    import torch
    out = []
    for _ in range(n):
        frame = torch.full((rp.height, rp.width, 3), -5.0, dtype=torch.float32, device="cuda:0")
        st = abi.Stats()
        rc = entry(dsc, C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st))
        assert rc == 0, hip.rtx_last_error()
        out.append((frame.cpu().numpy(), (st.rays_primary, st.rays_shadow, st.rays_bounce, st.shaded_hits)))
    return out


def test_static_workspace_entry_point_matches_render_device(gpu_device):
    hip = engine.hip_lib()
    static = hip.rtx_render_device_static
    static.argtypes = hip.rtx_render_device.argtypes
    scene = cases.load_scene("teapot")
    f = abi.features(**cases.BASE)
    gpu = GpuEngine(scene, f)
    dsc = gpu.device_scene()
    shapes = [  # (width, height, params): a larger frame first, then smaller ones on the same array
        (200, 120, dict(driver=abi.DRIVER_MRAYS, samples=4)),
        (96, 64, dict(driver=abi.DRIVER_MRAYS, samples=4)),
        (37, 21, dict()),
        (64, 48, dict(driver=abi.DRIVER_DOF, samples=5)),
    ]
    for w, h, kw in shapes:
        cam = engine.device_camera(make_camera(), w, h)
        rp = make_params(w, h, **kw)
        # three frames each: the first runs in list order, the next ones in the order
        # measured on the frame before (kept in the workspace from call to call)
        want = _frames(hip, hip.rtx_render_device, dsc, cam, f, rp, 3)
        got = _frames(hip, static, dsc, cam, f, rp, 3)
        for (a, ra), (b, rb) in zip(got, want):
            assert ra == rb, (w, h, ra, rb)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            assert np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max() <= 1e-3
            assert np.array_equal(np.nan_to_num(a), np.nan_to_num(b)), f"{w}x{h}: frames differ"
            assert a.min() >= 0.0   # every pixel written
    # the general path (area light, glossy reflection) through the same workspace
    area = cases.load_scene("cornell")
    area.lights = [abi.point_light((0.0, 0.5, 0.3), (0.3, 0.3, 0.3))] + list(cases._AREA)
    ff = abi.features(**dict(cases.BASE, soft_shadow=1, recursive=1, glossy_reflection=1))
    g2 = GpuEngine(area, ff)
    cam = engine.device_camera(make_camera(), 40, 30)
    rp = make_params(40, 30, shadow_samples=16, glossy_rays=2, ray_depth=2)
    (a, ra), = _frames(hip, static, g2.device_scene(), cam, ff, rp)
    (b, rb), = _frames(hip, hip.rtx_render_device, g2.device_scene(), cam, ff, rp)
    assert ra == rb and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))
    g2.close()
    # beyond the fixed workspace: refused with a message, nothing launched
    import torch
    big = make_params(4104, 8)
    small_out = torch.zeros(8, dtype=torch.float32, device="cuda:0")
    cam = engine.device_camera(make_camera(), 4104, 8)
    rc = static(dsc, C.byref(cam), C.byref(f), C.byref(big), C.c_void_p(small_out.data_ptr()), None)
    assert rc == -3 and b"fixed workspace" in hip.rtx_last_error()
    many = make_params(4096, 4096, tile_size=1024)   # fits
    many_blocks = make_params(4096, 4090, tile_size=1024)
    assert hip.rtx_tiles_per_rank(C.byref(many)) == 16 and hip.rtx_tiles_per_rank(C.byref(many_blocks)) == 16
    gpu.close()
