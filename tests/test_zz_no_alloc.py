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
    # (rtx_render_device renders such frames in three passes, this entry point in one kernel: the same frames
    # for another driver and for one rank's tiles of a partitioned frame, whose other pixels stay untouched)
    for w, h, kw in ((56, 40, dict(driver=abi.DRIVER_MRAYS, samples=3, shadow_samples=70, glossy_rays=3, ray_depth=1)),
                     (72, 48, dict(shadow_samples=8, glossy_rays=2, ray_depth=2, tile_size=16, tile_rank=1, tile_world=3)),
                     (48, 48, dict(driver=abi.DRIVER_DOF, samples=4, shadow_samples=20, ray_depth=0, tile_size=8, tile_rank=0, tile_world=2))):
        cam = engine.device_camera(make_camera(), w, h)
        rp = make_params(w, h, **kw)
        (a, ra), = _frames(hip, static, g2.device_scene(), cam, ff, rp)
        (b, rb), = _frames(hip, hip.rtx_render_device, g2.device_scene(), cam, ff, rp)
        assert ra == rb, (w, h, ra, rb)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b)), f"{w}x{h}: frames differ"
        assert (a == -5.0).any() == ("tile_world" in kw)
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


def test_static_entry_point_under_stream_capture_and_without_allocation(gpu_device):
    """What the entry point is for.  (1) Frames of shapes the workspace has not seen (their sampler
    tables are made on the way, by a kernel on the stream) leave the GPU's free memory where it
    was.  (2) A call with stats == NULL is captured into a HIP graph -- a capture in which any
    allocation, copy from pageable memory or wait would fail -- and the graph's replays give
    rtx_render_device's frame; with a stats pointer the call is refused while capturing."""
    import torch
    hip = engine.hip_lib()
    static = hip.rtx_render_device_static
    static.argtypes = hip.rtx_render_device.argtypes
    scene = cases.load_scene("monkey")
    f = abi.features(**cases.BASE)
    gpu = GpuEngine(scene, f)
    dsc = gpu.device_scene()
    torch.cuda.synchronize()
    out = torch.zeros((90, 130, 3), dtype=torch.float32, device="cuda:0")
    free0 = torch.cuda.mem_get_info()[0]
    for w, h, kw in ((130, 90, dict(driver=abi.DRIVER_MRAYS, samples=3)), (75, 41, dict(driver=abi.DRIVER_DOF, samples=6)),
                     (64, 64, dict(driver=abi.DRIVER_MOTION_BLUR, samples=5, prev_camera=engine.device_camera(make_camera(rotation_degrees=(21, 19, 0)), 64, 64)))):
        cam = engine.device_camera(make_camera(), w, h)
        rp = make_params(w, h, **kw)
        assert static(dsc, C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(out.data_ptr()), None) == 0, hip.rtx_last_error()
    assert hip.rtx_synchronize() == 0
    assert torch.cuda.mem_get_info()[0] == free0, "the static entry point changed the GPU's free memory"

    w, h = 112, 80
    cam = engine.device_camera(make_camera(), w, h)
    rp = make_params(w, h, driver=abi.DRIVER_MRAYS, samples=7)   # (a shape new to the workspace: the table kernel is part of the capture)
    (want, _), = _frames(hip, hip.rtx_render_device, dsc, cam, f, rp)
    frame = torch.full((h, w, 3), -5.0, dtype=torch.float32, device="cuda:0")
    side = torch.cuda.Stream()
    assert hip.rtx_set_stream(C.c_void_p(side.cuda_stream)) == 0
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            st = abi.Stats()
            refused = static(dsc, C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st))
            message = hip.rtx_last_error()
            rc = static(dsc, C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), None)
        assert refused == -1 and b"captured" in message
        assert rc == 0, hip.rtx_last_error()
        for _ in range(3):   # (the second and third replay run in the order measured by the one before)
            frame.fill_(-5.0)
            graph.replay()
            torch.cuda.synchronize()
            got = frame.cpu().numpy()
            assert np.array_equal(got, want), "a replayed graph's frame differs from rtx_render_device's"
    finally:
        assert hip.rtx_set_stream(None) == 0
    gpu.close()
