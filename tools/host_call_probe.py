#!/usr/bin/env python3
"""Wall time of a frame through the host-buffer entry point (rtx_render: the framebuffer is
the caller's host memory, as with the reference's Screen) next to the kernel time, for the
headline frame.  This is the PCIe-inclusive figure; bench.py reports the device-resident one."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import dragon_scene, make_camera, make_params
    f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
    gpu = engine.GpuEngine(dragon_scene(), f)
    cam = make_camera()
    for W, H in ((1920, 1080), (3840, 2160), (512, 512)):
        rp = make_params(W, H, driver=abi.DRIVER_MRAYS, samples=16)
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            gpu.render(cam, f, rp)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(f"{W}x{H} 16 spp: kernel {gpu.stats.kernel_ms:.3f} ms, rtx_render total {gpu.stats.total_ms:.3f} ms, python call {best:.3f} ms ({W * H * 12 / 1e6:.1f} MB framebuffer)", flush=True)
    # the C ABI directly: the caller's buffer as it is, then page-locked with rtx_host_register
    import ctypes as C
    import numpy as np
    hip = engine.hip_lib()
    hip.rtx_host_alloc.restype = C.c_void_p
    hip.rtx_host_alloc.argtypes = [C.c_size_t]
    hip.rtx_host_free.argtypes = [C.c_void_p]
    for W, H in ((1920, 1080), (3840, 2160)):
        rp = make_params(W, H, driver=abi.DRIVER_MRAYS, samples=16)
        dcam = engine.device_camera(cam, W, H)
        plain = np.zeros((H, W, 3), np.float32)
        locked = hip.rtx_host_alloc(C.c_size_t(plain.nbytes))
        assert locked, hip.rtx_last_error()
        for tag, ptr in (("pageable", plain.ctypes.data), ("page-locked", locked)):
            best = 1e9
            for _ in range(6):
                t0 = time.perf_counter()
                assert hip.rtx_render(gpu.device_scene(), C.byref(dcam), C.byref(f), C.byref(rp), C.c_void_p(ptr), None) == 0
                best = min(best, (time.perf_counter() - t0) * 1e3)
            print(f"{W}x{H} rtx_render into a {tag} buffer: {best:.3f} ms per call", flush=True)
        same = np.array_equal(plain, np.ctypeslib.as_array((C.c_float * (W * H * 3)).from_address(locked)).reshape(H, W, 3))
        print("  frames identical:", same)
        hip.rtx_host_free(C.c_void_p(locked))
    gpu.close()


if __name__ == "__main__":
    main()
