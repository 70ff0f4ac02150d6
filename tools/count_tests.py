import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
os.environ["RTX_COUNT_TESTS"] = "1"
import torch
from raytracer_amd import abi, engine
from raytracer_amd.scene import dragon_scene, make_camera, make_params
hip = engine.hip_lib()
f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
gpu = engine.GpuEngine(dragon_scene(), f)
frame = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
for name, ff, cam_kw in (("headline", f, {}), ("no shadows", abi.features(shading=1, hard_shadow=0, normal_interp=1, accel_structure=1, bvh_sah_binning=1), {}), ("close-up", f, dict(distance=1.2))):
    cam = engine.device_camera(make_camera(**cam_kw), 1920, 1080)
    rp = make_params(1920, 1080, driver=abi.DRIVER_MRAYS, samples=16)
    st = abi.Stats()
    for _ in range(3):
        assert hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(ff), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st)) == 0
    print(name, "kernel_ms %.3f" % st.kernel_ms, "rays", st.rays_primary, st.rays_shadow, "shaded", st.shaded_hits, "box tests", st.box_tests, "tri tests", st.tri_tests,
          "per shaded hit: box %.1f tri %.1f" % (st.box_tests / max(1, st.shaded_hits), st.tri_tests / max(1, st.shaded_hits)))
gpu.close()
