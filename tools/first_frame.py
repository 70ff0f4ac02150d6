# experiment: time of the FIRST frame of a shape (nothing measured yet) and of the following ones
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import abi, engine
from raytracer_amd.scene import dragon_scene, make_camera, make_params
hip = engine.hip_lib()
f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
gpu = engine.GpuEngine(dragon_scene(), f)
for W, H, spp in ((1920, 1080, 16), (1280, 720, 16), (1920, 1080, 1), (3840, 2160, 4)):
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    cam = engine.device_camera(make_camera(), W, H)
    rp = make_params(W, H, driver=abi.DRIVER_MRAYS, samples=spp)
    ms = []
    for _ in range(4):
        st = abi.Stats()
        assert hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st)) == 0
        ms.append(st.kernel_ms)
    print(f"{W}x{H} {spp} spp: first frame {ms[0]:.3f} ms, then " + ", ".join(f"{m:.3f}" for m in ms[1:]), flush=True)
gpu.close()
