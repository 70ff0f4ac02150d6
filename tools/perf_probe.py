#!/usr/bin/env python3
"""Kernel time of the frame kernel under variations of the headline workload
(camera distance, samples per pixel, shading on/off), to see where a frame's
time goes.  Prints one line per variation.  Needs a GPU.

    python tools/perf_probe.py [--reps 5] [--only name,name]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch
    from raytracer_amd import abi, engine
    from raytracer_amd.scene import dragon_scene, make_camera, make_params, winding_dragon_scene

    hip = engine.hip_lib()
    scene = dragon_scene()
    base = dict(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
    variations = [
        ("headline", dict(), dict(), dict(samples=16)),
        ("no shadows", dict(hard_shadow=0), dict(), dict(samples=16)),
        ("no shading", dict(shading=0, hard_shadow=0), dict(), dict(samples=16)),
        ("looking away", dict(), dict(look_at=(0.0, 50.0, 0.0)), dict(samples=16)),
        ("close-up d=1.2", dict(), dict(distance=1.2), dict(samples=16)),
        ("close-up d=1.2 no shadows", dict(hard_shadow=0), dict(distance=1.2), dict(samples=16)),
        ("1 spp", dict(), dict(), dict(samples=1)),
        ("4 spp", dict(), dict(), dict(samples=4)),
        ("64 spp", dict(), dict(), dict(samples=64)),
        ("plain driver", dict(), dict(), dict(driver=abi.DRIVER_PLAIN)),
        ("plain driver close-up d=1.2", dict(), dict(distance=1.2), dict(driver=abi.DRIVER_PLAIN)),
        ("4 spp close-up d=1.2", dict(), dict(distance=1.2), dict(samples=4)),
        ("recursive (general path)", dict(recursive=1), dict(), dict(samples=16)),
    ]
    # configs[2]-like: one parallelogram light, soft shadows, glossy reflections (general kernel)
    area = [abi.parallelogram_light((-0.2, 0.9, 0.0), (0.4, 0, 0), (0, 0, 0.4), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1))]
    variations += [
        ("area light 64 samples", dict(soft_shadow=1, hard_shadow=0), dict(), dict(driver=abi.DRIVER_PLAIN, shadow_samples=64, ray_depth=0), area),
        ("area light 16 samples", dict(soft_shadow=1, hard_shadow=0), dict(), dict(driver=abi.DRIVER_PLAIN, shadow_samples=16, ray_depth=0), area),
        ("area light 32 samples", dict(soft_shadow=1, hard_shadow=0), dict(), dict(driver=abi.DRIVER_PLAIN, shadow_samples=32, ray_depth=0), area),
        ("area light 128 samples", dict(soft_shadow=1, hard_shadow=0), dict(), dict(driver=abi.DRIVER_PLAIN, shadow_samples=128, ray_depth=0), area),
        ("area light 64, close-up", dict(soft_shadow=1, hard_shadow=0), dict(distance=1.2), dict(driver=abi.DRIVER_PLAIN, shadow_samples=64, ray_depth=0), area),
        ("general path 1 spp", dict(recursive=1), dict(), dict(driver=abi.DRIVER_PLAIN, ray_depth=0), None),
        ("area 64 + glossy depth 2", dict(soft_shadow=1, hard_shadow=0, recursive=1, glossy_reflection=1), dict(), dict(driver=abi.DRIVER_PLAIN, shadow_samples=64, glossy_rays=5, ray_depth=2), area),
        ("area 64 + glossy depth 4", dict(soft_shadow=1, hard_shadow=0, recursive=1, glossy_reflection=1), dict(), dict(driver=abi.DRIVER_PLAIN, shadow_samples=64, glossy_rays=5, ray_depth=4), area),
        ("mirror depth 4", dict(recursive=1), dict(), dict(driver=abi.DRIVER_PLAIN, ray_depth=4), None),
        ("composed 2x2x3", dict(), dict(), dict(driver=abi.DRIVER_COMPOSED, samples=2, lens_samples=2, time_steps=3, prev_camera=engine.device_camera(make_camera(rotation_degrees=(22, 18, 0)), 1920, 1080)), "glass"),
        ("composed 2x2x3 + transparency", dict(), dict(), dict(driver=abi.DRIVER_COMPOSED, samples=2, lens_samples=2, time_steps=3, transparency=1, prev_camera=engine.device_camera(make_camera(rotation_degrees=(22, 18, 0)), 1920, 1080)), "glass"),
        ("composed 2x2x3 + transparency on the general path", dict(), dict(), dict(driver=abi.DRIVER_COMPOSED, samples=2, lens_samples=2, time_steps=3, transparency=1, prev_camera=engine.device_camera(make_camera(rotation_degrees=(22, 18, 0)), 1920, 1080)), "glass", "RTX_LEAN_LAYERS_FROM"),
        ("jittered 16 + transparency", dict(), dict(), dict(driver=abi.DRIVER_COMPOSED, samples=16, transparency=1), "glass"),
        ("jittered 16 + transparency on the general path", dict(), dict(), dict(driver=abi.DRIVER_COMPOSED, samples=16, transparency=1), "glass", "RTX_LEAN_LAYERS_FROM"),
        ("mirror ks 0.5 depth 1", dict(recursive=1), dict(), dict(samples=16, ray_depth=1), "shiny"),
        ("mirror ks 0.5 depth 1 on the general path", dict(recursive=1), dict(), dict(samples=16, ray_depth=1), "shiny", "RTX_NO_LEAN_MIRROR"),
        ("transparency driver", dict(), dict(), dict(driver=abi.DRIVER_TRANSPARENCY), None),
        ("dof 16 samples", dict(), dict(), dict(driver=abi.DRIVER_DOF, samples=16), None),
        ("motion blur 7 steps", dict(), dict(), dict(driver=abi.DRIVER_MOTION_BLUR, samples=7, prev_camera=engine.device_camera(make_camera(rotation_degrees=(22, 18, 0)), 1920, 1080)), None),
    ]
    # the harder stand-in mesh (coiled body, thin wings, legs, spikes), host-built and device-built tree
    variations += [
        ("winding dragon", dict(), dict(), dict(samples=16), "winding"),
        ("winding dragon close-up", dict(), dict(distance=1.4, rotation_degrees=(-30, 140, 0)), dict(samples=16), "winding"),
        ("winding dragon on GPU-built tree", dict(), dict(), dict(samples=16), "winding_dev"),
        ("winding dragon close-up on GPU-built tree", dict(), dict(distance=1.4, rotation_degrees=(-30, 140, 0)), dict(samples=16), "winding_dev"),
    ]
    only = [s for s in args.only.split(",") if s]
    dev = torch.device("cuda", 0)
    frame = torch.zeros((1080, 1920, 3), dtype=torch.float32, device=dev)
    gpu = engine.GpuEngine(scene, abi.features(**base))
    area_scene = dragon_scene()
    area_scene.lights = area
    for m in area_scene.meshes:   # a glossy material: ks != 0, shininess 50
        m.material[3:6] = 0.5
        m.material[6] = 50.0
    gpu_area = None
    others = {}
    for var in variations:
        name, fchg, cchg, pchg = var[:4]
        if only and name not in only:
            continue
        use = gpu
        for k in ("RTX_NO_LEAN_MIRROR", "RTX_LEAN_LAYERS_FROM"):
            os.environ.pop(k, None)
        if len(var) > 5:
            os.environ[var[5]] = "1000"
        if len(var) > 4 and var[4] == "glass":
            if "glass" not in others:
                glass = dragon_scene()
                for m in glass.meshes:
                    m.material[7] = 0.5     # translucent: camera rays go on through it (src/transparency.cpp:6-20)
                others["glass"] = engine.GpuEngine(glass, abi.features(**base))
            use = others["glass"]
        elif len(var) > 4 and var[4] == "shiny":
            if "shiny" not in others:
                shiny = dragon_scene()
                for m in shiny.meshes:
                    m.material[3:6] = 0.5
                others["shiny"] = engine.GpuEngine(shiny, abi.features(**base))
            use = others["shiny"]
        elif len(var) > 4 and isinstance(var[4], str):
            if var[4] not in others:
                others[var[4]] = engine.GpuEngine(winding_dragon_scene(), abi.features(**base), device_build=var[4].endswith("_dev"))
            use = others[var[4]]
        elif len(var) > 4 and var[4] is not None:
            if gpu_area is None:
                gpu_area = engine.GpuEngine(area_scene, abi.features(**base))
            use = gpu_area
        f = abi.features(**dict(base, **fchg))
        cam = engine.device_camera(make_camera(**cchg), 1920, 1080)
        rp = make_params(1920, 1080, **dict(dict(driver=abi.DRIVER_MRAYS), **pchg))
        st = abi.Stats()
        best = 1e9
        for _ in range(args.reps):
            rc = hip.rtx_render_device(use.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st))
            assert rc == 0, hip.rtx_last_error()
            best = min(best, st.kernel_ms)
        print(f"{name:28s} {best:8.3f} ms  rays {st.rays/1e6:8.2f} M  shaded {st.shaded_hits/1e6:6.2f} M  {st.rays/best/1e3:9.1f} Mrays/s", flush=True)
    gpu.close()
    for e in others.values():
        e.close()
    if gpu_area is not None:
        gpu_area.close()


if __name__ == "__main__":
    main()
