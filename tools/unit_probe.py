# per-unit loop trips, start and duration of one rank's share of the headline
# frame (needs a diagnostic library build: hipcc $(HIPFLAGS) -DRTX_PROBE -shared -o tools/_probe.so raytracer_amd/csrc/rtx_hip.hip;
# the script puts it in place of librtx_hip.so in the tree it runs in, so run it on the GPU box's scratch copy only)
import ctypes as C, os, sys, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "tools", "_probe.so"), os.path.join(ROOT, "raytracer_amd", "librtx_hip.so"))
import torch
from raytracer_amd import abi, engine
from raytracer_amd.scene import dragon_scene, make_camera, make_params
hip = engine.hip_lib()
f = abi.features(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
gpu = engine.GpuEngine(dragon_scene(), f)
cam = engine.device_camera(make_camera(), 1920, 1080)
SPP = int(os.environ.get("RTX_PROBE_SPP", "16"))   # (samples per pixel of the probed frame)
for world in (1, 8):
    rp = make_params(1920, 1080, driver=abi.DRIVER_MRAYS, samples=SPP, tile_size=32, tile_rank=0, tile_world=world)
    per = hip.rtx_tiles_per_rank(C.byref(rp))
    tiles = torch.zeros((per, 32, 32, 3), dtype=torch.float32, device="cuda")
    st = abi.Stats()
    for _ in range(4):
        assert hip.rtx_render_tiles_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(tiles.data_ptr()), C.byref(st)) == 0
    n = per * 32 * 32 // max(1, 64 // min(SPP, 64))
    t = np.zeros((n, 4), np.uint32)
    assert hip.rtx_debug_probe(t.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == 0
    t = t[t[:, 0] > 0]    # units that ran (culled units do not report)
    # entries of positions the last frame did not run are left from earlier frames: keep the last frame's
    raw = np.sort(t[:, 3].astype(np.int64))
    gaps = np.nonzero(np.diff(raw) > 30000)[0]
    if len(gaps):
        t = t[t[:, 3].astype(np.int64) > raw[gaps[-1]]]
    us = t[:, 0] / 100.0
    lo_, li_, wo_, wi_ = t[:, 1] & 0xffff, t[:, 1] >> 16, t[:, 2] & 0xffff, t[:, 2] >> 16
    b = (t[:, 3] - t[:, 3].min()).astype(np.uint32) / 100.0
    e = b + us
    print(f"world {world}: kernel_ms {st.kernel_ms:.3f}, units that ran {len(t)}, span of unit activity {e.max():.1f} us")
    top = np.argsort(-us)[:8]
    for u in top:
        print(f"  longest: {us[u]:7.1f} us from {b[u]:6.1f}  lane {lo_[u]:5d}+{li_[u]:5d} trips; WAVE {wo_[u]:5d}+{wi_[u]:5d} -> {us[u]*1e3/max(1,wo_[u]+wi_[u]):6.0f} ns per wave trip")
    last = np.argsort(-e)[:8]
    for u in last:
        print(f"  last to end: ends {e[u]:6.1f}, began {b[u]:6.1f}, ran {us[u]:6.1f} us, WAVE {wo_[u]}+{wi_[u]} trips")
    edges = np.linspace(0, e.max(), 17)
    for i in range(16):
        lo, hi = edges[i], edges[i + 1]
        fl = (b < hi) & (e > lo)
        # wave-time inside this window
        busy = (np.minimum(e[fl], hi) - np.maximum(b[fl], lo)).sum()
        print(f"  t {lo:6.0f}-{hi:6.0f} us: started {((b >= lo) & (b < hi)).sum():6d}, in flight {fl.sum():6d}, busy waves (mean) {busy / (hi - lo):7.1f}")
gpu.close()
