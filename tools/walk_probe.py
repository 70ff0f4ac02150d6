#!/usr/bin/env python3
"""Where the walk of a glossy frame spends a deep pixel's time: per work unit of the walk (one pixel at depth 4)
the ticks spent making rays, walking the tree, on hit attributes, on the shading points' numbers, on records
(shading points, exchange, tape) and on the state machine's advance.  Needs a probe build of the library
(tools/dev_build.sh exp/probe.so -DRTX_PROBE), named by RTX_PROBE_SO; run on the GPU box's scratch copy only
(the script puts that library in place of raytracer_amd/librtx_hip.so)."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, os.environ.get("RTX_PROBE_SO", "exp/probe.so")), os.path.join(ROOT, "raytracer_amd", "librtx_hip.so"))
import torch
from raytracer_amd import abi, engine
from raytracer_amd.scene import dragon_scene, make_camera, make_params
hip = engine.hip_lib()
scene = dragon_scene()
scene.lights = [abi.parallelogram_light((-0.2, 0.9, 0.0), (0.4, 0, 0), (0, 0, 0.4), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1))]
for m in scene.meshes:
    m.material[3:6] = 0.5
    m.material[6] = 50.0
f = abi.features(shading=1, soft_shadow=1, recursive=1, glossy_reflection=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
gpu = engine.GpuEngine(scene, f)
W, H = 1920, 1080
cam = engine.device_camera(make_camera(), W, H)
rp = make_params(W, H, driver=abi.DRIVER_PLAIN, shadow_samples=64, glossy_rays=5, ray_depth=int(os.environ.get("DEPTH", "4")))
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
st = abi.Stats()
for _ in range(4):
    assert hip.rtx_render_device(gpu.device_scene(), C.byref(cam), C.byref(f), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(st)) == 0, hip.rtx_last_error()
print(f"frame {st.kernel_ms:.2f} ms, rays {st.rays/1e6:.2f} M, shaded {st.shaded_hits/1e6:.3f} M")
n = 1 << 20
t = np.zeros((n, 12), np.uint32)
assert hip.rtx_debug_walk_probe(t.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == 0
names = ["make rays", "walk the tree", "hit attributes", "point numbers", "records", "advance"]
total = t[:, :6].sum(1).astype(np.int64)
ran = total > 0
print(f"units that walked: {ran.sum()}, wave time in the walk {total.sum() / 100e3:.1f} ms summed over waves")
for i, name in enumerate(names):
    print(f"  all units  {name:16s} {t[:, i].astype(np.int64).sum() / 100e3:9.2f} ms ({100.0 * t[:, i].astype(np.int64).sum() / max(1, total.sum()):5.1f} %)")
# units by how many fans they served: how many there are, what they take and where it goes
fans = t[:, 6].astype(np.int64)
for lo, hi in ((0, 0), (1, 1), (2, 2), (3, 6), (7, 31), (32, 10**9)):
    sel = ran & (fans >= lo) & (fans <= hi)
    if not sel.any():
        continue
    part = t[sel][:, :6].astype(np.int64).sum(0)
    print(f"units with {lo}..{min(hi, 999)} fans: {sel.sum():7d}, {total[sel].sum() / 100e3:8.1f} ms of wave time, {total[sel].mean() / 100.0:7.1f} us each, others/unit {t[sel][:, 7].mean():.1f}: "
          + ", ".join(f"{n} {100.0 * v / max(1, part.sum()):.0f}%" for n, v in zip(names, part)))
for u in np.argsort(-total)[:2]:
    fan, other = int(t[u, 6]), int(t[u, 7])
    print(f"unit {u}: {total[u] / 100.0:9.1f} us, {fan} fan iterations + {other} others -> {total[u] / 100.0 / max(1, fan):6.1f} us per fan; "
          f"frontier rounds {t[u, 8]} ({t[u, 8] / max(1, fan):.1f} per fan, {t[u, 1] / 100.0 / max(1, t[u, 8]):.2f} us each), live entries {t[u, 9]} ({t[u, 9] / max(1, fan):.1f} per fan)")
    for i, name in enumerate(names):
        print(f"    {name:16s} {t[u, i] / 100.0:9.1f} us ({100.0 * t[u, i] / total[u]:5.1f} %)   {t[u, i] / 100.0 / max(1, fan + other):6.2f} us per iteration")

# the walk's timeline: when its longest units began and ended (ticks of the 100 MHz clock since the first unit began)
gpu2 = None
g = np.zeros((n, 4), np.uint32)
assert hip.rtx_debug_probe(g.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == 0
sel = ran & (g[:, 0] > 0)
began = g[:, 3].astype(np.int64)
# (keep the last frame's entries: the clock's low word, units of one frame lie within a few milliseconds)
newest = began[sel].max()
sel &= (newest - began) % (1 << 32) < 3_000_000
b = ((began - began[sel].min()) % (1 << 32)) / 100.0
e = b + g[:, 0] / 100.0
print(f"walk: first unit to last end {e[sel].max():.0f} us; units {sel.sum()}")
for u in np.argsort(-np.where(sel, e, 0))[:6]:
    print(f"  last to end: unit {u} began {b[u]:8.1f} us, ran {g[u, 0] / 100.0:8.1f} us, ended {e[u]:8.1f} us, fans {t[u, 6]}")
for u in np.argsort(-np.where(sel, g[:, 0], 0))[:6]:
    print(f"  longest: unit {u} began {b[u]:8.1f} us, ran {g[u, 0] / 100.0:8.1f} us, ended {e[u]:8.1f} us, fans {t[u, 6]}")
edges = np.linspace(0, e[sel].max(), 13)
for i in range(12):
    lo, hi = edges[i], edges[i + 1]
    fl = sel & (b < hi) & (e > lo)
    busy = (np.minimum(e[fl], hi) - np.maximum(b[fl], lo)).sum()
    print(f"  t {lo:6.0f}-{hi:6.0f} us: units begun {(sel & (b >= lo) & (b < hi)).sum():7d}, busy waves (mean) {busy / (hi - lo):7.1f}")
gpu.close()
