"""Python driver of the native host layer (librtx_host.so -> librtx_hip.so).

GpuEngine mirrors the reference's object model: it owns a Scene assembled from
arrays and a BvhInterface built over it, and exposes BvhInterface::intersect
(batched) and the renderRayTracing* drivers.  Everything is computed by the HIP
library; there is no CPU fallback -- if the libraries or the GPU are missing the
constructor raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .scene import SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(_HERE, "librtx_hip.so")
# (RTX_HOST_SO: another build of the host library, e.g. the sanitizer build of `make asan`)
HOST_SO = os.environ.get("RTX_HOST_SO") or os.path.join(_HERE, "librtx_host.so")

vp = C.c_void_p
_host = None
_hip = None


class NativeLibraryMissing(RuntimeError):
    pass


def hip_lib():
    """librtx_hip.so: the C ABI of include/rtx.h."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise NativeLibraryMissing(f"{HIP_SO} not built (run `make` or __graft_entry__.build())")
        lib = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        lib.rtx_last_error.restype = C.c_char_p
        lib.rtx_scene_create.restype = vp
        lib.rtx_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        lib.rtx_scene_destroy.argtypes = [vp]
        lib.rtx_scene_retain.argtypes = [vp]
        lib.rtx_trace.argtypes = [vp, vp, C.c_size_t, vp, C.POINTER(abi.Features), C.POINTER(abi.Stats)]
        lib.rtx_trace_device.argtypes = lib.rtx_trace.argtypes
        lib.rtx_shade.argtypes = [vp, vp, C.c_size_t, C.POINTER(abi.Features), C.POINTER(abi.RenderParams), vp, C.POINTER(abi.Stats)]
        lib.rtx_shade_device.argtypes = lib.rtx_shade.argtypes
        for name in ("rtx_render", "rtx_render_device", "rtx_render_tiles_device"):
            getattr(lib, name).argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.Features), C.POINTER(abi.RenderParams), vp, C.POINTER(abi.Stats)]
        lib.rtx_detile_device.argtypes = [vp, C.POINTER(abi.RenderParams), vp]
        lib.rtx_tiles_per_rank.argtypes = [C.POINTER(abi.RenderParams)]
        lib.rtx_set_stream.argtypes = [vp]
        lib.rtx_select_slot.argtypes = [C.c_int]
        lib.rtx_streams_overlap.argtypes = [vp, vp]
        lib.rtx_selfcheck_trig.argtypes = [vp, vp, C.c_size_t, vp]
        _hip = lib
    return _hip


def concurrent_streams(n, device, tries=16):
    """`n` torch streams on `device` whose kernels run at the same time (no two on the same
    hardware queue, rtx_streams_overlap), for frames in flight in as many frame slots.  Fewer
    are returned when the runtime has no more queues to offer.  rtx_init must have been called."""
    import torch
    lib = hip_lib()
    picked, kept = [torch.cuda.Stream(device=device)], []
    while len(picked) < n and tries > 0:
        tries -= 1
        s = torch.cuda.Stream(device=device)
        verdicts = [lib.rtx_streams_overlap(vp(p.cuda_stream), vp(s.cuda_stream)) for p in picked]
        if any(v < 0 for v in verdicts):
            raise RuntimeError(lib.rtx_last_error().decode())
        (picked if all(v == 1 for v in verdicts) else kept).append(s)
    return picked


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_SO):
            raise NativeLibraryMissing(f"{HOST_SO} not built (run `make` or __graft_entry__.build())")
        lib = C.CDLL(HOST_SO)
        lib.rtxh_last_error.restype = C.c_char_p
        for name in ("rtxh_scene_create", "rtxh_flatten", "rtxh_bvh_create", "rtxh_bvh_flat", "rtxh_bvh_device_scene"):
            getattr(lib, name).restype = vp
        _host = lib
    return _host


def _ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {host_lib().rtxh_last_error().decode()}")


class HostScene:
    """A Scene (src/scene.h:31-37) held by the native host layer."""

    def __init__(self, scene: SceneData):
        lib = host_lib()
        self.h = lib.rtxh_scene_create()
        for m in scene.meshes:
            tex = m.texture
            _check(lib.rtxh_scene_add_mesh(vp(self.h), _ptr(m.vertices), len(m.vertices), _ptr(m.triangles), len(m.triangles), _ptr(m.material),
                                           _ptr(tex), 0 if tex is None else tex.shape[1], 0 if tex is None else tex.shape[0]), "add_mesh")
        for l in scene.lights:
            _check(lib.rtxh_scene_add_light(vp(self.h), C.byref(l)), "add_light")
        for c, r, mat in scene.spheres:
            c, mat = np.ascontiguousarray(c, np.float32), np.ascontiguousarray(mat, np.float32)
            _check(lib.rtxh_scene_add_sphere(vp(self.h), _ptr(c), C.c_float(r), _ptr(mat)), "add_sphere")
        if scene.envmap is not None:
            e = scene.envmap
            img, bg = e["image"], np.ascontiguousarray(e["background"], np.float32)
            _check(lib.rtxh_scene_set_envmap(vp(self.h), _ptr(img), 0 if img is None else img.shape[1], 0 if img is None else img.shape[0],
                                             e["mapping"], C.c_float(e["vfov_degrees"]), _ptr(bg)), "set_envmap")

    def close(self):
        if self.h:
            host_lib().rtxh_scene_free(vp(self.h))
            self.h = None


class FlatScene:
    """The flattened, device-ready scene as numpy arrays (host only, no GPU)."""

    def __init__(self, handle, owned):
        lib = host_lib()
        self.h, self.owned = handle, owned
        cnt = np.zeros(9, np.int64)
        lib.rtxh_flat_counts(vp(handle), _ptr(cnt))
        self.num_levels, self.num_leaves = int(cnt[7]), int(cnt[8])
        self.nodes = np.zeros(cnt[0], abi.NODE_DTYPE)
        self.tris = np.zeros(cnt[1], abi.TRI_DTYPE)
        self.attrs = np.zeros(cnt[1], abi.TRI_ATTR_DTYPE)
        self.materials = np.zeros(cnt[4], abi.MATERIAL_DTYPE)
        self.textures = (abi.Texture * max(1, int(cnt[2])))()
        self.n_textures = int(cnt[2])
        self.texels = np.zeros((cnt[3], 3), np.float32)
        lib.rtxh_flat_copy(vp(handle), _ptr(self.nodes), _ptr(self.tris), _ptr(self.attrs), _ptr(self.materials),
                           C.cast(self.textures, vp), _ptr(self.texels))
        ref, lo, hi = C.c_int32(), np.zeros(3, np.float32), np.zeros(3, np.float32)
        lib.rtxh_flat_root(vp(handle), C.byref(ref), _ptr(lo), _ptr(hi))
        self.root_ref, self.root_lo, self.root_hi = ref.value, lo, hi

    def desc(self):
        """rtx_scene_desc pointing into the native FlatScene (valid while it lives)."""
        d = abi.SceneDesc()
        host_lib().rtxh_flat_desc(vp(self.h), C.byref(d))
        return d

    def close(self):
        if self.h and self.owned:
            host_lib().rtxh_flat_free(vp(self.h))
        self.h = None

    def tree(self):
        """The tree in the form the CPU checkers dump it (children lists per node in
        the reference's post-order numbering is not kept; this returns, per leaf in
        depth-first left-to-right order, the list of triangle ids, plus every node's
        box in the same traversal order) -- enough to compare two builders."""
        leaves, boxes = [], []

        def leaf_ids(ref):
            k = ~ref
            ids = []
            while True:
                ids.append(int(self.tris["id"][k]))
                if self.tris["flags"][k] & 1:
                    break
                k += 1
            return ids

        def walk(ref, lo, hi):
            boxes.append(np.concatenate([lo, hi]))
            if ref < 0:
                leaves.append(leaf_ids(ref))
                return
            n = self.nodes[ref]
            walk(int(n["ref0"]), n["lo0"], n["hi0"])
            walk(int(n["ref1"]), n["lo1"], n["hi1"])

        if self.root_ref != -2 ** 31:
            import sys
            sys.setrecursionlimit(10000)
            walk(self.root_ref, self.root_lo, self.root_hi)
        return leaves, np.array(boxes, np.float32).reshape(-1, 6)


def flatten(scene: SceneData, features: abi.Features) -> FlatScene:
    hs = HostScene(scene)
    try:
        h = host_lib().rtxh_flatten(vp(hs.h), C.byref(features))
        if not h:
            raise RuntimeError("flatten: " + host_lib().rtxh_last_error().decode())
        return FlatScene(h, True)
    finally:
        hs.close()


class GpuEngine:
    kind = "hip"

    def __init__(self, scene: SceneData, features: abi.Features, device=None, device_build=False):
        """device_build: the tree is built on the GPU (rtx_scene_build_device) instead of by the
        host's restatement of the reference's builder."""
        lib = host_lib()
        if device is not None:
            _check(lib.rtxh_select_device(int(device)), "select_device")
        self.scene = HostScene(scene)
        lib.rtxh_set_device_build(1 if device_build else 0)
        try:
            self.b = lib.rtxh_bvh_create(vp(self.scene.h), C.byref(features))
        finally:
            lib.rtxh_set_device_build(0)
        if not self.b:
            msg = lib.rtxh_last_error().decode()
            self.scene.close()
            raise RuntimeError("BvhInterface: " + msg)
        self.stats = abi.Stats()

    def close(self):
        if self.b:
            host_lib().rtxh_bvh_free(vp(self.b))
            self.b = None
        self.scene.close()

    def levels(self):
        return host_lib().rtxh_bvh_levels(vp(self.b))

    def leaves(self):
        return host_lib().rtxh_bvh_leaves(vp(self.b))

    def tree_info(self):
        """levels, leaves and, for a tree built on the GPU, the milliseconds its upload and build took."""
        out = (C.c_float * 4)()
        _check(host_lib().rtxh_bvh_tree_info(vp(self.b), out), "tree_info")
        return dict(levels=int(out[0]), leaves=int(out[1]), upload_ms=float(out[2]), build_ms=float(out[3]))

    def flat(self):
        return FlatScene(host_lib().rtxh_bvh_flat(vp(self.b)), False)

    def device_scene(self):
        """rtx_scene* for direct C-ABI calls."""
        return host_lib().rtxh_bvh_device_scene(vp(self.b))

    def intersect(self, rays, features, **_):
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        hits = np.zeros(len(rays), abi.HIT_DTYPE)
        _check(host_lib().rtxh_intersect_batch(vp(self.b), _ptr(rays), C.c_int64(len(rays)), C.byref(features), _ptr(hits), C.byref(self.stats)), "intersect")
        return hits

    def intersect_one(self, ray, features):
        """BvhInterface::intersect(Ray&, HitInfo&, Features) for one ray."""
        r = np.array(ray, abi.RAY_DTYPE).reshape(1).copy()
        out, hit = np.zeros(20, np.float32), C.c_int()
        _check(host_lib().rtxh_intersect_one(vp(self.b), _ptr(r), C.byref(features), _ptr(out), C.byref(hit)), "intersect_one")
        return bool(hit.value), np.float32(r["t"][0]), out

    def final_color(self, rays, features, ray_depth=0, one_by_one=False):
        """getFinalColor (src/render.h:29) for a batch of rays."""
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        out = np.zeros((len(rays), 3), np.float32)
        _check(host_lib().rtxh_final_color(vp(self.b), _ptr(rays), C.c_int64(len(rays)), C.byref(features), int(ray_depth), _ptr(out),
                                           int(one_by_one), C.byref(self.stats)), "final_color")
        return out

    def render(self, cam, features, params, **_):
        out = np.zeros((params.height, params.width, 3), np.float32)
        _check(host_lib().rtxh_render(vp(self.b), C.byref(cam), C.byref(features), C.byref(params), _ptr(out), C.byref(self.stats)), "render")
        return out

    def render_named(self, cam, features, driver, width, height, samples=0, aperture=0.1, focal=2.0, prev_cam=None):
        """The reference's renderRayTracing* functions with their own defaults."""
        out = np.zeros((height, width, 3), np.float32)
        prev = prev_cam if prev_cam is not None else cam
        _check(host_lib().rtxh_render_named(vp(self.b), C.byref(cam), C.byref(prev), C.byref(features), int(driver), int(width), int(height),
                                            int(samples), C.c_float(aperture), C.c_float(focal), _ptr(out)), "render_named")
        return out


def device_camera(cam: abi.Camera, width, height) -> abi.Camera:
    """Camera with the derived fields (position, quaternion, half extents) filled in
    by the host layer, as the C ABI expects it."""
    out = abi.Camera()
    host_lib().rtxh_camera(C.byref(cam), int(width), int(height), C.byref(out))
    return out


def camera_rays(cam: abi.Camera, width, height, step=1):
    """Trackball::generateRay through the corner of every step-th pixel (the positions
    renderRayTracing uses, src/render.cpp:76-79), rows bottom-up, as an abi.RAY_DTYPE array."""
    ys, xs = np.mgrid[0:height:step, 0:width:step]
    ndc = np.stack([xs.ravel().astype(np.float32) / np.float32(width) * np.float32(2) - np.float32(1),
                    ys.ravel().astype(np.float32) / np.float32(height) * np.float32(2) - np.float32(1)], 1).astype(np.float32)
    rays = np.zeros(len(ndc), abi.RAY_DTYPE)
    host_lib().rtxh_camera_rays(C.byref(cam), int(width), int(height), _ptr(np.ascontiguousarray(ndc)), C.c_int64(len(ndc)), _ptr(rays))
    return rays


def load_obj(path, normalize=False) -> "SceneData":
    """loadMesh (framework/src/mesh.cpp:52-176) through the native host loader:
    the meshes of an OBJ / MTL file (PNG textures decoded) as a SceneData."""
    from .scene import SceneData
    lib = host_lib()
    h = lib.rtxh_scene_create()
    try:
        _check(lib.rtxh_scene_load_obj(vp(h), str(path).encode(), int(bool(normalize))), "load_obj")
        s = SceneData()
        for m in range(lib.rtxh_scene_num_meshes(vp(h))):
            nv, nt, tw, th = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            lib.rtxh_mesh_sizes(vp(h), m, C.byref(nv), C.byref(nt), C.byref(tw), C.byref(th))
            v, t, mat = np.zeros((nv.value, 8), np.float32), np.zeros((nt.value, 3), np.uint32), np.zeros(8, np.float32)
            tex = np.zeros((th.value, tw.value, 3), np.float32) if tw.value else None
            lib.rtxh_mesh_copy(vp(h), m, _ptr(v), _ptr(t), _ptr(mat), _ptr(tex))
            s.add_mesh(v, t, mat, tex)
        return s
    finally:
        lib.rtxh_scene_free(vp(h))


def load_image(path):
    """Image (framework/src/image.cpp:13-36): (h, w, 3) float32 in [0, 1] from an 8-bit PNG."""
    lib = host_lib()
    w, h = C.c_int(), C.c_int()
    _check(lib.rtxh_image_load(str(path).encode(), C.byref(w), C.byref(h), None), "load_image")
    out = np.zeros((h.value, w.value, 3), np.float32)
    _check(lib.rtxh_image_load(str(path).encode(), C.byref(w), C.byref(h), _ptr(out)), "load_image")
    return out
