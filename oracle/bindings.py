"""TEST INFRASTRUCTURE.  ctypes drivers for the two CPU checkers:

  OracleEngine  oracle/liboracle.so   the CPU restatement (rt_oracle.cpp)
  RefEngine     oracle/_ref/libref.so the reference's own translation units
                                      (exists only where /root/reference was
                                      available at build time)

Both expose the same methods, so a test can run the same case through either.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd import abi
from raytracer_amd.scene import SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RTX_ORACLE_SO: another build of the restatement, e.g. the sanitizer build of `make asan`)
ORACLE_SO = os.environ.get("RTX_ORACLE_SO") or os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref.so")

vp = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


def have_ref():
    return os.path.exists(REF_SO)


class _Engine:
    prefix = None
    so = None
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            lib = C.CDLL(cls.so)
            p = cls.prefix
            for name in ("scene_create", "scene_empty", "scene_prebuilt", "bvh_create"):
                if hasattr(lib, p + name):
                    getattr(lib, p + name).restype = vp
            for name in ("num_nodes", "num_children", "dump", "bvh_num_nodes", "bvh_num_children", "bvh_dump"):
                if hasattr(lib, p + name):
                    getattr(lib, p + name).restype = C.c_int64
            cls._lib = lib
        return cls._lib

    def fn(self, name):
        return getattr(self.lib(), self.prefix + name)

    # ---- scene upload --------------------------------------------------------
    def _fill(self, handle, scene: SceneData):
        for m in scene.meshes:
            tex = m.texture
            self.fn("scene_add_mesh")(vp(handle), _ptr(m.vertices), len(m.vertices), _ptr(m.triangles), len(m.triangles),
                                      _ptr(m.material), _ptr(tex), 0 if tex is None else tex.shape[1], 0 if tex is None else tex.shape[0])
        for l in scene.lights:
            self.fn("scene_add_light")(vp(handle), C.byref(l))
        for c, r, mat in scene.spheres:
            c = np.ascontiguousarray(c, np.float32)
            mat = np.ascontiguousarray(mat, np.float32)
            self.fn("scene_add_sphere")(vp(handle), _ptr(c), C.c_float(r), _ptr(mat))
        if scene.envmap is not None:
            e = scene.envmap
            img = e["image"]
            bg = np.ascontiguousarray(e["background"], np.float32)
            self.fn("scene_set_envmap")(vp(handle), _ptr(img), 0 if img is None else img.shape[1], 0 if img is None else img.shape[0],
                                        e["mapping"], C.c_float(e["vfov_degrees"]), _ptr(bg))

    # ---- known-answer helpers (same signature in both libraries) -------------
    def kat_slab(self, lo, hi, ray):
        r = np.array(ray, abi.RAY_DTYPE).reshape(1).copy()
        lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
        ok = self.fn("kat_slab")(_ptr(lo), _ptr(hi), _ptr(r))
        return bool(ok), np.float32(r["t"][0])

    def kat_triangle(self, a, b, c, ray):
        r = np.array(ray, abi.RAY_DTYPE).reshape(1).copy()
        a, b, c = (np.asarray(x, np.float32) for x in (a, b, c))
        bary, n = np.zeros(3, np.float32), np.zeros(3, np.float32)
        ok = self.fn("kat_triangle")(_ptr(a), _ptr(b), _ptr(c), _ptr(r), _ptr(bary), _ptr(n))
        return bool(ok), np.float32(r["t"][0]), bary, n

    def kat_sphere(self, center, radius, ray):
        r = np.array(ray, abi.RAY_DTYPE).reshape(1).copy()
        center = np.asarray(center, np.float32)
        n = np.zeros(3, np.float32)
        ok = self.fn("kat_sphere")(_ptr(center), C.c_float(radius), _ptr(r), _ptr(n))
        return bool(ok), np.float32(r["t"][0]), n

    def kat_barycentric(self, a, b, c, p):
        a, b, c, p = (np.asarray(x, np.float32) for x in (a, b, c, p))
        out = np.zeros(3, np.float32)
        self.fn("kat_barycentric")(_ptr(a), _ptr(b), _ptr(c), _ptr(p), _ptr(out))
        return out

    def kat_shading(self, light_pos, light_col, ray, normal, material):
        r = np.array(ray, abi.RAY_DTYPE).reshape(1).copy()
        lp, lc, n, m = (np.asarray(x, np.float32) for x in (light_pos, light_col, normal, material))
        out = np.zeros(3, np.float32)
        self.fn("kat_shading")(_ptr(lp), _ptr(lc), _ptr(r), _ptr(n), _ptr(m), _ptr(out))
        return out

    def kat_reflect(self, ray, normal):
        r = np.array(ray, abi.RAY_DTYPE).reshape(1).copy()
        n = np.asarray(normal, np.float32)
        out = np.zeros(1, abi.RAY_DTYPE)
        self.fn("kat_reflect")(_ptr(r), _ptr(n), _ptr(out))
        return out[0]

    def kat_texture(self, image, mode, uv):
        image = np.ascontiguousarray(image, np.float32)
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 3), np.float32)
        self.fn("kat_texture")(_ptr(image), image.shape[1], image.shape[0], int(mode), _ptr(uv), len(uv), _ptr(out))
        return out

    def kat_mipmap(self, image):
        image = np.ascontiguousarray(image, np.float32)
        sizes = np.zeros(2 * 32, np.int32)
        n = self.fn("kat_mipmap")(_ptr(image), image.shape[1], image.shape[0], _ptr(sizes), None)
        # pixel count per level follows the builder's loop bounds (src/main.cpp:757-758), not w*h
        counts = []
        for l in range(n):
            if l == 0:
                counts.append(image.shape[0] * image.shape[1])
            else:
                pw, ph = int(sizes[2 * (l - 1)]), int(sizes[2 * (l - 1) + 1])
                counts.append(((ph + 1) // 2) * ((pw + 1) // 2))
        pix = np.zeros(sum(counts) * 3, np.float32)
        self.fn("kat_mipmap")(_ptr(image), image.shape[1], image.shape[0], _ptr(sizes), _ptr(pix))
        levels, off = [], 0
        for l in range(n):
            levels.append(pix[off:off + counts[l] * 3].reshape(-1, 3).copy())
            off += counts[l] * 3
        return sizes[:2 * n].reshape(n, 2).copy(), levels

    def kat_envmap(self, image, mapping, vfov_degrees, background, bilinear, dirs):
        image = np.ascontiguousarray(image, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        bg = np.asarray(background, np.float32)
        out = np.zeros((len(dirs), 3), np.float32)
        self.fn("kat_envmap")(_ptr(image), image.shape[1], image.shape[0], int(mapping), C.c_float(vfov_degrees), _ptr(bg),
                              int(bilinear), _ptr(dirs), len(dirs), _ptr(out))
        return out

    def kat_env_bins(self, image, cap=4096):
        image = np.ascontiguousarray(image, np.float32)
        out = np.zeros((cap, 4), np.float32)
        fn = self.fn("kat_env_bins")
        fn.restype = C.c_int
        n = fn(_ptr(image), image.shape[1], image.shape[0], _ptr(out), cap)
        assert n <= cap
        return out[:n].copy()

    def kat_sample2d(self, rect, h, k, seed, pixel, index):
        rect = np.asarray(rect, np.float32)
        out = np.zeros((h * k, 2), np.float32)
        self.fn("kat_sample2d")(_ptr(rect), int(h), int(k), C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(index), _ptr(out))
        return out

    def kat_env_rays(self, image, mapping, vfov_degrees, position, normal, n, seed, pixel, cap=4096):
        image = np.ascontiguousarray(image, np.float32)
        pos, nrm = np.asarray(position, np.float32), np.asarray(normal, np.float32)
        out = np.zeros(cap, abi.RAY_DTYPE)
        fn = self.fn("kat_env_rays")
        fn.restype = C.c_int
        cnt = fn(_ptr(image), image.shape[1], image.shape[0], int(mapping), C.c_float(vfov_degrees), _ptr(pos), _ptr(nrm), int(n),
                 C.c_uint32(seed), C.c_uint32(pixel), _ptr(out), cap)
        assert cnt <= cap
        return out[:cnt].copy()

    def kat_light_samples(self, light, seed, pixel, n):
        pos, col = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        self.fn("kat_light_samples")(C.byref(light), C.c_uint32(seed), C.c_uint32(pixel), n, _ptr(pos), _ptr(col))
        return pos, col

    def kat_glossy(self, reflection, blur, n_rays, seed, pixel):
        r = np.array(reflection, abi.RAY_DTYPE).reshape(1).copy()
        out = np.zeros(n_rays, abi.RAY_DTYPE)
        self.fn("kat_glossy")(_ptr(r), C.c_float(blur), n_rays, C.c_uint32(seed), C.c_uint32(pixel), _ptr(out))
        return out

    def kat_minstd(self, skip, n, a=0.0, b=1.0):
        out = np.zeros(n, np.float32)
        self.fn("kat_minstd")(C.c_uint64(skip), int(n), C.c_float(a), C.c_float(b), _ptr(out))
        return out

    def kat_camera(self, cam, aspect, ndc):
        ndc = np.ascontiguousarray(ndc, np.float32).reshape(-1, 2)
        out = np.zeros(len(ndc), abi.RAY_DTYPE)
        self.fn("kat_camera")(C.byref(cam), C.c_float(cam.fovy), C.c_float(aspect), _ptr(ndc), len(ndc), _ptr(out))
        return out


class OracleEngine(_Engine):
    prefix = "orc_"
    so = ORACLE_SO
    kind = "port"

    def __init__(self, scene: SceneData, features: abi.Features):
        self.h = self.fn("scene_create")()
        self._fill(self.h, scene)
        self.fn("build")(vp(self.h), C.byref(features))
        self.counters = np.zeros(6, np.uint64)

    def close(self):
        if self.h:
            self.fn("scene_free")(vp(self.h))
            self.h = None

    def levels(self):
        return self.fn("levels")(vp(self.h))

    def leaves(self):
        return self.fn("leaves")(vp(self.h))

    def dump(self):
        n = self.fn("num_nodes")(vp(self.h))
        nc = self.fn("num_children")(vp(self.h))
        aabb, leaf = np.zeros((n, 6), np.float32), np.zeros(n, np.int32)
        off, ch = np.zeros(n + 1, np.int64), np.zeros(nc, np.int64)
        root = self.fn("dump")(vp(self.h), _ptr(aabb), _ptr(leaf), _ptr(off), _ptr(ch))
        return dict(root=int(root), aabb=aabb, leaf=leaf, offsets=off, children=ch)

    def intersect(self, rays, features, threads=0):
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        hits = np.zeros(len(rays), abi.HIT_DTYPE)
        self.fn("intersect")(vp(self.h), _ptr(rays), C.c_int64(len(rays)), C.byref(features), _ptr(hits), _ptr(self.counters), int(threads))
        return hits

    def final_color(self, rays, features, params, threads=0):
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        out = np.zeros((len(rays), 3), np.float32)
        self.fn("shade")(vp(self.h), _ptr(rays), C.c_int64(len(rays)), C.byref(features), C.byref(params), _ptr(out), _ptr(self.counters), int(threads))
        return out

    def render(self, cam, features, params, threads=0, **_):
        out = np.zeros((params.height, params.width, 3), np.float32)
        rc = self.fn("render")(vp(self.h), C.byref(cam), C.byref(features), C.byref(params), _ptr(out), _ptr(self.counters), int(threads))
        if rc:
            raise RuntimeError("orc_render failed")
        return out


class RefEngine(_Engine):
    prefix = "ref_"
    so = REF_SO
    kind = "reference"

    def __init__(self, scene, features: abi.Features):
        """scene: SceneData, or an int SceneType to go through the reference's own
        loadScenePrebuilt (src/scene.cpp:5-103)."""
        if isinstance(scene, int):
            self.h = self.fn("scene_prebuilt")(scene)
            if not self.h:
                raise RuntimeError("reference failed to load prebuilt scene %d" % scene)
        else:
            self.h = self.fn("scene_empty")()
            self._fill(self.h, scene)
        self.b = self.fn("bvh_create")(vp(self.h), C.byref(features))

    def close(self):
        if self.h:
            if self.b:
                self.fn("bvh_free")(vp(self.b))
            self.fn("scene_free")(vp(self.h))
            self.h = None

    @classmethod
    def load_obj(cls, path, normalize=False) -> SceneData:
        """The reference's own loadMesh (framework/src/mesh.cpp:52-149) on an OBJ file."""
        e = cls.__new__(cls)
        fn = e.fn("scene_load_obj")
        fn.restype = vp
        e.h = fn(str(path).encode(), int(bool(normalize)))
        if not e.h:
            raise RuntimeError("the reference's loader rejected " + str(path))
        e.b = None
        try:
            return e.export_scene()
        finally:
            e.close()

    def set_envmap(self, image, mapping, vfov_degrees, background):
        img = None if image is None else np.ascontiguousarray(image, np.float32)
        bg = np.ascontiguousarray(background, np.float32)
        self.fn("scene_set_envmap")(vp(self.h), _ptr(img), 0 if img is None else img.shape[1], 0 if img is None else img.shape[0],
                                    int(mapping), C.c_float(vfov_degrees), _ptr(bg))

    def set_lights(self, lights):
        self.fn("scene_clear_lights")(vp(self.h))
        for l in lights:
            self.fn("scene_add_light")(vp(self.h), C.byref(l))

    def export_scene(self) -> SceneData:
        """The Scene as the reference holds it after loading (vertex de-duplication,
        per-material split and unit normalisation already applied)."""
        s = SceneData()
        for m in range(self.fn("scene_num_meshes")(vp(self.h))):
            nv, nt, tw, th = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            self.fn("mesh_sizes")(vp(self.h), m, C.byref(nv), C.byref(nt), C.byref(tw), C.byref(th))
            v, t = np.zeros((nv.value, 8), np.float32), np.zeros((nt.value, 3), np.uint32)
            mat = np.zeros(8, np.float32)
            tex = np.zeros((th.value, tw.value, 3), np.float32) if tw.value else None
            self.fn("mesh_copy")(vp(self.h), m, _ptr(v), _ptr(t), _ptr(mat), _ptr(tex))
            s.add_mesh(v, t, mat, tex)
        for i in range(self.fn("scene_num_lights")(vp(self.h))):
            l = abi.Light()
            self.fn("scene_get_light")(vp(self.h), i, C.byref(l))
            s.lights.append(l)
        for i in range(self.fn("scene_num_spheres")(vp(self.h))):
            c, r, mat = np.zeros(3, np.float32), C.c_float(), np.zeros(8, np.float32)
            self.fn("scene_get_sphere")(vp(self.h), i, _ptr(c), C.byref(r), _ptr(mat))
            s.add_sphere(c, r.value, mat)
        return s

    def levels(self):
        return self.fn("bvh_levels")(vp(self.b))

    def leaves(self):
        return self.fn("bvh_leaves")(vp(self.b))

    def dump(self):
        n = self.fn("bvh_num_nodes")(vp(self.b))
        nc = self.fn("bvh_num_children")(vp(self.b))
        aabb, leaf = np.zeros((n, 6), np.float32), np.zeros(n, np.int32)
        off, ch = np.zeros(n + 1, np.int64), np.zeros(nc, np.int64)
        root = self.fn("bvh_dump")(vp(self.b), _ptr(aabb), _ptr(leaf), _ptr(off), _ptr(ch))
        return dict(root=int(root), aabb=aabb, leaf=leaf, offsets=off, children=ch)

    def intersect(self, rays, features, threads=0):
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        hits = np.zeros(len(rays), abi.HIT_DTYPE)
        self.fn("intersect")(vp(self.b), vp(self.h), _ptr(rays), C.c_int64(len(rays)), C.byref(features), _ptr(hits))
        return hits

    def final_color(self, rays, features, params, threads=0):
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        out = np.zeros((len(rays), 3), np.float32)
        self.fn("shade")(vp(self.h), vp(self.b), _ptr(rays), C.c_int64(len(rays)), C.byref(features), C.byref(params), _ptr(out))
        return out

    def render(self, cam, features, params, threads=0, mode=0):
        """mode 0: the reference's own renderRayTracing* driver; mode 1: harness pixel
        loop around getFinalColor (selectable depth, keyed RNG)."""
        out = np.zeros((params.height, params.width, 3), np.float32)
        rc = self.fn("render")(vp(self.h), vp(self.b), C.byref(cam), C.byref(features), C.byref(params), _ptr(out), int(mode), int(threads))
        if rc:
            raise RuntimeError("ref_render failed")
        return out

    def image_load(self, path):
        w, h = C.c_int(), C.c_int()
        if self.fn("image_load")(path.encode(), C.byref(w), C.byref(h), None):
            raise RuntimeError("cannot load " + path)
        rgb = np.zeros((h.value, w.value, 3), np.float32)
        self.fn("image_load")(path.encode(), C.byref(w), C.byref(h), _ptr(rgb))
        return rgb


def primary_rays(cam: abi.Camera, width, height, engine_cls=OracleEngine, step=1):
    """Rays through pixel corners (src/render.cpp:76-79) of every step-th pixel."""
    ys, xs = np.mgrid[0:height:step, 0:width:step]
    ndc = np.stack([xs.ravel().astype(np.float32) / np.float32(width) * np.float32(2) - np.float32(1),
                    ys.ravel().astype(np.float32) / np.float32(height) * np.float32(2) - np.float32(1)], 1).astype(np.float32)
    eng = engine_cls.__new__(engine_cls)
    return eng.kat_camera(cam, np.float32(width) / np.float32(height), ndc)
