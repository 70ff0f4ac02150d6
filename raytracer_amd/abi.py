"""ctypes / numpy mirrors of the plain-data structs of include/rtx.h.

Field order and sizes must match the header exactly; tests/test_abi.py checks
every sizeof against the compiled library.
"""
import ctypes as C

import numpy as np

i32, u32, f32 = C.c_int32, C.c_uint32, C.c_float

RTX_TRI_NONE = -1
RTX_MAX_MIP_LEVELS = 16

DRIVER_PLAIN, DRIVER_MRAYS, DRIVER_DOF, DRIVER_TRANSPARENCY, DRIVER_MOTION_BLUR, DRIVER_COMPOSED = range(6)
LIGHT_POINT, LIGHT_SEGMENT, LIGHT_PARALLELOGRAM = range(3)
ENV_CYLINDRICAL, ENV_SPHERICAL = range(2)


class Features(C.Structure):
    """Features + ExtraFeatures of the reference (src/common.h:61-87)."""
    _fields_ = [(n, i32) for n in (
        "enable_shading", "enable_recursive", "enable_hard_shadow", "enable_soft_shadow",
        "enable_normal_interp", "enable_texture_mapping", "enable_accel_structure",
        "enable_environment_mapping", "enable_bvh_sah_binning",
        "enable_bilinear_texture_filtering", "enable_mipmap_texture_filtering",
        "enable_glossy_reflection")] + [("reserved", i32 * 4)]


class Camera(C.Structure):
    _fields_ = [("look_at", f32 * 3), ("euler", f32 * 3), ("distance", f32), ("fovy", f32),
                ("half_width", f32), ("half_height", f32), ("position", f32 * 3), ("quat", f32 * 4)]


class RenderParams(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("driver", i32), ("ray_depth", i32), ("samples", i32),
                ("aperture", f32), ("focal_length", f32), ("shadow_samples", i32), ("glossy_rays", i32),
                ("seed", u32), ("tile_size", i32), ("tile_rank", i32), ("tile_world", i32),
                ("prev_camera", Camera), ("lens_samples", i32), ("time_steps", i32), ("transparency", i32), ("reserved", i32 * 5)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_bounce", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("shaded_hits", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("rays_traced", C.c_uint64), ("layer_overflows", C.c_uint64)]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_bounce


class Light(C.Structure):
    _fields_ = [("type", i32), ("p", (f32 * 3) * 3), ("c", (f32 * 3) * 4), ("pad", i32 * 2)]


class EnvMap(C.Structure):
    _fields_ = [("texture", i32), ("mapping", i32), ("vfov_radians", f32), ("background", f32 * 3)]


class Texture(C.Structure):
    _fields_ = [("levels", i32), ("width", i32 * RTX_MAX_MIP_LEVELS), ("height", i32 * RTX_MAX_MIP_LEVELS),
                ("offset", u32 * RTX_MAX_MIP_LEVELS), ("count", u32 * RTX_MAX_MIP_LEVELS)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", u32), ("root_ref", i32), ("root_cnt", i32),
                ("root_lo", f32 * 3), ("root_hi", f32 * 3),
                ("tris", C.c_void_p), ("tri_attrs", C.c_void_p), ("n_tris", u32),
                ("materials", C.c_void_p), ("n_materials", u32),
                ("textures", C.c_void_p), ("n_textures", u32),
                ("texels", C.c_void_p), ("n_texels", C.c_uint64),
                ("lights", C.c_void_p), ("n_lights", u32),
                ("spheres", C.c_void_p), ("n_spheres", u32),
                ("envmap", EnvMap), ("num_levels", i32), ("num_leaves", i32),
                ("env_bins", C.c_void_p), ("n_env_bins", u32)]


RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("t", "<f4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("tri", "<i4"), ("normal", "<f4", 3), ("bary", "<f4", 3),
                      ("uv", "<f4", 2), ("material", "<i4"), ("hit", "<i4")])
NODE_DTYPE = np.dtype([("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3),
                       ("ref0", "<i4"), ("ref1", "<i4"), ("cnt0", "<i4"), ("cnt1", "<i4")])
TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("n", "<f4", 3), ("d", "<f4"),
                      ("id", "<i4"), ("material", "<i4"), ("flags", "<i4")])
TRI_ATTR_DTYPE = np.dtype([("n0", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3),
                           ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("pad", "<f4")])
MATERIAL_DTYPE = np.dtype([("kd", "<f4", 3), ("ks", "<f4", 3), ("shininess", "<f4"), ("transparency", "<f4"),
                           ("texture", "<i4"), ("pad", "<i4", 3)])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<i4"), ("pad", "<i4", 3)])

assert RAY_DTYPE.itemsize == 28 and HIT_DTYPE.itemsize == 48
assert NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 64 and TRI_ATTR_DTYPE.itemsize == 64
assert MATERIAL_DTYPE.itemsize == 48 and SPHERE_DTYPE.itemsize == 32 and C.sizeof(Light) == 96

GLM_RADIANS = np.float32(0.01745329251994329576923690768489)


def features(**kw):
    """Features with the given switches on, e.g. features(shading=1, hard_shadow=1)."""
    f = Features()
    for k, v in kw.items():
        name = k if k.startswith("enable_") else "enable_" + k
        if not hasattr(f, name):
            raise AttributeError(name)
        setattr(f, name, int(v))
    return f


def make_light(kind, p, c):
    l = Light()
    l.type = kind
    for i, v in enumerate(p):
        l.p[i][:] = [float(x) for x in v]
    for i, v in enumerate(c):
        l.c[i][:] = [float(x) for x in v]
    return l


def point_light(position, color=(1, 1, 1)):
    return make_light(LIGHT_POINT, [position], [color])


def segment_light(e0, e1, c0, c1):
    return make_light(LIGHT_SEGMENT, [e0, e1], [c0, c1])


def parallelogram_light(v0, edge01, edge02, c0, c1, c2, c3):
    return make_light(LIGHT_PARALLELOGRAM, [v0, edge01, edge02], [c0, c1, c2, c3])
