"""Seeded random scenes against the oracle: triangle soups of unrelated sizes (no mesh
structure for the tree builder to lean on, leaves that overlap, slivers, zero-area
triangles), spheres among them, lights inside and outside the geometry, cameras outside
and inside it, under every driver.  Bars as in test_gpu_parity.py: ray queries bit-exact,
frames within 1e-3 with NaN pixels in the same places, identical ray counts.
"""
import os

import numpy as np
import pytest

import cases
from raytracer_amd import abi
from raytracer_amd.engine import GpuEngine
from raytracer_amd.scene import SceneData, make_camera, make_params
from oracle.bindings import OracleEngine
from test_gpu_parity import RGB_TOL, assert_hits_equal, assert_images_close

pytestmark = pytest.mark.gpu


def soup(rng, n, spread, size, degenerate=0.03):
    """n unconnected triangles: centres within `spread` of the origin, edges up to `size`;
    vertex normals near the face normal; a few triangles with two equal corners."""
    c = (rng.random((n, 1, 3)) * 2 - 1) * spread
    p = c + (rng.random((n, 3, 3)) * 2 - 1) * (size * rng.random((n, 1, 1)) ** 2 + 0.01)
    # slivers: the third corner almost on the line through the other two
    thin = rng.random(n) < 0.1
    p[thin, 2] = p[thin, 0] + (p[thin, 1] - p[thin, 0]) * 0.5 + (rng.random((int(thin.sum()), 3)) - 0.5) * 1e-3
    same = rng.random(n) < degenerate
    p[same, 2] = p[same, 1]
    p = p.astype(np.float32)
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    fn = np.where(ln > 0, fn / np.where(ln > 0, ln, 1), (0, 1, 0))
    vn = fn[:, None, :] + (rng.random((n, 3, 3)) - 0.5) * 0.6
    vn /= np.linalg.norm(vn, axis=2, keepdims=True)
    v = np.zeros((n * 3, 8), np.float32)
    v[:, 0:3] = p.reshape(-1, 3)
    v[:, 3:6] = vn.reshape(-1, 3)
    v[:, 6:8] = rng.random((n * 3, 2))
    return v, np.arange(n * 3, dtype=np.uint32).reshape(n, 3)


def random_scene(seed):
    rng = np.random.default_rng(1000 + seed)
    s = SceneData()
    for n in rng.choice([7, 60, 200, 500], size=int(rng.integers(1, 4))):
        kd = rng.random(3)
        ks = rng.random(3) * (rng.random() < 0.4)
        transparency = 1.0 if rng.random() < 0.5 else 0.3 + 0.6 * rng.random()
        v, t = soup(rng, int(n), spread=0.3 + 0.4 * rng.random(), size=0.2 + 0.6 * rng.random())
        s.add_mesh(v, t, (*kd, *ks, 1.0 + 60.0 * rng.random(), transparency))
    for _ in range(int(rng.integers(0, 3))):
        s.add_sphere((rng.random(3) * 2 - 1) * 0.7, 0.05 + 0.25 * rng.random(), (*rng.random(3), *(rng.random(3) * 0.5 * (rng.random() < 0.5)), 20.0, 1.0))
    # (now and then more point lights than the four whose shadow rays get records of their own)
    for _ in range(int(rng.integers(1, 3)) if rng.random() < 0.85 else int(rng.integers(5, 8))):
        d = rng.normal(size=3)
        s.lights.append(abi.point_light(d / np.linalg.norm(d) * (0.2 + 2.5 * rng.random()), 0.4 + 0.6 * rng.random(3)))
    return s, rng


def random_camera(rng, distance=None):
    return make_camera(fov_degrees=35.0 + 25.0 * rng.random(), distance=distance if distance is not None else 1.3 + 1.2 * rng.random(),
                       look_at=tuple((rng.random(3) - 0.5) * 0.4), rotation_degrees=(rng.random() * 360 - 180, rng.random() * 360, 0))


def check_frame(gpu, orc, cam, f, rp, what, ties=0):
    """`ties`: pixels that may differ.  Where a ray meets two triangles at the same distance to the
    last bit (it crosses the line in which they intersect) the triangle tested first wins, in the
    reference as here, and the order of the tests is the one thing the walks do not share; one
    sample of one pixel in 200 soups' 400x224x4 frames does (soup 113)."""
    got = gpu.render(cam, f, rp)
    counts = (gpu.stats.rays_primary, gpu.stats.rays_shadow, gpu.stats.rays_bounce)
    want = orc.render(cam, f, rp)
    if ties:
        off = np.abs(np.nan_to_num(got) - np.nan_to_num(want)).max(-1) > RGB_TOL
        assert off.sum() <= ties, f"{what}: {off.sum()} pixels differ"
        got = np.where(off[..., None], want, got)
    assert_images_close(got, want, what)
    assert counts == tuple(int(x) for x in orc.counters[:3]), f"{what}: ray counts {counts} vs oracle {tuple(orc.counters[:3])}"


# (RTX_FUZZ_SEEDS=1000 for a longer hunt)
@pytest.mark.parametrize("seed", range(int(os.environ.get("RTX_FUZZ_SEEDS", "48"))))
def test_random_soups_under_every_driver(gpu_device, seed):
    scene, rng = random_scene(seed)
    outside, outside2, inside = random_camera(rng), random_camera(rng), random_camera(rng, distance=0.15)
    base = abi.features(**cases.BASE)
    mirror = abi.features(**dict(cases.BASE, recursive=1))
    gpu, orc = GpuEngine(scene, base), OracleEngine(scene, base)
    # ray queries: the tree walk alone, on rays of every kind
    rays = cases.ray_batch(scene, seed, n_random=3000, grid=32)
    assert_hits_equal(gpu.intersect(rays, base), orc.intersect(rays, base), f"soup {seed} rays")
    for cam, name in ((outside, "outside"), (inside, "inside")):
        check_frame(gpu, orc, cam, base, make_params(64, 36), f"soup {seed} {name}, plain")
        check_frame(gpu, orc, cam, mirror, make_params(48, 27, ray_depth=3), f"soup {seed} {name}, mirrors to depth 3")
        check_frame(gpu, orc, cam, base, make_params(48, 27, driver=abi.DRIVER_MRAYS, samples=4), f"soup {seed} {name}, 4 samples")
    # (most of a soup lies in some triangle's shadow: without shadows every hit's shading counts)
    unshadowed = abi.features(shading=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1, recursive=1)
    check_frame(gpu, orc, outside, unshadowed, make_params(64, 36), f"soup {seed}, no shadows")
    check_frame(gpu, orc, outside2, unshadowed, make_params(48, 27, driver=abi.DRIVER_MRAYS, samples=3), f"soup {seed}, no shadows, 3 samples")
    # few pixels against many triangles: the frame's rays no longer run side by side
    check_frame(gpu, orc, outside2, base, make_params(24, 14), f"soup {seed}, 24x14")
    check_frame(gpu, orc, outside2, base, make_params(40, 22, driver=abi.DRIVER_DOF, samples=4, aperture=0.08, focal_length=2.0), f"soup {seed}, depth of field")
    check_frame(gpu, orc, outside, mirror, make_params(40, 22, driver=abi.DRIVER_MOTION_BLUR, samples=3, prev_camera=outside2), f"soup {seed}, motion blur")
    check_frame(gpu, orc, outside, mirror, make_params(48, 27, driver=abi.DRIVER_TRANSPARENCY), f"soup {seed}, transparency")
    check_frame(gpu, orc, inside, base, make_params(37, 21, driver=abi.DRIVER_TRANSPARENCY), f"soup {seed}, transparency from inside")
    check_frame(gpu, orc, inside, mirror, make_params(33, 19, driver=abi.DRIVER_DOF, samples=3, aperture=float(0.3 * rng.random()), focal_length=float(0.2 + 3 * rng.random())),
                f"soup {seed}, depth of field from inside")
    check_frame(gpu, orc, inside, base, make_params(31, 17, driver=abi.DRIVER_MOTION_BLUR, samples=5, prev_camera=outside), f"soup {seed}, motion blur across the scene")
    gpu.close()
    orc.close()
    # the other tree (median split, flat normals) and no tree at all
    for feat in (dict(shading=1, hard_shadow=1, accel_structure=1), dict(shading=1, hard_shadow=1, normal_interp=1)):
        f = abi.features(**feat)
        gpu, orc = GpuEngine(scene, f), OracleEngine(scene, f)
        assert_hits_equal(gpu.intersect(rays[:1500], f), orc.intersect(rays[:1500], f), f"soup {seed} rays, {feat}")
        check_frame(gpu, orc, outside, f, make_params(40, 22), f"soup {seed}, {feat}")
        gpu.close()
        orc.close()
    # getFinalColor on rays that are no camera's: started inside and outside the soup, some along
    # an axis, some with a search limit in front of the geometry
    gpu, orc = GpuEngine(scene, mirror), OracleEngine(scene, mirror)
    loose = rays[-1200:]
    got, want = gpu.final_color(loose, mirror, ray_depth=2), orc.final_color(loose, mirror, make_params(1, 1, ray_depth=2))
    assert_images_close(got, want, f"soup {seed}, getFinalColor on loose rays")
    assert (gpu.stats.rays_primary, gpu.stats.rays_shadow, gpu.stats.rays_bounce) == tuple(int(x) for x in orc.counters[:3])
    gpu.close()
    orc.close()
    # the tree built on the device (another tree: same hits), and a frame large enough for the
    # scheduler's block classes, culling and ordering to matter
    dev, orc = GpuEngine(scene, base, device_build=True), OracleEngine(scene, base)
    got, want = dev.intersect(rays, base), orc.intersect(rays, base)
    hit = want["hit"] != 0
    assert np.array_equal(got["hit"] != 0, hit) and np.array_equal(got["t"][hit].view(np.uint32), want["t"][hit].view(np.uint32)), f"soup {seed}, device-built tree"
    check_frame(dev, orc, outside, mirror, make_params(56, 32), f"soup {seed}, device-built tree")
    dev.close()
    gpu = GpuEngine(scene, base)
    big = make_params(400, 224, driver=abi.DRIVER_MRAYS, samples=4)
    check_frame(gpu, orc, outside2, base, big, f"soup {seed}, 400x224x4", ties=1)
    check_frame(gpu, orc, outside2, base, big, f"soup {seed}, 400x224x4 again (blocks in the measured order)", ties=1)
    gpu.close()
    orc.close()
    # area lights: a segment and a parallelogram, glossy reflections to depth 2
    soft = abi.features(**dict(cases.BASE, hard_shadow=0, soft_shadow=1, recursive=1, glossy_reflection=1))
    scene.lights = [abi.segment_light((-0.5, 1.0, 0.2), (0.5, 1.1, -0.2), (1, 0.8, 0.6), (0.6, 0.8, 1)),
                    abi.parallelogram_light((0.9, -0.2, -0.2), (0, 0.4, 0), (0, 0, 0.4), (1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1))]
    gpu, orc = GpuEngine(scene, soft), OracleEngine(scene, soft)
    check_frame(gpu, orc, outside, soft, make_params(40, 22, shadow_samples=6, glossy_rays=2, ray_depth=2), f"soup {seed}, area lights + glossy")
    check_frame(gpu, orc, inside, soft, make_params(32, 18, driver=abi.DRIVER_MRAYS, samples=2, shadow_samples=3, glossy_rays=2, ray_depth=1), f"soup {seed}, area lights from inside")
    # the frame in the interleaved tiles of two and of five ranks: together they are the frame
    import ctypes as C
    from raytracer_amd.engine import host_lib, _ptr
    tiled = dict(width=53, height=31, shadow_samples=4, glossy_rays=2, ray_depth=2)
    full = gpu.render(outside, soft, make_params(**tiled))
    for world, tile in ((2, 8), (5, 16)):
        acc = np.full((31, 53, 3), -7.0, np.float32)
        for rank in range(world):
            rp = make_params(**tiled, tile_size=tile, tile_rank=rank, tile_world=world)
            assert host_lib().rtxh_render(C.c_void_p(gpu.b), C.byref(outside), C.byref(soft), C.byref(rp), _ptr(acc), C.byref(gpu.stats)) == 0
        assert np.array_equal(acc.view(np.uint32), full.view(np.uint32)), f"soup {seed}: tiles of {world} ranks do not assemble the frame"
    host_lib().rtxh_set_sampling(4, 3, 7)
    try:
        got = gpu.final_color(loose[:400], soft, ray_depth=2)
    finally:
        host_lib().rtxh_set_sampling(50, 5, 1)
    want = orc.final_color(loose[:400], soft, make_params(1, 1, ray_depth=2, shadow_samples=4, glossy_rays=3, seed=7))
    assert_images_close(got, want, f"soup {seed}, getFinalColor on loose rays under area lights")
    gpu.close()
    orc.close()


@pytest.mark.parametrize("scale", [0.004, 40.0, 3000.0])
@pytest.mark.parametrize("seed", range(int(os.environ.get("RTX_FUZZ_SEEDS", "8"))))
def test_random_soups_at_other_scales(gpu_device, seed, scale):
    """The soups shrunk and blown up, camera and lights with them: the shadow queries' shortcuts
    hold only within stated coordinate ranges and must step aside outside them; the 1e-3 and 1e-5
    offsets of the reference's shadow test and bounce rays mean something else at each scale."""
    scene, rng = random_scene(seed)
    for m in scene.meshes:
        m.vertices[:, 0:3] *= np.float32(scale)
    scene.spheres = [(c * np.float32(scale), r * scale, mat) for c, r, mat in scene.spheres]
    lights = []
    for l in scene.lights:
        lights.append(abi.point_light(tuple(float(np.float32(v) * np.float32(scale)) for v in l.p[0]), tuple(l.c[0])))
    scene.lights = lights
    cam = random_camera(rng)
    cam.distance = float(np.float32(cam.distance) * np.float32(scale))
    cam.look_at[:] = [float(np.float32(v) * np.float32(scale)) for v in cam.look_at]
    mirror = abi.features(**dict(cases.BASE, recursive=1))
    gpu, orc = GpuEngine(scene, mirror), OracleEngine(scene, mirror)
    check_frame(gpu, orc, cam, mirror, make_params(64, 36, ray_depth=2), f"soup {seed} x {scale}")
    check_frame(gpu, orc, cam, mirror, make_params(48, 27, driver=abi.DRIVER_MRAYS, samples=4), f"soup {seed} x {scale}, 4 samples")
    check_frame(gpu, orc, cam, mirror, make_params(20, 12), f"soup {seed} x {scale}, 20x12")
    gpu.close()
    orc.close()
    soft = abi.features(**dict(cases.BASE, hard_shadow=0, soft_shadow=1, recursive=1, glossy_reflection=1))
    e = np.float32(scale)
    scene.lights = [abi.parallelogram_light((0.9 * e, -0.2 * e, -0.2 * e), (0, 0.4 * e, 0), (0, 0, 0.4 * e), (1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1)),
                    abi.segment_light((-0.5 * e, 1.0 * e, 0.2 * e), (0.5 * e, 1.1 * e, -0.2 * e), (1, 0.8, 0.6), (0.6, 0.8, 1))]
    gpu, orc = GpuEngine(scene, soft), OracleEngine(scene, soft)
    check_frame(gpu, orc, cam, soft, make_params(40, 22, shadow_samples=5, glossy_rays=2, ray_depth=2), f"soup {seed} x {scale}, area lights + glossy")
    gpu.close()
    orc.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTX_FUZZ_SEEDS", "24"))))
def test_random_soups_with_textures_and_an_environment_map(gpu_device, seed):
    """The soups again with a random texture per mesh (every filter) in front of a random
    environment map that mirrors reflect and that lights the scene."""
    scene, rng = random_scene(seed)
    for m in scene.meshes:
        # (square: the reference's chain of smaller images, src/main.cpp:745-771, reads outside the
        # image once one side of it reaches 1 before the other, and leaves a level of width 0)
        side = [4, 64, 256, 512][int(rng.integers(0, 4))]
        m.texture = (np.floor(rng.random((side, side, 3)) * 256) / 255.0).astype(np.float32)
        m.vertices[:, 6:8] = (rng.random((len(m.vertices), 2)) * 3 - 1).astype(np.float32) if rng.random() < 0.3 else m.vertices[:, 6:8]
    mapping = abi.ENV_SPHERICAL if rng.random() < 0.5 else abi.ENV_CYLINDRICAL
    scene.set_envmap((np.floor(rng.random((256, 512, 3)) * 256) / 255.0).astype(np.float32), mapping, 180.0 if mapping == abi.ENV_SPHERICAL else 60.0 + 90.0 * rng.random(),
                     tuple(rng.random(3)))
    outside, inside = random_camera(rng), random_camera(rng, distance=0.15)
    for filters in (dict(), dict(bilinear_texture_filtering=1), dict(bilinear_texture_filtering=1, mipmap_texture_filtering=1)):
        f = abi.features(**dict(cases.BASE, recursive=1, texture_mapping=1, environment_mapping=1, **filters))
        gpu, orc = GpuEngine(scene, f), OracleEngine(scene, f)
        check_frame(gpu, orc, outside, f, make_params(64, 36), f"soup {seed}, textures {filters}")
        check_frame(gpu, orc, inside, f, make_params(48, 27, driver=abi.DRIVER_MRAYS, samples=2), f"soup {seed} from inside, textures {filters}")
        gpu.close()
        orc.close()
    # the environment as a light (importance-sampled), with and without the point lights' own shadows
    f = abi.features(**dict(cases.BASE, soft_shadow=1, texture_mapping=1, environment_mapping=1))
    gpu, orc = GpuEngine(scene, f), OracleEngine(scene, f)
    check_frame(gpu, orc, outside, f, make_params(40, 22, shadow_samples=7), f"soup {seed}, environment light")
    gpu.close()
    orc.close()


def test_scenes_at_the_edge_of_being_scenes(gpu_device):
    """Nothing at all, spheres without a triangle, triangles without a light, a lone zero-area
    triangle, a single triangle seen edge-on: under the plain, multi-sample and transparency
    drivers, with mirrors, on both trees and without a tree."""
    def spheres_only():
        s = SceneData()
        s.add_sphere((0.1, 0.0, 0.2), 0.4, (0.8, 0.3, 0.2, 0.5, 0.5, 0.5, 10.0, 1.0))
        s.add_sphere((-0.5, 0.2, -0.1), 0.25, (0.2, 0.8, 0.3, 0.0, 0.0, 0.0, 1.0, 1.0))
        s.lights.append(abi.point_light((1.0, 2.0, 1.5)))
        return s

    def unlit():
        s, _ = random_scene(3)
        s.lights = []
        return s

    def one_triangle(corners):
        s = SceneData()
        v = np.zeros((3, 8), np.float32)
        v[:, 0:3] = corners
        v[:, 3:6] = (0, 0, 1)
        s.add_mesh(v, [[0, 1, 2]], (0.7, 0.7, 0.2, 0.3, 0.3, 0.3, 5.0, 1.0))
        s.lights.append(abi.point_light((0.5, 1.0, 2.0)))
        return s

    empty = SceneData()
    empty.lights.append(abi.point_light((0.0, 1.0, 0.0)))
    cam = make_camera()
    edge_on = make_camera(rotation_degrees=(0, 90, 0))
    scenes = (("empty", empty, cam), ("nothing, not even a light", SceneData(), cam), ("spheres only", spheres_only(), cam), ("no lights", unlit(), cam),
              ("zero-area triangle", one_triangle([(0, 0, 0), (0.5, 0.5, 0), (0.5, 0.5, 0)]), cam),
              ("a point for a triangle", one_triangle([(0.1, 0.1, 0.1)] * 3), cam),
              ("triangle edge-on", one_triangle([(-0.5, -0.5, 0), (0.5, -0.5, 0), (0, 0.5, 0)]), edge_on))
    for name, scene, camera in scenes:
        for feat in (dict(cases.BASE, recursive=1), dict(shading=1, hard_shadow=1, recursive=1, accel_structure=1), dict(shading=1, hard_shadow=1, normal_interp=1)):
            f = abi.features(**feat)
            gpu, orc = GpuEngine(scene, f), OracleEngine(scene, f)
            for rp in (make_params(40, 24), make_params(33, 17, driver=abi.DRIVER_MRAYS, samples=3), make_params(24, 24, driver=abi.DRIVER_TRANSPARENCY)):
                check_frame(gpu, orc, camera, f, rp, f"{name}, {feat}, driver {rp.driver}")
            rays = cases.ray_batch(scene, 1, n_random=300, grid=12)
            assert_hits_equal(gpu.intersect(rays, f), orc.intersect(rays, f), f"{name}, {feat}: rays")
            gpu.close()
            orc.close()
        if scene.n_triangles:
            f = abi.features(**cases.BASE)
            dev, orc = GpuEngine(scene, f, device_build=True), OracleEngine(scene, f)
            check_frame(dev, orc, camera, f, make_params(40, 24), f"{name}, device-built tree")
            dev.close()
            orc.close()


def test_device_trigonometry_is_the_host_librarys(gpu_device):
    """Where the path turns an angle into a discrete outcome it must use the host C library's
    results, as the reference does: the motion-blur driver's per-sample cameras (glm::quat(euler),
    src/render.cpp:203-247: sinf / cosf) and the environment map's texture coordinates
    (src/environment_mapping.cpp:130-207: atan2f, acosf, tanf).  The device's restatements of those
    routines must give libm's bits: on half-angles a camera can have (float patterns in steps over
    +-2 pi, densely around the multiples of pi / 4 where the routines change branch), up to the
    sine's fast range of +-120, on cosines in [-1, 1] and on direction components."""
    import ctypes as C
    import ctypes.util
    import torch
    from raytracer_amd import engine
    lib = engine.hip_lib()
    assert lib.rtx_init(0) == 0
    libm = C.CDLL(ctypes.util.find_library("m"))
    for fn in (libm.sinf, libm.cosf, libm.tanf, libm.acosf):
        fn.restype, fn.argtypes = C.c_float, [C.c_float]
    libm.atan2f.restype, libm.atan2f.argtypes = C.c_float, [C.c_float, C.c_float]
    rng = np.random.default_rng(11)
    top = np.float32(2 * np.pi).view(np.uint32)
    bits = np.arange(0, int(top), 19997, dtype=np.uint32)                     # positive floats up to 2 pi
    near = np.concatenate([(np.float32(k * np.pi / 4).view(np.uint32) + np.arange(-150, 151)).astype(np.uint32) for k in range(1, 40)])
    unit = np.concatenate([rng.random(30000) * 2 - 1, 1 - rng.random(3000) * 1e-3, rng.random(3000) * 1e-3 - 1, [1.0, -1.0, 0.5, -0.5, 1.0000001]])
    x = np.concatenate([bits.view(np.float32), near.view(np.float32), (rng.random(30000) * 120).astype(np.float32), unit.astype(np.float32),
                        np.array([0.0, 1e-30, 2.0 ** -12, np.nextafter(np.float32(2.0 ** -12), np.float32(0)), 119.99999], np.float32)])
    x = np.concatenate([x, -x])
    y = (rng.normal(size=len(x)) * np.abs(x) * 10.0 ** rng.integers(-3, 4, len(x))).astype(np.float32)
    y[::17] = 0.0
    y[5::19] = x[5::19]
    n = len(x)
    assert np.abs(x).max() < 120 and n > 150000
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = torch.empty(5 * n, dtype=torch.float32, device="cuda")
    assert lib.rtx_selfcheck_trig(C.c_void_p(dx.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_size_t(n), C.c_void_p(out.data_ptr())) == 0, lib.rtx_last_error()
    got = out.cpu().numpy().reshape(5, n)
    want = np.array([[libm.sinf(float(v)) for v in x], [libm.cosf(float(v)) for v in x], [libm.tanf(float(v)) for v in x],
                     [libm.acosf(float(v)) for v in x], [libm.atan2f(float(b), float(a)) for a, b in zip(x, y)]], np.float32)
    for row, name in enumerate(("sinf", "cosf", "tanf", "acosf", "atan2f")):
        g, w = got[row], want[row]
        bad = np.flatnonzero((g.view(np.uint32) != w.view(np.uint32)) & ~(np.isnan(g) & np.isnan(w)))
        assert len(bad) == 0, f"{name} differs for {len(bad)} arguments, first x = {x[bad[:4]]}, y = {y[bad[:4]]}"


def test_nearest_texel_of_a_large_environment_map(gpu_device):
    """Environment lookups without filtering in a 4096 x 2048 map of random texels: the frame must be
    the oracle's in every pixel, which it only is when the texture coordinates have the reference's
    bits (one lookup in ten thousand lands on a neighbouring texel otherwise)."""
    rng = np.random.default_rng(5)
    for mapping, vfov in ((abi.ENV_SPHERICAL, 180.0), (abi.ENV_CYLINDRICAL, 120.0)):
        scene = cases.load_scene("triangle")
        scene.set_envmap((np.floor(rng.random((2048, 4096, 3)) * 256) / 255.0).astype(np.float32), mapping, vfov, (0.5, 0.5, 0.5))
        f = abi.features(**dict(cases.BASE, environment_mapping=1))
        gpu, orc = GpuEngine(scene, f), OracleEngine(scene, f)
        for rot in ((20, 20, 0), (-40, 130, 0), (75, 250, 0)):
            cam = make_camera(rotation_degrees=rot)
            rp = make_params(512, 288)
            assert_images_close(gpu.render(cam, f, rp), orc.render(cam, f, rp), f"mapping {mapping}, camera {rot}")
        gpu.close()
        orc.close()


def test_pretest_of_origin_frames_at_its_limits(gpu_device):
    """The records made per ray origin (camera position, point lights) carry a conservative pretest per
    triangle; what it may skip must be work whose answer was "no hit".  Geometry chosen against it: a
    tessellated plane seen and lit at grazing angles (most rays pass close to an edge), a camera and a
    light IN the plane of triangles, a camera a hair above a triangle much larger than its distance, a
    fan of slivers around one vertex with the camera looking at that vertex, and the same at scales where
    the pretest's products would leave fp32's range and it must stand aside."""
    def grid_plane(n, size, wobble, rng):
        xs = np.linspace(-size, size, n + 1)
        p = np.array([(x, wobble * rng.standard_normal(), z) for z in xs for x in xs], np.float32)
        v = np.zeros((len(p), 8), np.float32)
        v[:, 0:3] = p
        v[:, 3:6] = (0, 1, 0)
        t = []
        for j in range(n):
            for i in range(n):
                a = j * (n + 1) + i
                t += [[a, a + 1, a + n + 2], [a, a + n + 2, a + n + 1]]
        return v, np.array(t, np.uint32)

    def sliver_fan(n, rng):
        ang = np.sort(rng.random(n) * 2 * np.pi)
        v = np.zeros((n * 3, 8), np.float32)
        for k in range(n):
            a0, a1 = ang[k], ang[k] + 1e-3 + 0.3 * rng.random() ** 4
            v[3 * k + 0, 0:3] = (0.05, 0.1, -0.02)
            v[3 * k + 1, 0:3] = (0.05 + np.cos(a0), 0.1 + 0.2 * rng.random(), -0.02 + np.sin(a0))
            v[3 * k + 2, 0:3] = (0.05 + np.cos(a1), 0.1 + 0.2 * rng.random(), -0.02 + np.sin(a1))
        v[:, 3:6] = (0, 1, 0)
        return v, np.arange(n * 3, dtype=np.uint32).reshape(n, 3)

    rng = np.random.default_rng(77)
    mat = (0.7, 0.6, 0.3, 0.2, 0.2, 0.2, 8.0, 1.0)
    flat, bumpy, fan, big = SceneData(), SceneData(), SceneData(), SceneData()
    flat.add_mesh(*grid_plane(24, 1.0, 0.0, rng), mat)
    bumpy.add_mesh(*grid_plane(24, 1.0, 0.004, rng), mat)
    fan.add_mesh(*sliver_fan(150, rng), mat)
    v = np.zeros((3, 8), np.float32)
    v[:, 0:3] = [(-40, 0, -40), (40, 0, -35), (0, 0, 45)]
    v[:, 3:6] = (0, 1, 0)
    big.add_mesh(v, [[0, 1, 2]], mat)
    big.add_mesh(*grid_plane(6, 0.3, 0.02, rng), (0.2, 0.5, 0.8, 0, 0, 0, 1.0, 1.0))
    for s in (flat, bumpy, fan, big):
        # a light well above, one grazing, one exactly in the plane y = 0 of the flat meshes
        s.lights = [abi.point_light((0.3, 1.5, 0.4)), abi.point_light((1.8, 0.02, -0.3), (0.5, 0.5, 0.9)), abi.point_light((0.2, 0.0, 0.1), (0.3, 0.9, 0.3))]
    cams = {
        "from above": make_camera(distance=2.5, rotation_degrees=(60, 30, 0)),
        "grazing": make_camera(distance=2.2, rotation_degrees=(1.5, 200, 0), look_at=(0, 0.0, 0)),
        "in the plane": make_camera(distance=1.7, rotation_degrees=(0, 75, 0), look_at=(0, 0.0, 0)),
        "a hair above": make_camera(distance=0.6, rotation_degrees=(0.4, 10, 0), look_at=(0.3, 0.0, 0.2)),
        "at the fan's vertex": make_camera(distance=0.8, rotation_degrees=(35, 140, 0), look_at=(0.05, 0.1, -0.02)),
    }
    f = abi.features(**cases.BASE)
    for sname, scene in (("flat grid", flat), ("bumpy grid", bumpy), ("sliver fan", fan), ("huge triangle", big)):
        for scale in (1.0, 1e-5, 3e6):
            sc = SceneData()
            for m in scene.meshes:
                vv = m.vertices.copy()
                vv[:, 0:3] *= np.float32(scale)
                sc.add_mesh(vv, m.triangles, m.material)
            sc.lights = [abi.point_light(tuple(float(np.float32(x) * np.float32(scale)) for x in l.p[0]), tuple(l.c[0])) for l in scene.lights]
            gpu, orc = GpuEngine(sc, f), OracleEngine(sc, f)
            for cname, cam0 in cams.items():
                cam = make_camera(fov_degrees=float(np.degrees(cam0.fovy)), distance=float(np.float32(cam0.distance) * np.float32(scale)),
                                  look_at=tuple(float(np.float32(x) * np.float32(scale)) for x in cam0.look_at), rotation_degrees=tuple(float(np.degrees(x)) for x in cam0.euler))
                what = f"{sname} x {scale}, {cname}"
                check_frame(gpu, orc, cam, f, make_params(160, 90, driver=abi.DRIVER_MRAYS, samples=4), what + ", 4 samples", ties=2)
                check_frame(gpu, orc, cam, f, make_params(96, 54), what + ", plain", ties=1)
            gpu.close()
            orc.close()
