#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ by running the REFERENCE's own
translation units (oracle/_ref/libref.so, built by `make -C oracle ref` from
/root/reference) on the cases of tests/cases.py.

Run in the build container only (it needs /root/reference):

    make -C oracle ref && python tests/golden/make_fixtures.py

Outputs (all data, no reference source):
    scenes/<name>.npz   the Scene exactly as the reference's loader leaves it
                        (vertex de-duplication, per-material split, unit
                        normalisation applied), lights and spheres included
    bvh.npz             per scene x {median, sweep}: levels, leaves, leaf triangle
                        lists and node boxes in depth-first order
    rays.npz            per scene: a ray batch and the reference's HitInfo for it
    images.npz          per image case: the reference's frame (fp32 RGB)
    shade.npz           getFinalColor on a batch of caller-supplied rays
    config.npz          readConfigFile's result for the TOML files of toml/
    kat.npz             per-function known answers (slab, triangle, sphere,
                        barycentric, shading, reflection, textures, mip chain,
                        environment map, light sampling, glossy fan, RNG stream)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracer_amd import abi  # noqa: E402
from raytracer_amd.scene import make_camera  # noqa: E402
from oracle.bindings import RefEngine, have_ref  # noqa: E402
import cases  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def walk(d):
    leaves, boxes = [], []

    def rec(i):
        boxes.append(d["aabb"][i])
        ch = d["children"][d["offsets"][i]:d["offsets"][i + 1]]
        if d["leaf"][i]:
            leaves.append(np.array(ch, np.int32))
            return
        rec(int(ch[0]))
        rec(int(ch[1]))

    if d["root"] >= 0:
        sys.setrecursionlimit(100000)
        rec(d["root"])
    sizes = np.array([len(l) for l in leaves], np.int32)
    flat = np.concatenate(leaves).astype(np.int32) if leaves else np.zeros(0, np.int32)
    return sizes, flat, np.array(boxes, np.float32).reshape(-1, 6)


def make_scenes():
    os.makedirs(os.path.join(OUT, "scenes"), exist_ok=True)
    f = abi.features(accel_structure=1)
    for name, st in cases.SCENES.items():
        ref = RefEngine(st, f)
        ref.export_scene().save(os.path.join(OUT, "scenes", name + ".npz"))
        ref.close()
        print("scene", name)


def make_bvh():
    out = {}
    for name in cases.SCENES:
        scene = cases.load_scene(name)
        for sah in (0, 1):
            f = abi.features(accel_structure=1, bvh_sah_binning=sah)
            ref = RefEngine(scene, f)
            sizes, flat, boxes = walk(ref.dump())
            key = f"{name}_{'sweep' if sah else 'median'}"
            out[key + "_meta"] = np.array([ref.levels(), ref.leaves()], np.int32)
            out[key + "_leaf_sizes"], out[key + "_leaf_tris"], out[key + "_boxes"] = sizes, flat, boxes
            ref.close()
        print("bvh", name)
    np.savez_compressed(os.path.join(OUT, "bvh.npz"), **out)


def make_rays():
    out = {}
    for i, name in enumerate(cases.SCENES):
        scene = cases.load_scene(name)
        rays = cases.ray_batch(scene, seed=100 + i)
        for tag, feat in (("interp", dict(accel_structure=1, bvh_sah_binning=1, normal_interp=1)),
                          ("flat_median", dict(accel_structure=1)),
                          ("brute", dict(normal_interp=1))):
            if tag == "brute" and scene.n_triangles > 2000:
                continue
            f = abi.features(**feat)
            ref = RefEngine(scene, f)
            out[f"{name}_{tag}_hits"] = ref.intersect(rays, f)
            ref.close()
        out[f"{name}_rays"] = rays
        print("rays", name, len(rays))
    np.savez_compressed(os.path.join(OUT, "rays.npz"), **out)


def make_images():
    out = {}
    for c in cases.IMAGE_CASES:
        scene, f, cam, rp = c.build()
        ref = RefEngine(scene, f)
        out[c.name] = ref.render(cam, f, rp, mode=c.ref_mode)
        ref.close()
        print("image", c.name, out[c.name].shape, float(np.nansum(out[c.name])))
    np.savez_compressed(os.path.join(OUT, "images.npz"), **out)


def make_shade():
    out = {"rays": cases.shade_rays()}
    for name, scene_name, feat, params, lights in cases.SHADE_CASES:
        scene = cases.load_scene(scene_name)
        if lights is not None:
            scene.lights = list(lights)
        f = abi.features(**feat)
        ref = RefEngine(scene, f)
        from raytracer_amd.scene import make_params
        out[name] = ref.final_color(out["rays"], f, make_params(**params))
        ref.close()
        print("shade", name, float(np.nansum(out[name])))
    np.savez_compressed(os.path.join(OUT, "shade.npz"), **out)


def _png(path, pixels, color_type, palette=None, seed=0, depth=8):
    """A PNG file written by hand (zlib + per-row filters chosen at random, so the
    reader's unfiltering is exercised).  pixels: (h, w, channels) sample values."""
    import struct
    import zlib
    h, w, ch = pixels.shape
    rng = np.random.default_rng(seed)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)

    samples = pixels.reshape(h, w * ch).astype(int)
    if depth == 8:
        flat = samples
    elif depth == 16:
        flat = np.stack([samples >> 8, samples & 255], 2).reshape(h, -1)
    else:
        per = 8 // depth
        padded = np.zeros((h, (w * ch + per - 1) // per * per), int)
        padded[:, :w * ch] = samples
        flat = sum(padded[:, k::per] << (8 - depth * (k + 1)) for k in range(per))
    unit = max(1, ch * depth // 8)
    raw = bytearray()
    for y in range(h):
        ft = int(rng.integers(0, 5))
        raw.append(ft)
        for i in range(flat.shape[1]):
            a = flat[y, i - unit] if i >= unit else 0
            b = flat[y - 1, i] if y else 0
            c = flat[y - 1, i - unit] if (y and i >= unit) else 0
            pred = [0, a, b, (a + b) // 2, paeth(a, b, c)][ft]
            raw.append((flat[y, i] - pred) & 255)
    body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if palette is not None:
        body += chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).ravel()))
    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2
    body += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")
    with open(path, "wb") as fh:
        fh.write(body)


OBJ_CASES = [("mixed", False), ("mixed", True), ("plain", True), ("plain", False)]


def make_obj():
    """Synthetic OBJ / MTL / PNG files (written here, seeded) and what the reference's
    loader (tinyobjloader + framework/src/mesh.cpp + stb_image) makes of them."""
    d = os.path.join(OUT, "obj")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(21)
    _png(os.path.join(d, "rgb.png"), rng.integers(0, 256, (9, 14, 3), dtype=np.uint8), 2, seed=1)
    _png(os.path.join(d, "rgba.png"), rng.integers(0, 256, (8, 8, 4), dtype=np.uint8), 6, seed=2)
    _png(os.path.join(d, "grey_alpha.png"), rng.integers(0, 256, (5, 7, 2), dtype=np.uint8), 4, seed=3)
    _png(os.path.join(d, "grey.png"), rng.integers(0, 256, (6, 6, 1), dtype=np.uint8), 0, seed=4)
    _png(os.path.join(d, "palette.png"), rng.integers(0, 16, (7, 5, 1), dtype=np.uint8), 3, palette=rng.integers(0, 256, (16, 3)), seed=5)
    _png(os.path.join(d, "palette1.png"), rng.integers(0, 2, (9, 11, 1)), 3, palette=rng.integers(0, 256, (2, 3)), seed=6, depth=1)
    _png(os.path.join(d, "palette4.png"), rng.integers(0, 16, (5, 7, 1)), 3, palette=rng.integers(0, 256, (16, 3)), seed=7, depth=4)
    _png(os.path.join(d, "grey2.png"), rng.integers(0, 4, (6, 9, 1)), 0, seed=8, depth=2)
    _png(os.path.join(d, "grey4.png"), rng.integers(0, 16, (4, 5, 1)), 0, seed=9, depth=4)
    _png(os.path.join(d, "rgb16.png"), rng.integers(0, 65536, (5, 6, 3)), 2, seed=10, depth=16)
    _png(os.path.join(d, "grey16.png"), rng.integers(0, 65536, (4, 4, 1)), 0, seed=11, depth=16)
    with open(os.path.join(d, "mixed.mtl"), "w") as fh:
        fh.write("# materials of mixed.obj\n"
                 "newmtl red\nKd 0.8 0.1 0.05\nKs 0.5 0.5 0.25\nNs 96.078431\nd 0.7\n\n"
                 "newmtl textured\nKa 0 0 0\nmap_Kd rgb.png\nKs .125 1e-1 2.5E-1\nTr 0.25\n"
                 "newmtl late_kd\nmap_Kd palette.png\nKd 0.2 0.3 0.4\nNs 1.5e1\nTr 0.1\nd 0.9\n"
                 "newmtl grey\nKd 1 1 1\nmap_Kd grey_alpha.png\n")
    v = rng.uniform(-2, 2, (40, 3))
    lines = ["# seeded test mesh", "mtllib mixed.mtl", "o first"]
    fmts = ["%.6f", "%.3f", "%.9f", "%e", "%.2E", "%g"]
    for i, p in enumerate(v):
        lines.append("v " + " ".join(fmts[(i + k) % len(fmts)] % c for k, c in enumerate(p)))
    lines += ["v .5 -.25 +1.", "v 1e0 2E+0 -3e-0", "v 0.1234567890123 12345.678 -0.000001234"]      # 41, 42, 43 (1-based)
    for i in range(12):
        n = rng.normal(size=3)
        lines.append("vn %.5f %.5f %.5f" % tuple(n / np.linalg.norm(n)))
    for i in range(10):
        lines.append("vt %.4f %.4f" % tuple(rng.uniform(-0.5, 1.5, 2)))
    lines += ["usemtl red", "f 1/1/1 2/2/2 3/3/3", "f 2/2/2 3/3/3 4/4/4", "f 4//5 5//6 6//7", "f 7/3 8/4 9/5",
              "f 10 11 12", "f 1/1/1 5/6/2 9/3/3 13/4/4",                    # a quad
              "usemtl textured", "f 14/1/1 15/2/1 16/3/1 17/4/1", "f 18/5 19/6 20/7", "f -1 -2 -3", "f -4/-1/-1 -5/-2/-2 -6/-3/-3",
              "usemtl nosuchmaterial", "f 21 22 23", "f 22 23 24",
              "g second group", "f 25/8/9 26/9/10 27/10/11", "usemtl late_kd", "f 28 29 30", "f 29 30 31 32",
              "usemtl grey", "f 33 34 35",                                    # a material change on the last triangle of the shape
              "g third", "usemtl red", "f 36/1/1 37/2/2 38/3/3", "f 36/1/1 37/2/2 38/3/3", "f 38/3/3 39/4/4 40/5/5", "s off", "l 1 2"]
    with open(os.path.join(d, "mixed.obj"), "w") as fh:
        fh.write("\r\n".join(lines[:20]) + "\r\n" + "\n".join(lines[20:]) + "\n")
    # no normals, no materials, every face its own copy of the corner vertices -> de-duplication
    grid = [(x, y) for y in range(5) for x in range(6)]
    lines = ["# plain grid, bent"]
    for x, y in grid:
        lines.append("v %d %d %.3f" % (x, y, 0.3 * np.sin(x) * np.cos(y)))
    for y in range(4):
        for x in range(5):
            a, b, c, e = 1 + y * 6 + x, 2 + y * 6 + x, 8 + y * 6 + x, 7 + y * 6 + x
            lines.append("f %d %d %d %d" % (a, b, c, e) if (x + y) % 2 else "f %d %d %d" % (a, b, c))
            if not (x + y) % 2:
                lines.append("f %d %d %d" % (a, c, e))
    with open(os.path.join(d, "plain.obj"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    for name, norm in OBJ_CASES:
        s = RefEngine.load_obj(os.path.join(d, name + ".obj"), norm)
        s.save(os.path.join(d, f"{name}_{'normalized' if norm else 'raw'}.npz"))
        print("obj", name, norm, [(len(m.vertices), len(m.triangles)) for m in s.meshes])
    imgs = {}
    R = RefEngine.__new__(RefEngine)
    for png in ("rgb", "rgba", "grey_alpha", "grey", "palette", "palette1", "palette4", "grey2", "grey4", "rgb16", "grey16"):
        path = os.path.join(d, png + ".png").encode()
        w, h = C.c_int(), C.c_int()
        fn = R.fn("image_load")
        fn(path, C.byref(w), C.byref(h), None)
        out = np.zeros((h.value, w.value, 3), np.float32)
        fn(path, C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p))
        imgs[png] = out
    np.savez_compressed(os.path.join(d, "images.npz"), **imgs)


def make_kat():
    rng = np.random.default_rng(7)
    E = RefEngine.__new__(RefEngine)
    out = {}
    f32 = np.float32
    fmax = np.finfo(np.float32).max

    # slab test: random boxes/rays + origin inside + axis-parallel + on-face origins
    n = 400
    lo = rng.uniform(-1, 0, (n, 3)).astype(f32)
    hi = (lo + rng.uniform(0.01, 1.5, (n, 3))).astype(f32)
    rays = np.zeros(n, abi.RAY_DTYPE)
    rays["origin"] = rng.uniform(-2, 2, (n, 3))
    rays["direction"] = rng.normal(size=(n, 3))
    rays["t"] = fmax
    rays["origin"][:60] = (lo[:60] + hi[:60]) / 2          # origin inside
    rays["direction"][60:120, 0] = 0                       # axis-parallel
    rays["direction"][120:150, 1] = 0
    rays["origin"][150:170, 0] = lo[150:170, 0]            # origin on a face, parallel to it: 0 * inf
    rays["direction"][150:170, 0] = 0
    rays["t"][200:260] = rng.uniform(0, 2, 60)             # finite limit
    res = [E.kat_slab(lo[i], hi[i], rays[i]) for i in range(n)]
    out["slab_lo"], out["slab_hi"], out["slab_rays"] = lo, hi, rays
    out["slab_hit"] = np.array([r[0] for r in res], np.int32)
    out["slab_t"] = np.array([r[1] for r in res], f32)

    # triangle test: random, edge/vertex hits, degenerate, back-facing, behind, limited t
    n = 500
    a, b, c = (rng.uniform(-1, 1, (n, 3)).astype(f32) for _ in range(3))
    c[:20] = a[:20] + (b[:20] - a[:20]) * f32(0.5)        # degenerate (collinear)
    b[20:30] = a[20:30]                                     # degenerate (repeated vertex)
    rays = np.zeros(n, abi.RAY_DTYPE)
    w = rng.dirichlet((1, 1, 1), n).astype(f32)
    w[30:80] = rng.dirichlet((1, 1, 1), 50) * [1, 1, 0] / 1  # push towards an edge
    w[30:80] /= np.maximum(w[30:80].sum(1, keepdims=True), 1e-6)
    w[80:100] = [1, 0, 0]                                   # exactly a vertex
    target = (a * w[:, :1] + b * w[:, 1:2] + c * w[:, 2:3]).astype(f32)
    target[300:] = rng.uniform(-1.5, 1.5, (n - 300, 3))    # mostly misses
    rays["origin"] = rng.uniform(-3, 3, (n, 3))
    rays["direction"] = target - rays["origin"]
    rays["direction"][250:300] *= -1                        # triangle behind the origin
    rays["t"] = fmax
    rays["t"][100:160] = rng.uniform(0.2, 1.5, 60)
    res = [E.kat_triangle(a[i], b[i], c[i], rays[i]) for i in range(n)]
    out["tri_a"], out["tri_b"], out["tri_c"], out["tri_rays"] = a, b, c, rays
    out["tri_hit"] = np.array([r[0] for r in res], np.int32)
    out["tri_t"] = np.array([r[1] for r in res], f32)
    out["tri_bary"] = np.array([r[2] for r in res], f32)
    out["tri_n"] = np.array([r[3] for r in res], f32)

    # spheres
    n = 200
    cen = rng.uniform(-2, 2, (n, 3)).astype(f32)
    rad = rng.uniform(0.2, 1.5, n).astype(f32)
    rays = np.zeros(n, abi.RAY_DTYPE)
    rays["origin"] = rng.uniform(-3, 3, (n, 3))
    rays["origin"][:40] = cen[:40] + rng.uniform(-0.1, 0.1, (40, 3))   # inside
    rays["direction"] = cen + rng.normal(scale=0.7, size=(n, 3)) - rays["origin"]
    rays["t"] = fmax
    rays["t"][100:140] = rng.uniform(0.1, 2, 40)
    res = [E.kat_sphere(cen[i], float(rad[i]), rays[i]) for i in range(n)]
    out["sph_c"], out["sph_r"], out["sph_rays"] = cen, rad, rays
    out["sph_hit"] = np.array([r[0] for r in res], np.int32)
    out["sph_t"] = np.array([r[1] for r in res], f32)
    out["sph_n"] = np.array([r[2] for r in res], f32)

    # barycentric coordinates
    n = 200
    a, b, c = (rng.uniform(-1, 1, (n, 3)).astype(f32) for _ in range(3))
    w = rng.dirichlet((1, 1, 1), n).astype(f32)
    p = (a * w[:, :1] + b * w[:, 1:2] + c * w[:, 2:3]).astype(f32)
    out["bary_a"], out["bary_b"], out["bary_c"], out["bary_p"] = a, b, c, p
    out["bary_out"] = np.array([E.kat_barycentric(a[i], b[i], c[i], p[i]) for i in range(n)], f32)

    # Phong shading + reflection
    n = 300
    rays = np.zeros(n, abi.RAY_DTYPE)
    rays["origin"] = rng.uniform(-2, 2, (n, 3))
    rays["direction"] = rng.normal(size=(n, 3))
    rays["t"] = rng.uniform(0.5, 3, n)
    nrm = rng.normal(size=(n, 3)).astype(f32)
    lp = rng.uniform(-3, 3, (n, 3)).astype(f32)
    lc = rng.uniform(0, 2, (n, 3)).astype(f32)
    mat = np.zeros((n, 8), f32)
    mat[:, :3] = rng.uniform(0, 1, (n, 3))
    mat[:, 3:6] = rng.uniform(0, 1, (n, 3))
    mat[:, 6] = rng.uniform(1, 200, n)
    mat[:, 7] = 1
    out["shade_rays"], out["shade_n"], out["shade_lp"], out["shade_lc"], out["shade_mat"] = rays, nrm, lp, lc, mat
    out["shade_out"] = np.array([E.kat_shading(lp[i], lc[i], rays[i], nrm[i], mat[i]) for i in range(n)], f32)
    out["reflect_out"] = np.array([E.kat_reflect(rays[i], nrm[i]) for i in range(n)], abi.RAY_DTYPE)

    # textures: default.png of the reference's data set (128x128) and a seeded 16x8
    tex = E.image_load(os.path.join("/root/reference/data", "default.png"))
    small = (np.floor(rng.random((8, 16, 3)) * 256) / 255).astype(f32)
    uv = rng.uniform(-0.5, 1.5, (400, 2)).astype(f32)
    uv[:50] = rng.uniform(0, 1, (50, 2))
    uv[50:60] = [[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5], [1, 0.5], [0.5, 1], [0, 0.5], [0.999999, 0.999999], [1e-7, 1e-7]]
    out["tex_default"], out["tex_small"], out["tex_uv"] = tex, small, uv
    for nm, img in (("default", tex), ("small", small)):
        out[f"tex_{nm}_nearest"] = E.kat_texture(img, 0, uv)
        out[f"tex_{nm}_bilinear"] = E.kat_texture(img, 1, uv)
        sizes, levels = E.kat_mipmap(img)
        out[f"mip_{nm}_sizes"] = sizes
        for l, px in enumerate(levels):
            out[f"mip_{nm}_level{l}"] = px

    # environment map: both mappings, both filters, directions all around incl. the poles
    env = cases.env_image()
    dirs = rng.normal(size=(400, 3)).astype(f32)
    dirs[:6] = [[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]]
    out["env_image"], out["env_dirs"] = env, dirs
    for mapping, vfov, nm in ((abi.ENV_SPHERICAL, 180.0, "sph180"), (abi.ENV_SPHERICAL, 90.0, "sph90"), (abi.ENV_CYLINDRICAL, 120.0, "cyl120")):
        for bil in (0, 1):
            out[f"env_{nm}_{'bilinear' if bil else 'nearest'}"] = E.kat_envmap(env, mapping, vfov, (0.5, 0.25, 0.125), bil, dirs)

    # light sampling / glossy fan with the keyed RNG, and the driver-level engine's stream
    seg = abi.segment_light((1.5, 0.5, -0.6), (-1, 0.5, -0.5), (0.9, 0.2, 0.1), (0.2, 1, 0.3))
    par = abi.parallelogram_light((-0.2, 0.5, 0), (0.4, 0, 0), (0, 0, 0.4), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1))
    for nm, l in (("segment", seg), ("parallelogram", par)):
        pos, col = E.kat_light_samples(l, 1, 12345, 64)
        out[f"light_{nm}_pos"], out[f"light_{nm}_col"] = pos, col
    refl = np.zeros(1, abi.RAY_DTYPE)
    refl["origin"], refl["direction"], refl["t"] = (0.1, 0.2, 0.3), (0.3, 0.8, -0.5), fmax
    out["glossy_in"] = refl
    out["glossy_out"] = E.kat_glossy(refl[0], 0.1, 16, 1, 777)
    out["minstd_0"] = E.kat_minstd(0, 64)
    out["minstd_100000"] = E.kat_minstd(100000, 64)
    out["minstd_range"] = E.kat_minstd(7, 64, -0.05, 0.05)
    # environment light importance sampling: radiance bins, multi-jittered samples
    # (std::shuffle of this container's libstdc++ on the keyed engine), sampling rays
    sun = cases.sun_image()
    out["envlight_image"] = sun
    out["envlight_bins_sun"] = E.kat_env_bins(sun)
    out["envlight_bins_noise"] = E.kat_env_bins(env)
    rects = rng.uniform(0, 0.5, (12, 4)).astype(f32)
    rects[:, 2:] += 0.5
    out["sample2d_rects"] = rects
    for hk in (1, 2, 3, 5, 8):
        out[f"sample2d_{hk}"] = np.stack([E.kat_sample2d(rects[i], hk, hk, 11, i, 2 * i + 1) for i in range(len(rects))])
    pos = rng.uniform(-1, 1, (6, 3)).astype(f32)
    nrm = rng.normal(size=(6, 3)).astype(f32)
    out["envrays_pos"], out["envrays_nrm"] = pos, nrm
    for tag, mapping, vfov in (("sph180", abi.ENV_SPHERICAL, 180.0), ("cyl100", abi.ENV_CYLINDRICAL, 100.0)):
        for n in (1, 9, 50, 100):
            for i in range(len(pos)):
                out[f"envrays_{tag}_{n}_{i}"] = E.kat_env_rays(sun, mapping, vfov, pos[i], nrm[i], n, 5, i)
    np.savez_compressed(os.path.join(OUT, "kat.npz"), **out)
    print("kat", len(out), "arrays")


def make_config():
    """What the reference's readConfigFile (src/config.cpp:201-382) makes of the TOML files of
    tests/golden/toml/ (written for this build; the reference ships no configuration file)."""
    import glob
    import tempfile
    lib = RefEngine.lib()
    out = {}
    for path in sorted(glob.glob(os.path.join(OUT, "toml", "*.toml"))):
        name = os.path.splitext(os.path.basename(path))[0]
        text = open(path).read().replace("@DATA@", os.path.join(OUT, "obj_scene"))
        with tempfile.NamedTemporaryFile("w", suffix=".toml", delete=False) as fh:
            fh.write(text)
        values = (C.c_double * 4096)()
        scene = C.create_string_buffer(256)
        n = lib.ref_read_config(fh.name.encode(), values, 4096, scene, 256)
        os.unlink(fh.name)
        assert n > 0, path
        out[name] = np.array(values[:n], np.float64)
        out[name + "_scene"] = np.frombuffer(scene.value, np.uint8)
    np.savez_compressed(os.path.join(OUT, "config.npz"), **out)


def _ref_image(path):
    R = RefEngine.__new__(RefEngine)
    fn = R.fn("image_load")
    w, h = C.c_int(), C.c_int()
    fn(path.encode(), C.byref(w), C.byref(h), None)
    out = np.zeros((h.value, w.value, 3), np.float32)
    fn(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p))
    return out


def jpeg_cases():
    """(name, keyword arguments of jpeg_writer.write_jpeg): every sampling layout, scan layout and
    colour convention the reference's loader (stb_image) tells apart."""
    rng = np.random.default_rng(5)

    def smooth(h, w, n):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(x / 5 + c) * np.cos(y / 7 - c) + rng.normal(0, 12, (h, w)) for c in range(n)], 2)
        return np.clip(img, 0, 255).astype(np.uint8)

    return [
        ("grey", dict(samples=smooth(17, 33, 1))),
        ("c444", dict(samples=smooth(16, 24, 3))),
        ("c422", dict(samples=smooth(19, 35, 3), sampling=[(2, 1), (1, 1), (1, 1)])),
        ("c440", dict(samples=smooth(21, 18, 3), sampling=[(1, 2), (1, 1), (1, 1)])),
        ("c420_restart", dict(samples=smooth(37, 41, 3), sampling=[(2, 2), (1, 1), (1, 1)], restart=2)),
        ("c411", dict(samples=smooth(15, 50, 3), sampling=[(4, 1), (1, 1), (1, 1)])),
        ("c410", dict(samples=smooth(40, 40, 3), sampling=[(4, 2), (1, 1), (1, 1)])),
        ("one_pixel", dict(samples=smooth(1, 1, 3), sampling=[(2, 2), (1, 1), (1, 1)])),
        ("two_wide", dict(samples=smooth(9, 2, 3), sampling=[(2, 1), (1, 1), (1, 1)])),
        ("scan_per_component", dict(samples=smooth(20, 28, 3), sampling=[(2, 2), (1, 1), (1, 1)], interleaved=False, restart=3)),
        ("rgb_by_ids", dict(samples=smooth(12, 12, 3), ids=[ord("R"), ord("G"), ord("B")])),
        ("adobe_rgb", dict(samples=smooth(12, 12, 3), jfif=False, adobe=0)),
        ("adobe_rgb_but_jfif", dict(samples=smooth(12, 12, 3), jfif=True, adobe=0)),
        ("cmyk", dict(samples=smooth(13, 11, 4), jfif=False, adobe=0)),
        ("ycck", dict(samples=smooth(13, 11, 4), jfif=False, adobe=2, sampling=[(2, 2), (1, 1), (1, 1), (2, 2)])),
        ("four_components_no_adobe", dict(samples=smooth(10, 10, 4), jfif=False)),
        ("progressive_420", dict(samples=smooth(33, 47, 3), sampling=[(2, 2), (1, 1), (1, 1)], progressive=True)),
        ("progressive_444_restart", dict(samples=smooth(24, 24, 3), progressive=True, restart=2, filler=True)),
        ("progressive_grey", dict(samples=smooth(30, 19, 1), progressive=True)),
        ("progressive_scan_per_component", dict(samples=smooth(18, 27, 3), sampling=[(2, 1), (1, 1), (1, 1)], progressive=True, interleaved=False)),
        ("noise", dict(samples=rng.integers(0, 256, (24, 24, 3), dtype=np.uint8), quality_tables=[np.ones(64, int)] * 2)),
        ("progressive_noise", dict(samples=rng.integers(0, 256, (16, 16, 3), dtype=np.uint8), quality_tables=[np.ones(64, int) * 2] * 2,
                                   progressive=True, sampling=[(2, 2), (1, 1), (1, 1)])),
    ]


def make_jpeg():
    """JPEG files written by tests/golden/jpeg_writer.py (seeded) and what the reference's Image
    class (framework/src/image.cpp:13-36, stb_image) reads from them; plus a digest of what it
    reads from the one JPEG the reference ships (data/environment_map_cylindrical.jpg)."""
    import hashlib
    from jpeg_writer import write_jpeg
    d = os.path.join(OUT, "jpeg")
    os.makedirs(d, exist_ok=True)
    imgs = {}
    for name, kw in jpeg_cases():
        path = os.path.join(d, name + ".jpg")
        write_jpeg(path, seed=len(name), **kw)
        imgs[name] = _ref_image(path)
        print("jpeg", name, os.path.getsize(path), imgs[name].shape)
    own = _ref_image("/root/reference/data/environment_map_cylindrical.jpg")
    imgs["reference_environment_map_shape"] = np.array(own.shape)
    imgs["reference_environment_map_sha256"] = np.frombuffer(hashlib.sha256(own.tobytes()).digest(), np.uint8)
    imgs["reference_environment_map_rows"] = own[::37].copy()
    np.savez_compressed(os.path.join(d, "images.npz"), **imgs)


if __name__ == "__main__":
    if not have_ref():
        sys.exit("oracle/_ref/libref.so is missing: run `make -C oracle ref` where /root/reference exists")
    what = sys.argv[1:] or ["scenes", "bvh", "rays", "images", "shade", "kat", "obj", "config", "jpeg"]
    for w in what:
        globals()["make_" + w]()
