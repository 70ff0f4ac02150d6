"""CPU tests of the host side: the flattener builds the reference's tree and the
device records carry the reference's plane arithmetic; the C-ABI library loads,
exports every symbol include/rtx.h declares and lays its structs out as the
ctypes mirrors do; without a GPU the product path fails loudly instead of
falling back to a CPU implementation.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import cases
from raytracer_amd import abi, engine
from raytracer_amd.scene import SceneData, dragon_standin, make_camera
from oracle.bindings import OracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("scene_name", list(cases.SCENES))
def test_flattener_builds_the_reference_tree(golden_bvh, scene_name):
    scene = cases.load_scene(scene_name)
    for sah in (0, 1):
        f = abi.features(accel_structure=1, bvh_sah_binning=sah)
        flat = engine.flatten(scene, f)
        key = f"{scene_name}_{'sweep' if sah else 'median'}"
        assert [flat.num_levels, flat.num_leaves] == golden_bvh[key + "_meta"].tolist()
        leaves, boxes = flat.tree()
        assert [len(l) for l in leaves] == golden_bvh[key + "_leaf_sizes"].tolist()
        assert [t for l in leaves for t in l] == golden_bvh[key + "_leaf_tris"].tolist()
        assert np.array_equal(bits(boxes), bits(golden_bvh[key + "_boxes"]))
        # leaf-ordered records: every triangle exactly once, leaf ends flagged
        assert sorted(flat.tris["id"].tolist()) == list(range(scene.n_triangles))
        assert int((flat.tris["flags"] & 1).sum()) == flat.num_leaves
        flat.close()


def test_threaded_build_of_a_large_mesh_is_the_serial_tree():
    """Above 131072 triangles the flattener orders subtrees on several threads; the
    tree must still be the one the (serial) CPU restatement of the reference builds."""
    scene = SceneData()
    scene.meshes.append(dragon_standin(seed=3, subdivisions=7))     # 327 680 triangles
    for sah in (1, 0):
        f = abi.features(accel_structure=1, bvh_sah_binning=sah)
        flat = engine.flatten(scene, f)
        orc = OracleEngine(scene, f)
        assert [flat.num_levels, flat.num_leaves] == [orc.levels(), orc.leaves()]
        d = orc.dump()
        order = []

        def rec(i):
            ch = d["children"][d["offsets"][i]:d["offsets"][i + 1]]
            if d["leaf"][i]:
                order.extend(int(c) for c in ch)
                return
            rec(int(ch[0]))
            rec(int(ch[1]))

        sys.setrecursionlimit(100000)
        rec(d["root"])
        orc.close()
        assert flat.tris["id"].tolist() == order          # leaf order = depth-first order of the reference tree
        assert int((flat.tris["flags"] & 1).sum()) == flat.num_leaves
        flat.close()


def test_triangle_records_carry_the_reference_plane():
    """n and D of every record equal what intersectRayWithTriangle computes per test
    (src/intersect.cpp:50-58); degenerate triangles are encoded as NaN planes."""
    scene = cases.load_scene("monkey")
    # add a degenerate triangle (collinear) and a repeated-vertex one
    m = scene.meshes[0]
    nv = len(m.vertices)
    extra_v = np.zeros((3, 8), np.float32)
    extra_v[:, :3] = [[0, 0, 0], [0.5, 0.5, 0.5], [1, 1, 1]]
    scene.meshes[0] = type(m)(np.concatenate([m.vertices, extra_v]), np.concatenate([m.triangles, [[nv, nv + 1, nv + 2], [nv, nv, nv + 1]]]).astype(np.uint32), m.material)
    flat = engine.flatten(scene, abi.features(accel_structure=1))
    E = OracleEngine.__new__(OracleEngine)
    ray = np.zeros(1, abi.RAY_DTYPE)
    checked = 0
    for rec in flat.tris[:: max(1, len(flat.tris) // 200)]:
        c = (rec["v0"] + rec["v1"] + rec["v2"]) / 3
        ray["origin"], ray["direction"], ray["t"] = c + rec["n"], -rec["n"], np.finfo(np.float32).max
        hit, t, bary, n = E.kat_triangle(rec["v0"], rec["v1"], rec["v2"], ray[0])
        if hit:
            assert np.array_equal(bits(n), bits(rec["n"]))
            checked += 1
    assert checked > 100
    degenerate = flat.tris[flat.tris["id"] >= scene.n_triangles - 2]
    assert len(degenerate) == 2 and np.isnan(degenerate["n"]).all() and np.isnan(degenerate["d"]).all()
    flat.close()


def test_textures_are_flattened_with_their_mip_chains(golden_kat):
    scene = cases.load_scene("cube_textured")
    flat = engine.flatten(scene, abi.features(accel_structure=1))
    assert flat.n_textures == 1
    t = flat.textures[0]
    sizes = golden_kat["mip_default_sizes"]
    assert t.levels == len(sizes)
    for l in range(t.levels):
        assert [t.width[l], t.height[l]] == sizes[l].tolist()
        px = flat.texels[t.offset[l]: t.offset[l] + t.count[l]]
        assert np.array_equal(bits(px), bits(golden_kat[f"mip_default_level{l}"]))
    assert flat.materials["texture"].max() == 0
    flat.close()


def test_environment_radiance_bins_are_the_reference_bins(golden_kat):
    """EnvironmentMap's constructor + buildRadianceHierarchy (src/environment_mapping.cpp:42-105),
    restated in the host flattener, against the bins the reference built."""
    for image, key in ((golden_kat["envlight_image"], "envlight_bins_sun"), (golden_kat["env_image"], "envlight_bins_noise")):
        scene = cases.load_scene("triangle")
        scene.set_envmap(image, abi.ENV_SPHERICAL, 180.0, (0, 0, 0))
        fl = engine.flatten(scene, abi.features(accel_structure=1))
        d = fl.desc()
        bins = np.ctypeslib.as_array(C.cast(d.env_bins, C.POINTER(C.c_float)), shape=(d.n_env_bins, 4)).copy()
        fl.close()
        assert bins.shape == golden_kat[key].shape
        assert np.array_equal(bins.view(np.uint32), golden_kat[key].view(np.uint32))
    # a constant-colour environment has no bins
    scene = cases.load_scene("triangle")
    scene.set_envmap(None, abi.ENV_SPHERICAL, 180.0, (0.5, 0.5, 0.5))
    fl = engine.flatten(scene, abi.features(accel_structure=1))
    assert fl.desc().n_env_bins == 0
    fl.close()


@pytest.mark.parametrize("name,normalize", [("mixed", False), ("mixed", True), ("plain", True), ("plain", False)])
def test_obj_loader_matches_the_reference_loader(name, normalize):
    """rt::loadMesh against what the reference's loader (tinyobjloader +
    framework/src/mesh.cpp) made of the same files (tests/golden/make_fixtures.py):
    number formats, index forms, quads, groups, per-material split including its
    last-triangle quirk, vertex de-duplication order, materials, textures, unit scaling."""
    d = os.path.join(cases.GOLDEN, "obj")
    got = engine.load_obj(os.path.join(d, name + ".obj"), normalize)
    want = SceneData.load(os.path.join(d, f"{name}_{'normalized' if normalize else 'raw'}.npz"))
    assert len(got.meshes) == len(want.meshes)
    for i, (g, w) in enumerate(zip(got.meshes, want.meshes)):
        assert np.array_equal(g.triangles, w.triangles), i
        assert g.vertices.shape == w.vertices.shape and np.array_equal(bits(g.vertices), bits(w.vertices)), i
        assert np.array_equal(bits(g.material), bits(w.material)), i
        assert (g.texture is None) == (w.texture is None), i
        if g.texture is not None:
            assert g.texture.shape == w.texture.shape and np.array_equal(bits(g.texture), bits(w.texture)), i
    if name == "mixed":
        assert sum(m.texture is not None for m in got.meshes) >= 2


def test_png_reader_matches_the_reference_image_class():
    d = os.path.join(cases.GOLDEN, "obj")
    with np.load(os.path.join(d, "images.npz")) as z:
        for png in z.files:
            got = engine.load_image(os.path.join(d, png + ".png"))
            assert got.shape == z[png].shape and np.array_equal(bits(got), bits(z[png])), png
    with pytest.raises(RuntimeError):
        engine.load_image(os.path.join(d, "mixed.obj"))      # not a PNG
    with pytest.raises(RuntimeError):
        engine.load_obj(os.path.join(d, "no_such_file.obj"))


def test_jpeg_reader_matches_the_reference_image_class(tmp_path):
    """Baseline and progressive files of every sampling / scan / colour layout, against what the
    reference's loader (stb_image through framework/src/image.cpp:13-36) read from the same files."""
    import hashlib
    d = os.path.join(cases.GOLDEN, "jpeg")
    with np.load(os.path.join(d, "images.npz")) as z:
        names = [n for n in z.files if not n.startswith("reference_")]
        assert len(names) >= 22
        for name in names:
            got = engine.load_image(os.path.join(d, name + ".jpg"))
            assert got.shape == z[name].shape and np.array_equal(bits(got), bits(z[name])), name
        own = "/root/reference/data/environment_map_cylindrical.jpg"     # 2000 x 666, 4:2:0; only in the build container
        if os.path.exists(own):
            got = engine.load_image(own)
            assert tuple(z["reference_environment_map_shape"]) == got.shape
            assert np.array_equal(bits(got[::37]), bits(z["reference_environment_map_rows"]))
            assert hashlib.sha256(got.tobytes()).digest() == z["reference_environment_map_sha256"].tobytes()
    # what is not read is refused, never guessed at
    raw = open(os.path.join(d, "c420_restart.jpg"), "rb").read()
    sof = raw.index(b"\xff\xc0")
    for name, data in (("cut_short", raw[:len(raw) // 2]), ("arithmetic", raw[:sof] + b"\xff\xc9" + raw[sof + 2:]),
                       ("twelve_bit", raw[:sof + 4] + b"\x0c" + raw[sof + 5:]), ("no_frame", raw[:sof] + b"\xff\xd9")):
        (tmp_path / (name + ".jpg")).write_bytes(data)
        with pytest.raises(RuntimeError):
            engine.load_image(str(tmp_path / (name + ".jpg")))
    # a Huffman table that claims more codes of one length than the length can tell apart is refused
    # before its lookup table is filled (3 and 255 one-bit codes)
    dht = raw.index(b"\xff\xc4")
    for count in (3, 255):
        values = bytes(range(count))
        segment = b"\xff\xc4" + (2 + 1 + 16 + count).to_bytes(2, "big") + b"\x00" + bytes([count]) + bytes(15) + values
        (tmp_path / f"dht{count}.jpg").write_bytes(raw[:dht] + segment + raw[dht:])
        with pytest.raises(RuntimeError, match="Huffman"):
            engine.load_image(str(tmp_path / f"dht{count}.jpg"))


def test_bmp_reader(tmp_path):
    """24-bit bottom-up, 32-bit top-down and 8-bit palette bitmaps (what the reference reads through stb_image)."""
    import struct
    rng = np.random.default_rng(3)
    px = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)

    def header(bpp, h, extra=0, colours=0):
        stride = (7 * bpp // 8 + 3) & ~3
        return b"BM" + struct.pack("<IHHI", 54 + extra + stride * 5, 0, 0, 54 + extra) + struct.pack("<IiiHHIIiiII", 40, 7, h, 1, bpp, 0, stride * 5, 2835, 2835, colours, 0)

    rows24 = b"".join(px[y, :, ::-1].tobytes() + b"\0" * 3 for y in range(4, -1, -1))
    (tmp_path / "a.bmp").write_bytes(header(24, 5) + rows24)
    rows32 = b"".join(np.concatenate([px[y, :, ::-1], np.full((7, 1), 9, np.uint8)], 1).tobytes() for y in range(5))
    (tmp_path / "b.bmp").write_bytes(header(32, -5) + rows32)
    palette = rng.integers(0, 256, (16, 4), dtype=np.uint8)
    index = rng.integers(0, 16, (5, 7), dtype=np.uint8)
    (tmp_path / "c.bmp").write_bytes(header(8, 5, extra=64, colours=16) + palette.tobytes() + b"".join(index[y].tobytes() + b"\0" for y in range(4, -1, -1)))
    want = px.astype(np.float32) / np.float32(255)
    assert np.array_equal(engine.load_image(str(tmp_path / "a.bmp")), want)
    assert np.array_equal(engine.load_image(str(tmp_path / "b.bmp")), want)
    assert np.array_equal(engine.load_image(str(tmp_path / "c.bmp")), palette[index][:, :, 2::-1].astype(np.float32) / np.float32(255))
    (tmp_path / "d.bmp").write_bytes((header(24, 5) + rows24)[:80])
    with pytest.raises(RuntimeError):
        engine.load_image(str(tmp_path / "d.bmp"))
    # a header size or pixel offset that points outside the file (or into the header) is refused
    good = header(24, 5) + rows24
    for name, data in (("e", good[:14] + struct.pack("<I", 0x7fffff00) + good[18:]), ("f", good[:10] + struct.pack("<I", 20) + good[14:])):
        (tmp_path / (name + ".bmp")).write_bytes(data)
        with pytest.raises(RuntimeError, match="header size or pixel offset"):
            engine.load_image(str(tmp_path / (name + ".bmp")))


def test_empty_scene_and_single_triangle():
    s = SceneData()
    flat = engine.flatten(s, abi.features(accel_structure=1))
    assert flat.root_ref == -2 ** 31 and len(flat.nodes) == 0 and len(flat.tris) == 0 and flat.num_levels == 0
    flat.close()
    flat = engine.flatten(cases.load_scene("triangle"), abi.features(accel_structure=1))
    assert flat.root_ref == -1 and len(flat.nodes) == 0 and flat.num_levels == 1 and flat.num_leaves == 1  # root is a leaf
    flat.close()


def test_bad_triangle_index_is_rejected():
    s = SceneData()
    with pytest.raises(ValueError):
        s.add_mesh(np.zeros((3, 8), np.float32), [[0, 1, 3]])


def test_non_finite_vertices_are_rejected_by_the_flattener():
    """A NaN or infinite position (a damaged OBJ) would leave the sweep-split builder with NaN
    costs and an empty half; the host layer refuses the scene instead (the reference's behaviour
    is undefined there)."""
    for bad in (np.nan, np.inf):
        for sah in (0, 1):
            s = cases.load_scene("cube")
            s.meshes[0].vertices[2, 1] = bad
            with pytest.raises(RuntimeError, match="not finite"):
                engine.flatten(s, abi.features(accel_structure=1, bvh_sah_binning=sah))


def test_scene_npz_round_trip_and_seeded_standin(tmp_path):
    a = cases.load_scene("cornell_parallelogram")
    a.set_envmap(cases.env_image(), abi.ENV_CYLINDRICAL, 120.0, (0.1, 0.2, 0.3))
    a.add_sphere((1, 2, 3), 0.5, (1, 0, 0, 0, 0, 0, 1, 1))
    a.save(tmp_path / "s.npz")
    b = SceneData.load(tmp_path / "s.npz")
    for k, v in a.to_arrays().items():
        assert np.array_equal(v, b.to_arrays()[k]), k
    m1, m2 = dragon_standin(seed=1, subdivisions=3), dragon_standin(seed=1, subdivisions=3)
    assert np.array_equal(m1.vertices, m2.vertices) and len(m1.triangles) == 20 * 4 ** 3
    assert abs(np.linalg.norm(m1.vertices[:, :3], axis=1).max() - 1.0) < 1e-5
    assert not np.array_equal(m1.vertices, dragon_standin(seed=2, subdivisions=3).vertices)


# ---- the C ABI ---------------------------------------------------------------
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = engine.hip_lib()
    names = declared_symbols()
    assert {"rtx_init", "rtx_scene_create", "rtx_scene_destroy", "rtx_trace", "rtx_trace_device", "rtx_render", "rtx_render_device",
            "rtx_render_tiles_device", "rtx_detile_device", "rtx_tiles_per_rank", "rtx_set_stream", "rtx_synchronize", "rtx_last_error",
            "rtx_scene_retain"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/rtx.h but not exported"


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sizes.c"
    structs = ["rtx_features", "rtx_ray", "rtx_hit", "rtx_node", "rtx_tri", "rtx_tri_attr", "rtx_material", "rtx_texture", "rtx_light",
               "rtx_sphere", "rtx_envmap", "rtx_scene_desc", "rtx_camera", "rtx_render_params", "rtx_stats"]
    body = "\n".join(f'    printf("%zu\\n", sizeof({s}));' for s in structs)
    src.write_text('#include <stdio.h>\n#include "rtx.h"\nint main(void) {\n' + body + "\n    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    mirrors = [C.sizeof(abi.Features), abi.RAY_DTYPE.itemsize, abi.HIT_DTYPE.itemsize, abi.NODE_DTYPE.itemsize, abi.TRI_DTYPE.itemsize,
               abi.TRI_ATTR_DTYPE.itemsize, abi.MATERIAL_DTYPE.itemsize, C.sizeof(abi.Texture), C.sizeof(abi.Light), abi.SPHERE_DTYPE.itemsize,
               C.sizeof(abi.EnvMap), C.sizeof(abi.SceneDesc), C.sizeof(abi.Camera), C.sizeof(abi.RenderParams), C.sizeof(abi.Stats)]
    assert sizes == mirrors, list(zip(structs, sizes, mirrors))
    assert sizes[3] == sizes[4] == sizes[5] == 64   # node / triangle / attribute records are one 64-byte line each


def test_tile_partition_arithmetic():
    lib = engine.hip_lib()
    from raytracer_amd.scene import make_params
    from raytracer_amd.distributed import tiles_per_rank
    for (w, h, ts, world) in ((1920, 1080, 32, 8), (100, 70, 16, 3), (8, 8, 8, 2), (3840, 2160, 64, 4)):
        rp = make_params(w, h, tile_size=ts, tile_rank=0, tile_world=world)
        assert lib.rtx_tiles_per_rank(C.byref(rp)) == tiles_per_rank(w, h, ts, world)
    bad = make_params(64, 64, tile_size=12)
    assert lib.rtx_tiles_per_rank(C.byref(bad)) < 0 and b"multiple of 8" in lib.rtx_last_error()


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    """Runs in a child process so the GPU state of this one is untouched.  On a
    machine without a GPU rtx_init fails and so does every product entry point; on
    a machine with one this test only checks the not-initialised error path."""
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from raytracer_amd import abi, engine
import cases
lib = engine.hip_lib()
f = abi.features(accel_structure=1)
st = abi.Stats()
rc = lib.rtx_trace(None, None, 0, None, C.byref(f), C.byref(st))
assert rc == -4 and b"rtx_init" in lib.rtx_last_error(), (rc, lib.rtx_last_error())
rc = lib.rtx_init(0)
if rc != 0:
    assert lib.rtx_last_error() != b""
    try:
        engine.GpuEngine(cases.load_scene("cube"), f)
    except RuntimeError as e:
        print("LOUD", e)
    else:
        raise SystemExit("GpuEngine worked without a GPU: something fell back to the CPU")
else:
    print("GPU present")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "LOUD" in out.stdout or "GPU present" in out.stdout


@pytest.mark.parametrize("camera", [dict(), dict(distance=1.2, rotation_degrees=(35, -50, 10)), dict(fov_degrees=100.0, look_at=(0.3, -0.2, 0.1))])
def test_host_camera_rays_are_the_reference_rays(camera):
    """Trackball::generateRay of the host layer (framework/src/trackball.cpp:101-110) gives,
    bit for bit, the rays of the CPU restatement that the golden fixtures pin."""
    from oracle.bindings import primary_rays
    cam = make_camera(**camera)
    got, want = engine.camera_rays(cam, 96, 54, step=5), primary_rays(cam, 96, 54, step=5)
    assert got.dtype == want.dtype and len(got) == len(want) > 0
    assert got.tobytes() == want.tobytes()


def test_page_locked_allocation_falls_back_without_a_gpu():
    """rtx_host_alloc returns NULL (with an error text) where no page-locked memory can be
    had, and the host layer's Screen then keeps its pixels in plain memory: a Screen can be
    made, filled and written to a bitmap on a machine without a GPU (child process, so this
    process' GPU state is untouched)."""
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from raytracer_amd import engine
lib = engine.hip_lib()
lib.rtx_host_alloc.restype = C.c_void_p
lib.rtx_host_alloc.argtypes = [C.c_size_t]
lib.rtx_host_free.argtypes = [C.c_void_p]
p = lib.rtx_host_alloc(1 << 20)
if p:
    C.memset(p, 7, 1 << 20)
    lib.rtx_host_free(p)
    print("PINNED")
else:
    assert lib.rtx_last_error() != b""
    print("PLAIN")
lib.rtx_host_free(None)
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "PINNED" in out.stdout or "PLAIN" in out.stdout
