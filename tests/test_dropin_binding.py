"""The reference-side binding (integration/bvh_interface.cpp, integration/render.cpp): the
drop-in bodies of the reference's src/bvh_interface.cpp and src/render.cpp.

* CPU (build container, where /root/reference exists): the two files compile against the
  reference's own unmodified headers (src/bvh_interface.h:12-41, src/render.h:18-29,
  src/scene.h, src/screen.h, framework/trackball.h) and link, together with a caller that
  uses nothing but those headers (integration/dropin_check.cpp, written like
  src/main.cpp:50-63,159,226), against the reference's remaining objects and librtx_hip.so.
* GPU: that caller renders the reference's golden frames / ray batches through
  BvhInterface, renderRayTracing*, BvhInterface::intersect and getFinalColor.  The binary is
  built beforehand (oracle/Makefile, target dropin) and travels with the tree.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

import cases
from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "oracle", "_ref", "dropin_check")
REFERENCE = "/root/reference"


def test_binding_compiles_and_links_against_the_reference_headers():
    if not os.path.isdir(os.path.join(REFERENCE, "src")):
        pytest.skip("the reference checkout exists only in the build container")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "dropin"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert os.access(BINARY, os.X_OK)
    # the drop-in objects define the reference's seam and call the C ABI; nothing of the
    # reference's own tree or frame drivers is linked
    syms = subprocess.run(["nm", "-C", BINARY], capture_output=True, text=True).stdout
    for defined in ("BvhInterface::intersect(Ray&, HitInfo&, Features const&) const", "renderRayTracingMotionBlur(", "getFinalColor("):
        assert any(defined in l and " T " in l for l in syms.splitlines()), defined
    for wanted in ("rtx_render", "rtx_trace", "rtx_shade", "rtx_scene_create"):
        assert any(l.strip().startswith("U " + wanted) for l in syms.splitlines()), wanted
    assert "bvhSplitHelper" not in syms and "BoundingVolumeHierarchy::intersect" not in syms


# ---- files the caller reads ---------------------------------------------------------
def write_scene(scene, path):
    with open(path, "wb") as fh:
        fh.write(struct.pack("<i", len(scene.meshes)))
        for m in scene.meshes:
            fh.write(struct.pack("<ii", len(m.vertices), len(m.triangles)))
            fh.write(m.vertices.tobytes() + m.triangles.tobytes() + m.material.tobytes())
        fh.write(struct.pack("<i", len(scene.lights)))
        for l in scene.lights:
            row = np.zeros(22, np.float32)
            row[0] = l.type
            row[1:10] = np.array([list(r) for r in l.p], np.float32).ravel()
            row[10:22] = np.array([list(r) for r in l.c], np.float32).ravel()
            fh.write(row.tobytes())
        fh.write(struct.pack("<i", len(scene.spheres)))
        for c, r, m in scene.spheres:
            fh.write(np.concatenate([c, [r], m]).astype(np.float32).tobytes())


def write_bmp(image, path):
    """(h, w, 3) floats that are multiples of 1/255 -> 24-bit BMP (what Image, framework/src/image.cpp:22-32,
    reads back to the same floats)."""
    u8 = np.rint(image * 255.0).astype(np.uint8)
    assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), image), "texels are not 8-bit values"
    h, w, _ = u8.shape
    pad = (-3 * w) % 4
    rows = b"".join(u8[y, :, ::-1].tobytes() + b"\0" * pad for y in range(h - 1, -1, -1))
    with open(path, "wb") as fh:
        fh.write(b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54))
        fh.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(rows), 2835, 2835, 0, 0))
        fh.write(rows)


def hexf(x):
    return float(np.float32(x)).hex()


def pose(cam):
    return " ".join(hexf(v) for v in (cam.fovy, cam.distance, *cam.look_at, *cam.euler))


def job_lines(scene, f, tmp_path):
    write_scene(scene, tmp_path / "scene.bin")
    lines = [f"scene {tmp_path / 'scene.bin'}", f"out {tmp_path / 'out.bin'}",
             "features " + " ".join(str(getattr(f, n)) for n, _ in abi.Features._fields_[:12])]
    for i, m in enumerate(scene.meshes):
        if m.texture is not None:
            write_bmp(m.texture, tmp_path / f"tex{i}.bmp")
            lines.append(f"texture {i} {tmp_path / f'tex{i}.bmp'}")
    if scene.envmap is not None:
        e = scene.envmap
        name = "-"
        if e["image"] is not None:
            name = str(tmp_path / "env.bmp")
            write_bmp(e["image"], name)
        lines.append(f"env {name} {e['mapping']} {hexf(e['vfov_degrees'])} " + " ".join(hexf(v) for v in e["background"]))
    return lines


def run_job(lines, tmp_path):
    (tmp_path / "job.txt").write_text("\n".join(lines) + "\n")
    r = subprocess.run([BINARY, str(tmp_path / "job.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = (tmp_path / "out.bin").read_bytes()
    return np.frombuffer(raw[:8], np.int32), np.frombuffer(raw[8:], np.float32)


def need_binary():
    if not os.access(BINARY, os.X_OK):
        pytest.skip("oracle/_ref/dropin_check is built only where the reference checkout exists (make -C oracle dropin)")


FRAME_CASES = ["cfg1_cornell", "cornell_median_flat", "cornell_albedo", "cornell_bruteforce", "monkey_mrays4",
               "teapot_mrays_default64", "monkey_dof5", "transparency_cg3", "monkey_blur3", "spheres",
               "cornell_parallelogram_soft", "cube_tex_bilinear", "cube_tex_trilinear", "monkey_env_spherical",
               "monkey_env_cylindrical_bilinear"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", FRAME_CASES)
def test_reference_callers_render_the_golden_frames(gpu_device, golden_images, tmp_path, name):
    need_binary()
    from test_gpu_parity import RGB_TOL, assert_images_close
    case = cases.CASES_BY_NAME[name]
    scene, f, cam, rp = case.build()
    lines = job_lines(scene, f, tmp_path) + [
        "mode frame", f"size {rp.width} {rp.height}",
        f"driver {rp.driver} {rp.samples} {hexf(rp.aperture or 0.1)} {hexf(rp.focal_length or 2.0)}",
        f"globals {rp.shadow_samples or 50} {rp.glossy_rays or 5}", "camera " + pose(cam)]
    if rp.driver == abi.DRIVER_MOTION_BLUR:
        lines.append("prev " + pose(rp.prev_camera))
    shape, px = run_job(lines, tmp_path)
    got = px.reshape(rp.height, rp.width, 3)
    assert_images_close(got, golden_images[name], name + " through the reference's API", max(RGB_TOL, case.tol))


@pytest.mark.gpu
def test_reference_callers_intersect_and_get_final_color(gpu_device, golden_rays, golden_shade, golden_bvh, tmp_path):
    need_binary()
    from test_gpu_parity import assert_images_close
    # BvhInterface::intersect, ray by ray, against the reference's own answers
    scene = cases.load_scene("cornell")
    f = abi.features(accel_structure=1, bvh_sah_binning=1, normal_interp=1)
    rays = golden_rays["cornell_rays"][::9].copy()
    want = golden_rays["cornell_interp_hits"][::9]
    rays.tofile(tmp_path / "rays.bin")
    shape, rec = run_job(job_lines(scene, f, tmp_path) + ["mode trace", f"rays {tmp_path / 'rays.bin'}"], tmp_path)
    rec = rec.reshape(len(rays), 16)
    assert np.array_equal(rec[:, 0] != 0, want["hit"] != 0)
    hit = want["hit"] != 0
    assert np.array_equal(rec[hit, 1].view(np.uint32), want["t"][hit].view(np.uint32))
    assert np.array_equal(rec[~hit, 1].view(np.uint32), rays["t"][~hit].view(np.uint32)), "a miss leaves ray.t alone"
    assert np.array_equal(rec[hit, 2:5].view(np.uint32), want["normal"][hit].view(np.uint32))
    assert np.array_equal(rec[hit, 5:8].view(np.uint32), want["bary"][hit].view(np.uint32))
    assert np.array_equal(rec[hit, 8:10].view(np.uint32), want["uv"][hit].view(np.uint32))
    # HitInfo::mesh / triangle / material are the caller's own Scene objects
    for i in np.flatnonzero(hit)[:200]:
        mesh = scene.meshes[int(rec[i, 13])]
        assert tuple(int(v) for v in rec[i, 10:13]) in {tuple(int(x) for x in t) for t in mesh.triangles}
        assert rec[i, 14] == mesh.material[0] and rec[i, 15] == mesh.material[6]
    assert tuple(shape) == tuple(int(v) for v in golden_bvh["cornell_sweep_meta"]), "numLevels / numLeaves of the reference's tree"
    # getFinalColor(scene, bvh, ray, features, rayDepth) ray by ray
    name, scene_name, feat, params, lights = cases.SHADE_CASES[1]
    scene = cases.load_scene(scene_name)
    f = abi.features(**feat)
    srays = golden_shade["rays"][::7].copy()
    srays.tofile(tmp_path / "srays.bin")
    _, rgb = run_job(job_lines(scene, f, tmp_path) + ["mode shade", "depth 0", f"rays {tmp_path / 'srays.bin'}"], tmp_path)
    assert_images_close(rgb.reshape(-1, 3), golden_shade[name][::7], "getFinalColor through the reference's API")
