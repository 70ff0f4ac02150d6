"""Host-side scene container (the reference's Scene, src/scene.h:31-37, as flat
numpy arrays), its .npz form, the default camera and seeded procedural meshes.
"""
import math

import numpy as np

from . import abi


class MeshData:
    """One Mesh (framework/include/framework/mesh.h:36-43): vertices are rows of
    (position, normal, texCoord), triangles index them, one Material per mesh
    as (kd, ks, shininess, transparency), and an optional kd texture (h, w, 3)."""

    def __init__(self, vertices, triangles, material, texture=None):
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 8)
        self.triangles = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        self.material = np.ascontiguousarray(material, np.float32).reshape(8)
        self.texture = None if texture is None else np.ascontiguousarray(texture, np.float32)
        if self.triangles.size and int(self.triangles.max()) >= len(self.vertices):
            raise ValueError("triangle index out of range")


class SceneData:
    def __init__(self):
        self.meshes = []
        self.lights = []          # abi.Light
        self.spheres = []         # (center(3), radius, material(8))
        self.envmap = None        # dict(image=(h,w,3) or None, mapping, vfov_degrees, background)

    @property
    def n_triangles(self):
        return sum(len(m.triangles) for m in self.meshes)

    def add_mesh(self, vertices, triangles, material=(1, 1, 1, 0, 0, 0, 1, 1), texture=None):
        self.meshes.append(MeshData(vertices, triangles, material, texture))
        return self

    def add_sphere(self, center, radius, material):
        self.spheres.append((np.asarray(center, np.float32), float(radius), np.asarray(material, np.float32)))
        return self

    def set_envmap(self, image, mapping=abi.ENV_SPHERICAL, vfov_degrees=180.0, background=(0, 0, 0)):
        self.envmap = dict(image=None if image is None else np.ascontiguousarray(image, np.float32),
                           mapping=int(mapping), vfov_degrees=float(vfov_degrees),
                           background=np.asarray(background, np.float32))
        return self

    # ---- .npz form (fixtures) ------------------------------------------------
    def to_arrays(self):
        d = {"n_meshes": np.int32(len(self.meshes))}
        for i, m in enumerate(self.meshes):
            d[f"m{i}_v"], d[f"m{i}_t"], d[f"m{i}_mat"] = m.vertices, m.triangles, m.material
            if m.texture is not None:
                d[f"m{i}_tex"] = m.texture
        lights = np.zeros((len(self.lights), 22), np.float32)
        for i, l in enumerate(self.lights):
            lights[i, 0] = l.type
            lights[i, 1:10] = np.array([list(r) for r in l.p], np.float32).ravel()
            lights[i, 10:22] = np.array([list(r) for r in l.c], np.float32).ravel()
        d["lights"] = lights
        d["spheres"] = np.array([np.concatenate([c, [r], m]) for c, r, m in self.spheres], np.float32).reshape(-1, 12)
        if self.envmap is not None:
            e = self.envmap
            d["env_meta"] = np.array([e["mapping"], e["vfov_degrees"], *e["background"]], np.float32)
            if e["image"] is not None:
                d["env_image"] = e["image"]
        return d

    @staticmethod
    def from_arrays(d):
        s = SceneData()
        for i in range(int(d["n_meshes"])):
            tex = d[f"m{i}_tex"] if f"m{i}_tex" in d else None
            s.add_mesh(d[f"m{i}_v"], d[f"m{i}_t"], d[f"m{i}_mat"], tex)
        for row in d["lights"]:
            s.lights.append(abi.make_light(int(row[0]), row[1:10].reshape(3, 3), row[10:22].reshape(4, 3)))
        for row in d["spheres"]:
            s.add_sphere(row[0:3], row[3], row[4:12])
        if "env_meta" in d:
            m = d["env_meta"]
            s.set_envmap(d["env_image"] if "env_image" in d else None, int(m[0]), float(m[1]), m[2:5])
        return s

    def save(self, path):
        np.savez_compressed(path, **self.to_arrays())

    @staticmethod
    def load(path):
        with np.load(path) as z:
            return SceneData.from_arrays({k: z[k] for k in z.files})


def make_camera(fov_degrees=50.0, distance=3.0, look_at=(0, 0, 0), rotation_degrees=(20, 20, 0)):
    """CameraConfig defaults (src/config.h:16-21) converted like src/main.cpp:566-567
    (glm::radians in fp32).  Derived fields are filled by the host library."""
    cam = abi.Camera()
    cam.look_at[:] = [float(x) for x in look_at]
    cam.euler[:] = [float(np.float32(x) * abi.GLM_RADIANS) for x in rotation_degrees]
    cam.distance = float(distance)
    cam.fovy = float(np.float32(fov_degrees) * abi.GLM_RADIANS)
    return cam


def make_params(width, height, driver=abi.DRIVER_PLAIN, **kw):
    rp = abi.RenderParams()
    rp.width, rp.height, rp.driver = int(width), int(height), int(driver)
    rp.ray_depth = -1
    rp.seed = 1
    for k, v in kw.items():
        if k == "prev_camera":
            rp.prev_camera = v
        else:
            if not hasattr(rp, k):
                raise AttributeError(k)
            setattr(rp, k, v)
    return rp


# ---- procedural geometry ------------------------------------------------------
def _vertex_normals(pos, tris):
    n = np.zeros_like(pos, dtype=np.float64)
    a, b, c = pos[tris[:, 0]], pos[tris[:, 1]], pos[tris[:, 2]]
    fn = np.cross(b - a, c - a)
    for k in range(3):
        np.add.at(n, tris[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1
    return (n / ln).astype(np.float32)


def _normalize_unit(pos):
    """centerAndScaleToUnitMesh (framework/src/mesh.cpp:151-176) in spirit: centre on
    the vertex mean and scale the farthest vertex to distance 1."""
    pos = pos - pos.mean(axis=0, keepdims=True)
    return (pos / np.linalg.norm(pos, axis=1).max()).astype(np.float32)


def icosphere(subdivisions):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(subdivisions):
        edges = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        n = len(f)
        ab, bc, ca = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        f = np.concatenate([np.stack([f[:, 0], ab, ca], 1), np.stack([f[:, 1], bc, ab], 1),
                            np.stack([f[:, 2], ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return v, f


def dragon_standin(seed=1, subdivisions=6, material=(0.8, 0.6, 0.4, 0.0, 0.0, 0.0, 1.0, 1.0)):
    """Seeded stand-in for data/dragon.obj, which is not part of the reference
    checkout: an icosphere (subdivisions=6 -> 81 920 triangles, the dragon has
    about 87 K) displaced by a few octaves of seeded sinusoids into a lumpy,
    self-occluding blob, then normalised to the unit ball like loadMesh(...,
    normalize=true).  Returns one MeshData."""
    rng = np.random.default_rng(seed)
    v, f = icosphere(subdivisions)
    disp = np.zeros(len(v))
    for octave in range(5):
        freq = 1.5 * 2.0 ** octave
        for _ in range(3):
            k = rng.normal(size=3)
            k *= freq / np.linalg.norm(k)
            disp += (0.28 / 2.0 ** octave) * np.sin(v @ k + rng.uniform(0, 2 * math.pi))
    pos = v * (1.0 + disp)[:, None]
    pos[:, 0] *= 1.4
    pos[:, 1] *= 0.8
    pos = _normalize_unit(pos)
    tris = f.astype(np.uint32)
    nrm = _vertex_normals(pos.astype(np.float64), tris)
    uv = np.stack([np.arctan2(v[:, 2], v[:, 0]) / (2 * math.pi) + 0.5, np.arccos(np.clip(v[:, 1], -1, 1)) / math.pi], 1)
    verts = np.concatenate([pos, nrm, uv.astype(np.float32)], axis=1).astype(np.float32)
    return MeshData(verts, tris, material)


def dragon_scene(seed=1, subdivisions=6):
    """SceneType::Dragon (src/scene.cpp:68-73) with the stand-in mesh: one point
    light at (-1, 1, -1), white."""
    s = SceneData()
    s.meshes.append(dragon_standin(seed, subdivisions))
    s.lights.append(abi.point_light((-1, 1, -1), (1, 1, 1)))
    return s


def _tube(centres, radii, around, wobble=None):
    """A closed tube: rings of `around` vertices about the polyline `centres` (parallel-transported
    frames, so the rings do not twist), radius per ring, end caps as fans.  -> (vertices, faces)"""
    centres = np.asarray(centres, np.float64)
    n = len(centres)
    tang = np.gradient(centres, axis=0)
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    u = np.cross(tang[0], [0.0, 0.0, 1.0])
    if np.linalg.norm(u) < 1e-6:
        u = np.cross(tang[0], [0.0, 1.0, 0.0])
    u /= np.linalg.norm(u)
    phi = np.linspace(0, 2 * math.pi, around, endpoint=False)
    verts = []
    for i in range(n):
        u = u - tang[i] * (u @ tang[i])
        u /= np.linalg.norm(u)
        w = np.cross(tang[i], u)
        r = radii[i] * (1.0 if wobble is None else wobble[i])
        verts.append(centres[i] + np.outer(np.cos(phi), u) * r + np.outer(np.sin(phi), w) * r)
    verts = np.concatenate(verts + [centres[:1], centres[-1:]])
    i, j = np.meshgrid(np.arange(n - 1), np.arange(around), indexing="ij")
    a, b = i * around + j, i * around + (j + 1) % around
    faces = np.concatenate([np.stack([a, b, a + around], 2).reshape(-1, 3), np.stack([b, b + around, a + around], 2).reshape(-1, 3)])
    k = np.arange(around)
    head, tail = n * around, n * around + 1
    caps = np.concatenate([np.stack([np.full(around, head), (k + 1) % around, k], 1),
                           np.stack([np.full(around, tail), (n - 1) * around + k, (n - 1) * around + (k + 1) % around], 1)])
    return verts, np.concatenate([faces, caps])


def winding_dragon(seed=1, material=(0.8, 0.6, 0.4, 0.0, 0.0, 0.0, 1.0, 1.0), detail=1.0):
    """A second, harder seeded stand-in for data/dragon.obj (about 87 K triangles at detail 1):
    a body that winds through space in an S (so rays cross it several times: depth complexity
    up to six, against two for the lumpy ball of dragon_standin), tapering into a long thin tail
    and neck, with a scaly surface, two thin curved wing membranes, four thin legs and a row of
    spikes along the back.  Normalised to the unit ball like loadMesh(..., normalize=true)."""
    rng = np.random.default_rng(seed)
    parts = []

    def add(v, f):
        parts.append((np.asarray(v, np.float64), np.asarray(f, np.int64)))

    rings, around = max(8, int(380 * detail)), max(6, int(96 * detail))
    s = np.linspace(0.0, 1.0, rings)
    turn = 2 * math.pi * 1.6 * s + 0.5
    spine = np.stack([0.62 * np.cos(turn) * (0.55 + 0.45 * s) + 0.10 * np.sin(9 * s), 1.05 * (s - 0.5) + 0.16 * np.sin(2 * math.pi * 2.2 * s),
                      0.62 * np.sin(turn) * (0.55 + 0.45 * s)], 1)
    radius = 0.02 + 0.21 * np.sin(math.pi * np.clip(s * 1.12 - 0.04, 0, 1)) ** 1.4
    radius[s > 0.86] += 0.05 * np.sin(math.pi * (s[s > 0.86] - 0.86) / 0.14)          # the head
    scales = 1.0 + 0.06 * np.sin(s * rings * 0.9) + 0.03 * rng.normal(size=rings)
    body_v, body_f = _tube(spine, radius, around, scales)
    ring_phase = np.tile(np.sin(np.arange(around) * 8 * 2 * math.pi / around), rings)
    body_v[:rings * around] += 0.012 * ring_phase[:, None] * (body_v[:rings * around] - np.repeat(spine, around, axis=0))
    add(body_v, body_f)
    # wings: thin closed shells, curved, attached behind the shoulders
    gu, gv = max(4, int(44 * detail)), max(3, int(26 * detail))
    for side in (-1.0, 1.0):
        a, b = np.meshgrid(np.linspace(0, 1, gu), np.linspace(0, 1, gv), indexing="ij")
        root = spine[int(rings * 0.62)]
        span = 0.75 * a
        px = root[0] - 0.15 + 0.55 * b - 0.25 * a * a
        py = root[1] + 0.10 + 0.45 * np.sin(a * 2.2) - 0.08 * np.sin(b * 9) * a
        pz = root[2] + side * (0.12 + span)
        top = np.stack([px, py + 0.006, pz], 2).reshape(-1, 3)
        bot = np.stack([px, py - 0.006, pz], 2).reshape(-1, 3)
        i, j = np.meshgrid(np.arange(gu - 1), np.arange(gv - 1), indexing="ij")
        q = (i * gv + j).reshape(-1)
        quad = np.concatenate([np.stack([q, q + gv, q + 1], 1), np.stack([q + 1, q + gv, q + gv + 1], 1)])
        add(top, quad if side > 0 else quad[:, ::-1])
        add(bot, quad[:, ::-1] if side > 0 else quad)
    # legs
    for at, side in ((0.40, -1.0), (0.40, 1.0), (0.68, -1.0), (0.68, 1.0)):
        hip = spine[int(rings * at)]
        tt = np.linspace(0, 1, max(4, int(30 * detail)))
        leg = np.stack([hip[0] + 0.10 * tt * tt, hip[1] - 0.45 * tt, hip[2] + side * (0.10 + 0.12 * np.sin(tt * 2.5))], 1)
        add(*_tube(leg, 0.045 * (1.0 - 0.75 * tt), max(5, int(16 * detail))))
    # spikes along the back
    for at in np.linspace(0.2, 0.84, max(3, int(24 * detail))):
        base = spine[int(rings * at)] + np.array([0.0, radius[int(rings * at)] * 0.9, 0.0])
        tt = np.linspace(0, 1, 6)
        spike = base + np.outer(tt, [-0.03, 0.13, 0.0])
        add(*_tube(spike, 0.022 * (1.0 - 0.95 * tt), max(4, int(10 * detail))))
    offset, vs, fs = 0, [], []
    for v, f in parts:
        vs.append(v)
        fs.append(f + offset)
        offset += len(v)
    pos = _normalize_unit(np.concatenate(vs))
    tris = np.concatenate(fs).astype(np.uint32)
    nrm = _vertex_normals(pos.astype(np.float64), tris)
    uv = np.stack([pos[:, 0] * 0.5 + 0.5, pos[:, 2] * 0.5 + 0.5], 1)
    verts = np.concatenate([pos, nrm, uv.astype(np.float32)], axis=1).astype(np.float32)
    return MeshData(verts, tris, material)


def winding_dragon_scene(seed=1, detail=1.0):
    """SceneType::Dragon (src/scene.cpp:68-73) with the winding stand-in mesh."""
    s = SceneData()
    s.meshes.append(winding_dragon(seed, detail=detail))
    s.lights.append(abi.point_light((-1, 1, -1), (1, 1, 1)))
    return s


def instanced_scene(copies, seed=1, subdivisions=6, transparent_every=3):
    """BASELINE configs[4] stand-in: `copies` instances of the seeded mesh baked to
    flat triangles on a jittered cubic grid inside the unit cube (122 copies =
    9 994 240 triangles), every `transparent_every`-th one with transparency 0.6,
    one point light."""
    base = dragon_standin(seed, subdivisions)
    rng = np.random.default_rng(seed)
    s = SceneData()
    g = int(np.ceil(copies ** (1 / 3)))
    for i in range(copies):
        v = base.vertices.copy()
        c = np.array([i % g, (i // g) % g, i // (g * g)], np.float32) / np.float32(g) - np.float32(0.5) + np.float32(0.5 / g)
        c = c + rng.uniform(-0.02, 0.02, 3).astype(np.float32)
        v[:, :3] = v[:, :3] * np.float32(0.8 / g) + c
        alpha = 0.6 if transparent_every and i % transparent_every == 0 else 1.0
        s.add_mesh(v, base.triangles.copy(), (0.8, 0.6, 0.4, 0.0, 0.0, 0.0, 1.0, alpha))
    s.lights.append(abi.point_light((-1, 1, -1), (1, 1, 1)))
    return s
