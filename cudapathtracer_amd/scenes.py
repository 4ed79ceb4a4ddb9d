"""Scene definitions beyond the reference's single built-in one.

The reference ships `models/bunny/bunny.obj` but no scene that uses it, and has
no transform support (a MeshTriangle is the OBJ's vertices verbatim,
common/Triangle.cpp:28-53).  The scenes here are therefore *defined* as triangle
soups in float32 (already transformed) and handed to ``Scene.Add`` -- the same
arrays can be given to any other implementation of the reference's
``MeshTriangle`` to build the identical scene.

* ``bunny_in_cornell``  Cornell walls + light, the two boxes replaced by the
  Stanford bunny (4968 triangles) standing on the floor.
* ``glass_cornell``  the reference's Cornell box with the short box made of the
  glass the reference defines for it but never switches on (common/Scene.cpp:45-47:
  roughness 0.2, ior 1.5, type left Lambertian): exercises ``Material::Type::Glass``.
* ``heightfield_in_cornell(n)``  Cornell walls + light over an n x n displaced
  height field (2 n^2 triangles; n = 708 gives ~1.0 M): a BVH-fetch stress scene.

``*_triangles`` functions return plain numpy data (no library needed), so tests
can build the same scene in the CPU oracle.
"""
import os

import numpy as np

from .api import DEFAULT_MODELS_DIR, Material, Scene

CORNELL_CAM = (278.0, 273.0, -800.0)

# materials of the reference's Cornell box (common/Scene.cpp:34-53)
RED = dict(type=0, albedo=(0.63, 0.065, 0.05))
GREEN = dict(type=0, albedo=(0.14, 0.45, 0.091))
WHITE = dict(type=0, albedo=(0.725, 0.71, 0.68))
METAL = dict(type=1, albedo=(0.7, 0.5, 0.2), metallic=1.0, roughness=0.005)
GLASS = dict(type=2, roughness=0.2, ior=1.5)


def light_emission():
    f = np.float32
    e = (f(8.0) * np.array([f(0.747) + f(0.058), f(0.747) + f(0.258), f(0.747)], np.float32)
         + f(15.6) * np.array([f(0.740) + f(0.287), f(0.740) + f(0.160), f(0.740)], np.float32)
         + f(18.4) * np.array([f(0.737) + f(0.642), f(0.737) + f(0.159), f(0.737)], np.float32))
    return tuple(float(v) for v in e.astype(np.float32))


def read_obj_triangles(path):
    """`v` / triangular `f` records -> float32 array (n, 3, 3), faces in file order."""
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([np.float32(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(x.split("/")[0]) for x in t[1:]]
                if len(idx) != 3:
                    raise ValueError("only triangular faces are supported: " + path)
                faces.append([i - 1 if i > 0 else len(verts) + i for i in idx])
    v = np.asarray(verts, np.float32)
    return v[np.asarray(faces, np.int64)]


def cornell_shell_triangles(models_dir=DEFAULT_MODELS_DIR):
    """[(name, triangles, material kwargs)] for floor/ceiling/back wall, left, right, light."""
    d = os.path.join(models_dir, "cornellbox")
    return [
        ("floor", read_obj_triangles(os.path.join(d, "floor.obj")), WHITE),
        ("left", read_obj_triangles(os.path.join(d, "left.obj")), RED),
        ("right", read_obj_triangles(os.path.join(d, "right.obj")), GREEN),
        ("light", read_obj_triangles(os.path.join(d, "light.obj")), dict(type=0, emission=light_emission())),
    ]


def bunny_triangles(models_dir=DEFAULT_MODELS_DIR, scale=2000.0, center=(278.0, 280.0)):
    """The bunny scaled by `scale`, its bounding box centred at (x, z) = `center`, feet on y = 0."""
    t = read_obj_triangles(os.path.join(models_dir, "bunny", "bunny.obj"))
    lo, hi = t.reshape(-1, 3).min(0), t.reshape(-1, 3).max(0)
    s = np.float32(scale)
    off = np.array([np.float32(center[0]) - s * (lo[0] + hi[0]) * np.float32(0.5), -s * lo[1],
                    np.float32(center[1]) - s * (lo[2] + hi[2]) * np.float32(0.5)], np.float32)
    return (t * s + off).astype(np.float32)


def heightfield_triangles(n, seed=1234, lo=(30.0, 30.0), hi=(526.0, 529.0), base=40.0, amplitude=60.0):
    """n x n cells over the floor rectangle [lo, hi] (x, z), heights base + amplitude * smooth noise;
    two upward-facing triangles per cell -> (2 n^2, 3, 3) float32."""
    rs = np.random.RandomState(seed)
    # smooth noise: a coarse random grid, bilinearly upsampled, plus a little per-vertex jitter
    coarse = rs.rand(9, 9)
    u = np.linspace(0, 8, n + 1)
    i0 = np.minimum(u.astype(int), 7)
    f = u - i0
    rows = coarse[i0][:, i0] * np.outer(1 - f, 1 - f) + coarse[i0 + 1][:, i0] * np.outer(f, 1 - f) \
        + coarse[i0][:, i0 + 1] * np.outer(1 - f, f) + coarse[i0 + 1][:, i0 + 1] * np.outer(f, f)
    h = base + amplitude * rows + (amplitude / n) * rs.rand(n + 1, n + 1)
    x = np.linspace(lo[0], hi[0], n + 1)
    z = np.linspace(lo[1], hi[1], n + 1)
    X, Z = np.meshgrid(x, z, indexing="ij")
    P = np.stack([X, h, Z], -1).astype(np.float32)  # (n+1, n+1, 3)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    # counter-clockwise seen from above (+y): normal = (v1-v0) x (v2-v0) points up
    t1 = np.stack([a, d, c], 2)
    t2 = np.stack([a, c, b], 2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)]).astype(np.float32)


def _assemble(parts, width, height, max_depth):
    sc = Scene(width, height, camPos=CORNELL_CAM, fov=40.0, maxDepth=max_depth)
    for _, tris, mat in parts:
        sc.Add(tris, Material(**mat))
    return sc.buildBVH()


def glass_cornell_parts(models_dir=DEFAULT_MODELS_DIR, glass=None):
    """The six meshes of the reference's Cornell box in Scene::Add order (common/Scene.cpp:63-68)."""
    d = os.path.join(models_dir, "cornellbox")
    shell = {name: (tris, mat) for name, tris, mat in cornell_shell_triangles(models_dir)}
    box = {n: read_obj_triangles(os.path.join(d, n + ".obj")) for n in ("shortbox", "tallbox")}
    return [("floor",) + shell["floor"], ("shortbox", box["shortbox"], dict(glass or GLASS)), ("tallbox", box["tallbox"], METAL),
            ("left",) + shell["left"], ("right",) + shell["right"], ("light",) + shell["light"]]


def bunny_in_cornell_parts(models_dir=DEFAULT_MODELS_DIR):
    return cornell_shell_triangles(models_dir) + [("bunny", bunny_triangles(models_dir), WHITE)]


def heightfield_in_cornell_parts(n, models_dir=DEFAULT_MODELS_DIR):
    return cornell_shell_triangles(models_dir) + [("heightfield", heightfield_triangles(n), WHITE)]


def glass_cornell(width=1024, height=1024, max_depth=20, models_dir=DEFAULT_MODELS_DIR):
    sc = _assemble(glass_cornell_parts(models_dir), width, height, max_depth)
    sc.name = "Glass Cornell"
    return sc


def bunny_in_cornell(width=1024, height=1024, max_depth=20, models_dir=DEFAULT_MODELS_DIR):
    sc = _assemble(bunny_in_cornell_parts(models_dir), width, height, max_depth)
    sc.name = "Bunny"
    return sc


def heightfield_in_cornell(n=708, width=2048, height=2048, max_depth=20, models_dir=DEFAULT_MODELS_DIR):
    sc = _assemble(heightfield_in_cornell_parts(n, models_dir), width, height, max_depth)
    sc.name = "Heightfield %d" % n
    return sc
