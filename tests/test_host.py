"""CPU-side tests of the product's host code: BVH topology and flattened layout
against the reference fixtures and the oracle, C-ABI libraries load and export
every declared symbol, PPM writer, error behaviour.  No GPU compute is called."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from orclib import primary_rays, random_rays, shadow_rays

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


@pytest.fixture(scope="module")
def cornell(hp):
    return hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5)


def test_libraries_export_every_declared_symbol(hp):
    hpt, host = hp.load_libraries()
    for header, lib in (("hpt.h", hpt), ("hpt_host.h", host)):
        text = open(os.path.join(ROOT, "include", header)).read()
        names = set(re.findall(r"\b(hpt_[a-z_0-9]+)\s*\(", text))
        assert len(names) >= 10
        for n in sorted(names):
            assert hasattr(lib, n), "%s declares %s but the library does not export it" % (header, n)
    assert b"gfx950" in hpt.hpt_version()
    # the library is built for gfx950 only
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + os.path.join(ROOT, "cudapathtracer_amd", "libhpt.so")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout if os.path.exists(
        "/opt/rocm/lib/llvm/bin/clang-offload-bundler") else "gfx950"
    assert "gfx950" in out or out.strip() == ""


def test_headers_are_plain_c(tmp_path):
    """the boundary is a C ABI: both headers compile as C99 on their own"""
    import shutil
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "hpt.h"\n#include "hpt_host.h"\nint main(void) { hpt_stats s; hpt_scene_desc d; (void)s; (void)d; return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


def test_host_bvh_equals_reference_fixture(cornell):
    g = np.load(os.path.join(G, "cornell_bvh.npz"))
    for mesh in range(-1, 6):
        d = cornell.bvh_dump(mesh)
        for k in ("bounds", "payload", "area", "miss"):
            assert np.array_equal(bits(d[k]), bits(g["m%d_%s" % (mesh, k)])), (mesh, k)


def test_flattened_layout(cornell):
    g = np.load(os.path.join(G, "cornell_bvh.npz"))
    f = cornell.flat_arrays()
    nodes, tris = f["nodes"], f["tris"]
    # 11 top-level nodes, 6 of them leaves replaced by mesh BVHs of 11+19+19+3+3+3 nodes
    assert nodes.shape == (5 + 58, 8) and tris.shape == (32, 16)
    miss, leaf = nodes.view(np.int32)[:, 3], nodes.view(np.int32)[:, 7]
    assert sorted(leaf[leaf >= 0]) == list(range(32)), "every triangle is referenced by exactly one leaf"
    assert (leaf[leaf >= 0] == np.arange(32)).all(), "triangles are stored in the order the traversal meets them"
    assert (miss > np.arange(len(nodes))).all() and miss.max() == len(nodes)
    # leaf order of prim ids: top level meets meshes 4 1 0 2 3 5, each in its own leaf order
    base = [0, 6, 16, 26, 28, 30]
    want = []
    for m in [4, 1, 0, 2, 3, 5]:
        want += [base[m] + int(x) for x in g["m%d_payload" % m] if x >= 0]
    assert list(f["prim_of_tri"]) == want
    assert list(tris.view(np.int32)[:, 11]) == want
    # triangle records carry the reference's precomputed v0/e1/e2/normal/area
    for t in range(32):
        prim = want[t]
        m = max(i for i in range(6) if base[i] <= prim)
        ref = g["mesh%d_tris" % m][prim - base[m]]
        assert np.array_equal(bits(tris[t, [0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14]]), bits(ref[:12]))
        assert bits(tris[t, 3:4])[0] == bits(ref[12:13])[0]
        assert tris.view(np.int32)[t, 7] == m and tris.view(np.int32)[t, 15] == m
    assert f["lights"] == [(13650.0, 13650.0, 60, 5)]
    assert np.array_equal(bits(np.array([f["sum_light_area"], f["sum_light_power"]], np.float32)), bits(g["light_totals"]))
    d = cornell.describe()
    for p in (d.nodes, d.tris, d.tri_v12, d.node_area):
        assert C.cast(p, C.c_void_p).value % 64 == 0, "flattened arrays are 64-byte aligned"


def _walk(f, o, d, shadow=False, tmax=0.0):
    """Plain-Python walk of the flattened arrays with the same rules the device
    kernel uses; float32 arithmetic via numpy scalars."""
    F = np.float32
    nodes, tris = f["nodes"], f["tris"]
    ni, ti = nodes.view(np.int32), tris.view(np.int32)
    with np.errstate(all="ignore"):
        inv = (F(1) / d).astype(np.float32)
        neg = d < 0
        best, best_t = -1, float(F(1e30))
        limit = tmax - float(F(5) * F(1e-4))
        node, end = 0, len(nodes)
        while node < end:
            lo, hi = nodes[node, 0:3], nodes[node, 4:7]
            a, b = ((lo - o) * inv).astype(np.float32), ((hi - o) * inv).astype(np.float32)
            tmin3, tmax3 = np.where(neg, b, a), np.where(neg, a, b)
            tMin, tMax = tmin3[0], tmax3[0]
            hit = not ((tMin > tmax3[1]) or (tmin3[1] > tMax))
            if tmin3[1] > tMin:
                tMin = tmin3[1]
            if tmax3[1] < tMax:
                tMax = tmax3[1]
            hit = hit and not ((tMin > tmax3[2]) or (tmin3[2] > tMax))
            if not hit:
                node = ni[node, 3]
                continue
            tri = ni[node, 7]
            node += 1
            if tri < 0:
                continue
            T = tris[tri]
            v0, e1, e2, n = T[0:3], T[4:7], T[8:11], T[12:15]
            dot = lambda p, q: F(F(F(p[0] * q[0]) + F(p[1] * q[1])) + F(p[2] * q[2]))
            cross = lambda x, y: np.array([F(x[1] * y[2]) - F(y[1] * x[2]), F(x[2] * y[0]) - F(y[2] * x[0]),
                                           F(x[0] * y[1]) - F(y[0] * x[1])], np.float32)
            if dot(d, n) > 0:
                continue
            pvec = cross(d, e2)
            det = float(dot(e1, pvec))
            if abs(det) < float(F(1e-5)):
                continue
            det_inv = 1.0 / det
            tvec = (o - v0).astype(np.float32)
            u = float(dot(tvec, pvec)) * det_inv
            if u < 0 or u > 1:
                continue
            qvec = cross(tvec, e1)
            v = float(dot(d, qvec)) * det_inv
            if v < 0 or u + v > 1:
                continue
            t = float(dot(e2, qvec)) * det_inv
            if t < 0:
                continue
            if shadow:
                if t < limit:
                    return int(ti[tri, 11]), t
            elif t < best_t:
                best, best_t = int(ti[tri, 11]), t
    return best, best_t


def test_flattened_tree_walk_equals_oracle(cornell, oracle):
    """The single flattened tree gives the two-level reference traversal's answers."""
    f = cornell.flat_arrays()
    osc = oracle.cornell()
    o1, d1 = primary_rays(16, 16, (278, 273, -800), f["cam_scale"])
    o2, d2 = random_rays(200, 21)
    o = np.concatenate([o1, o2])
    d = np.concatenate([d1, d2])
    wp, wt = osc.intersect(o, d)
    for r in range(len(o)):
        p, t = _walk(f, o[r], d[r])
        assert p == wp[r] and t == wt[r], (r, p, wp[r], t, wt[r])
    o, d, sh, tm = shadow_rays(150, 22)
    wp, wt = osc.intersect(o, d, sh, tm)
    for r in range(len(o)):
        p, t = _walk(f, o[r], d[r], True, tm[r])
        assert (p >= 0) == (wp[r] >= 0), r
        if p >= 0:
            assert t < tm[r]


def test_custom_scene_and_bunny_topology(hp, oracle):
    """4968-triangle mesh: std::sort over more than 16 equal-ish keys; leaf order must
    equal the oracle's (which equals the compiled reference's where that is present)."""
    bunny = os.path.join(ROOT, "models", "bunny", "bunny.obj")
    sc = hp.Scene(32, 24, camPos=(0, 0.1, -0.4), fov=40, maxDepth=4)
    sc.Add(bunny, hp.Material(0, albedo=(0.7, 0.7, 0.7)))
    quad = np.array([[[-.05, .25, -.05], [.05, .25, .05], [-.05, .25, .05]],
                     [[-.05, .25, -.05], [.05, .25, -.05], [.05, .25, .05]]], np.float32)
    sc.Add(quad, hp.Material(0, emission=(20, 20, 20)))
    sc.buildBVH()
    from orclib import Material
    osc = oracle.scene(32, 24, (0, 0.1, -0.4), 40)
    osc.add_obj(bunny, Material.make(0, albedo=(0.7, 0.7, 0.7)))
    osc.add_triangles(quad, Material.make(0, emission=(20, 20, 20)))
    osc.build()
    for mesh in (-1, 0, 1):
        a, b = sc.bvh_dump(mesh), osc.bvh_dump(mesh)
        for k in a:
            assert np.array_equal(bits(a[k]), bits(b[k])), (mesh, k)
    f = sc.flat_arrays()
    assert f["nodes"].shape[0] == 1 + (2 * 4968 - 1) + 3 and f["tris"].shape[0] == 4970
    assert len(f["lights"]) == 1 and f["lights"][0][3] == 1


def test_builtin_bunny_scene_equals_the_python_definition(hp):
    """Scene::CreateBuiltinScene(BunnyInCornell) (C++: OBJ reader + float transform) and
    scenes.bunny_in_cornell (numpy) flatten to the same arrays, bit for bit."""
    from cudapathtracer_amd import scenes
    a = hp.Scene.CreateBuiltinScene(hp.Scene.BunnyInCornell, 7, 64, 48).flat_arrays()
    b = scenes.bunny_in_cornell(64, 48, 7).flat_arrays()
    assert a["tris"].shape == (4980, 16)
    for k in ("nodes", "tris", "tri_v12", "node_area"):
        assert np.array_equal(np.asarray(a[k]).view(np.uint32), np.asarray(b[k]).view(np.uint32)), k
    assert a["cam_scale"] == b["cam_scale"]


def test_builtin_glass_cornell_equals_the_python_definition(hp):
    """Scene::CreateBuiltinScene(GlassCornell) == scenes.glass_cornell: same flattened arrays, and
    the only difference from CornellBox is the short box's material type."""
    from cudapathtracer_amd import scenes
    a = hp.Scene.CreateBuiltinScene(hp.Scene.GlassCornell, 7, 64, 48).flat_arrays()
    b = scenes.glass_cornell(64, 48, 7).flat_arrays()
    c = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 7, 64, 48).flat_arrays()
    raw = lambda v: np.atleast_1d(np.asarray(v)).tobytes()
    for k in a:
        assert raw(a[k]) == raw(b[k]), k
    diff = [k for k in a if raw(a[k]) != raw(c[k])]
    assert diff == ["materials"], diff


def _walk_octant_copy(nodes, table, k, nn, o, inv):
    """The marking walk of the device (hpt_device.h, trace<> of the marking variants) in numpy
    float32: returns the OR of the leaf words of the records whose box the ray hits."""
    f32 = np.float32
    copy = nodes[k]
    words = copy.view(np.uint32)
    at, stop, pending = 0, int(table[k]), 0
    steps = 0
    while at < stop:
        steps += 1
        near, far = copy[at, 0:3], copy[at, 4:7]
        if k < 8:
            tn = ((near - o) * inv).astype(f32).max()
            tf = ((far - o) * inv).astype(f32).min()
        else:
            a, b = ((near - o) * inv).astype(f32), ((far - o) * inv).astype(f32)
            tn, tf = np.minimum(a, b).max(), np.maximum(a, b).min()
        if tn <= tf:
            pending |= int(words[at, 7])
            at += 1
        else:
            at = int(words[at, 3]) // 32 - k * nn
    return pending, steps


@pytest.mark.parametrize("scene_name", ["cornell", "glass_cornell_soup"])
def test_octant_copies_reach_exactly_the_leaves_whose_boxes_are_hit(hp, scene_name):
    """The per-octant trees of the marking variants (their own topology, identical boxes merged)
    against the definition they must meet: for a ray with finite reciprocals the triangles marked
    are those whose LEAF box passes the slab test, minus the triangles that face away from every
    ray of the octant -- and the same set comes out of the copies built on the reference's
    topology (HPT_OCTANT_TREES=0).  Float32 arithmetic as on the device.  No GPU needed."""
    import os
    from cudapathtracer_amd import scenes
    f32 = np.float32
    if scene_name == "cornell":
        sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5, 16, 16)
    else:
        rs = np.random.RandomState(3)
        soup = ((rs.rand(20, 1, 3) * 400 + 80) + (rs.rand(20, 3, 3) - 0.5) * 150).astype(np.float32)
        parts = scenes.cornell_shell_triangles() + [("soup", soup, scenes.WHITE)]
        sc = scenes._assemble(parts, 16, 16, 5)
    flat = sc.flat_arrays()
    nn, nt = len(flat["nodes"]), len(flat["tris"])
    assert nt <= 32
    nodes, table = sc.octant_copies(True)
    os.environ["HPT_OCTANT_TREES"] = "0"
    try:
        ref_nodes, ref_table = sc.octant_copies(True)
    finally:
        del os.environ["HPT_OCTANT_TREES"]
    assert int(table[:8].sum()) < int(ref_table[:8].sum()), "the rebuilt copies are smaller"
    assert np.array_equal(nodes[8], ref_nodes[8]) and table[8] == nn      # copy 8 is the reference's tree
    leaf_of = {int(t): i for i, t in enumerate(flat["nodes"][:, 7].view(np.int32)) if t >= 0}
    lo, hi = flat["nodes"][:, 0:3], flat["nodes"][:, 4:7]
    nrm = flat["tris"][:, 12:15]
    rs = np.random.RandomState(9)
    o_all = (rs.rand(600, 3) * [556, 548, 559]).astype(f32)
    d_all = rs.randn(600, 3).astype(f32)
    d_all[:40] *= f32(1e-4) ** rs.randint(0, 2, (40, 3)).astype(f32)      # some nearly axis-parallel rays
    total_new = total_ref = 0
    for o, d in zip(o_all, d_all):
        inv = (f32(1.0) / d).astype(f32)
        assert np.isfinite(inv).all()
        k = int(d[0] < 0) | int(d[1] < 0) << 1 | int(d[2] < 0) << 2
        sgn = np.array([-1 if k & 1 else 1, -1 if k & 2 else 1, -1 if k & 4 else 1], f32)
        want = 0
        for t in range(nt):
            a = nrm[t] * sgn
            if (a >= 0).all() and (a >= 0.5).any():
                continue                                   # rejected for every ray of the octant
            i = leaf_of[t]
            ta, tb = ((lo[i] - o) * inv).astype(f32), ((hi[i] - o) * inv).astype(f32)
            if np.minimum(ta, tb).max() <= np.maximum(ta, tb).min():
                want |= 1 << t
        got, steps = _walk_octant_copy(nodes, table, k, nn, o, inv)
        got_ref, steps_ref = _walk_octant_copy(ref_nodes, ref_table, k, nn, o, inv)
        assert got == want and got_ref == want, (o, d, bin(got), bin(got_ref), bin(want))
        total_new += steps
        total_ref += steps_ref
    assert total_new < 0.75 * total_ref        # and with fewer box tests


def test_octant_trees_are_tuned_for_the_slowest_lane_of_a_wave(hp):
    """The upload tunes the per-octant trees on sampled rays so that the LARGEST step count among the 64 rays of a
    wave goes down (tune_octant_trees in csrc/hpt_core.hip; HPT_OCTANT_TUNE=0 keeps the area rule's trees).  Checked
    here on rays drawn independently of the library's sample: origins on the Cornell box's surfaces, cosine-distributed
    directions.  Both sets of copies mark the same triangles; the tuning is deterministic.  No GPU needed."""
    import os
    f32 = np.float32
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5, 16, 16)
    flat = sc.flat_arrays()
    nn = len(flat["nodes"])
    tuned, table = sc.octant_copies(True)
    again, table_again = sc.octant_copies(True)
    assert np.array_equal(tuned.view(np.uint32), again.view(np.uint32)) and np.array_equal(table, table_again)
    os.environ["HPT_OCTANT_TUNE"] = "0"
    try:
        plain, plain_table = sc.octant_copies(True)
    finally:
        del os.environ["HPT_OCTANT_TUNE"]
    assert not np.array_equal(tuned.view(np.uint32), plain.view(np.uint32))
    tris = flat["tris"]
    rs = np.random.RandomState(21)
    v0, e1, e2, nrm = (tris[:, a:a + 3].astype(float) for a in (0, 4, 8, 12))
    area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    n = 1280
    t = rs.choice(len(tris), n, p=area / area.sum())
    su, v = np.sqrt(rs.rand(n)), rs.rand(n)
    p = v0[t] + e1[t] * (su * (1 - v))[:, None] + e2[t] * (su * v)[:, None]
    r1, phi = rs.rand(n), 2 * np.pi * rs.rand(n)
    N = nrm[t]
    h = np.where(np.abs(N[:, [0]]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
    T = np.cross(h, N)
    T /= np.linalg.norm(T, axis=1, keepdims=True)
    B = np.cross(N, T)
    d = (T * (np.sqrt(r1) * np.cos(phi))[:, None] + B * (np.sqrt(r1) * np.sin(phi))[:, None] + N * np.sqrt(1 - r1)[:, None]).astype(f32)
    o = p.astype(f32)
    steps_tuned, steps_plain = [], []
    for oi, di in zip(o, d):
        if (di == 0).any():
            continue
        inv = (f32(1.0) / di).astype(f32)
        k = int(di[0] < 0) | int(di[1] < 0) << 1 | int(di[2] < 0) << 2
        got_t, st = _walk_octant_copy(tuned, table, k, nn, oi, inv)
        got_p, sp = _walk_octant_copy(plain, plain_table, k, nn, oi, inv)
        assert got_t == got_p
        steps_tuned.append(st)
        steps_plain.append(sp)
    m = len(steps_tuned) // 64 * 64
    wave_tuned = np.array(steps_tuned[:m]).reshape(-1, 64).max(axis=1).mean()
    wave_plain = np.array(steps_plain[:m]).reshape(-1, 64).max(axis=1).mean()
    assert wave_tuned < wave_plain, (wave_tuned, wave_plain)


def test_scene_errors(hp, tmp_path):
    with pytest.raises(hp.HptError):
        hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5, models_dir=str(tmp_path))
    sc = hp.Scene(8, 8)
    with pytest.raises(hp.HptError):
        sc.Add(str(tmp_path / "missing.obj"), hp.Material())
    quadface = tmp_path / "quad.obj"
    quadface.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    with pytest.raises(hp.HptError):
        sc.Add(str(quadface), hp.Material())
    with pytest.raises(hp.HptError):
        sc.buildBVH()          # no meshes
    with pytest.raises(hp.HptError):
        sc.describe()          # not built
    with pytest.raises(hp.HptError):
        hp.Scene.CreateBuiltinScene(3, 5)   # 0 Cornell box, 1 bunny, 2 glass Cornell box; nothing else


def test_save_ppm_tone_curve(hp, tmp_path):
    g = np.load(os.path.join(G, "cornell_shading.npz"))
    v = g["tone_in"]
    n = len(v)
    fb = np.stack([v, v, v], -1).astype(np.float32)
    path = tmp_path / "t.ppm"
    hp.SavePPM(str(path), fb, n, 1)
    raw = path.read_bytes()
    header = b"P6\n%d 1\n255\n" % n
    assert raw.startswith(header) and len(raw) == len(header) + 3 * n
    px = np.frombuffer(raw[len(header):], np.uint8).reshape(n, 3)
    assert np.array_equal(px[:, 0].astype(np.int32), g["tone_out"]) and (px[:, 0] == px[:, 2]).all()
    with pytest.raises(hp.HptError):
        hp.SavePPM(str(tmp_path / "no_such_dir" / "x.ppm"), fb, n, 1)


def test_save_pfm_round_trip(hp, tmp_path):
    w, h = 5, 3
    fb = np.random.RandomState(3).rand(h * w, 3).astype(np.float32) * 40 - 1
    path = tmp_path / "x.pfm"
    hp.SavePFM(path, fb, w, h)
    raw = path.read_bytes()
    head = b"PF\n5 3\n-1.0\n"
    assert raw.startswith(head) and len(raw) == len(head) + w * h * 12
    got = np.frombuffer(raw[len(head):], "<f4").reshape(h, w, 3)[::-1]   # rows are stored bottom to top
    assert np.array_equal(got.reshape(-1, 3), fb)


def test_cli_is_built():
    assert os.access(os.path.join(ROOT, "build", "gpu_path_tracer"), os.X_OK)


@pytest.mark.parametrize("tag", ["bunny", "heightfield64", "heightfield708"])
def test_host_bvh_equals_the_reference_on_large_scenes(hp, tag):
    """The product's own BVH builder (host/BVH.cpp) against what the COMPILED REFERENCE built from
    the same triangle soups (tests/golden/scene_*.npz): leaf order of the large mesh, and bounds,
    miss links and payloads of the top-level BVH and of every mesh BVH."""
    from cudapathtracer_amd import scenes
    from orclib import bvh_hash, fnv_bytes
    g = np.load(os.path.join(G, "scene_%s.npz" % tag))
    parts = scenes.bunny_in_cornell_parts() if tag == "bunny" else scenes.heightfield_in_cornell_parts(int(tag[11:]))
    sc = scenes._assemble(parts, 40, 32, 6)
    order = sc.bvh_dump(len(parts) - 1)["payload"]
    if "big_mesh_leaf_order" in g:
        assert np.array_equal(order, g["big_mesh_leaf_order"])
    assert fnv_bytes(order.astype("<i4").tobytes()) == int(g["big_mesh_leaf_order_fnv"][0])
    assert [bvh_hash(sc.bvh_dump(m)) for m in range(-1, len(parts))] == [int(v) for v in g["bvh_hashes"]]
    f = sc.flat_arrays()
    assert np.array_equal(bits(np.array([f["sum_light_area"], f["sum_light_power"]], np.float32)), bits(g["light_totals"]))


def test_builtin_heightfield_scene_equals_the_python_definition(hp):
    """Scene::CreateHeightfieldScene (C++: MT19937 as numpy runs it, the bilinear noise in double) and
    scenes.heightfield_in_cornell (numpy) flatten to the same arrays, bit for bit -- the 1.0 M-triangle
    scene of the last benchmark configuration included."""
    from cudapathtracer_amd import scenes
    for n in (5, 64, 708):
        a = hp.Scene.CreateNamedScene("heightfield:%d" % n, 7, 64, 48)
        assert a.name == "Heightfield %d" % n
        a, b = a.flat_arrays(), scenes.heightfield_in_cornell(n, 64, 48, 7).flat_arrays()
        assert a["tris"].shape == (12 + 2 * n * n, 16)
        for k in ("nodes", "tris", "tri_v12", "node_area", "prim_of_tri"):
            assert np.array_equal(np.asarray(a[k]).view(np.uint32), np.asarray(b[k]).view(np.uint32)), (n, k)
        assert a["cam_scale"] == b["cam_scale"] and a["lights"] == b["lights"]
    with pytest.raises(hp.HptError):
        hp.Scene.CreateNamedScene("heightfield:0")
    with pytest.raises(hp.HptError):
        hp.Scene.CreateNamedScene("teapot")


def test_mesh_transform(hp, tmp_path):
    """A MeshTriangle with a Transform (translation, rotation, scale composed as the reference's unused
    Math::buildTransformationMatrix, common/MathUtils.hpp:109-116) against the same matrix in numpy."""
    mat = hp.Material(albedo=(0.5, 0.5, 0.5))
    obj = os.path.join(ROOT, "models", "cornellbox", "shortbox.obj")
    t, r, s = (10.0, -20.0, 30.0), (30.0, 45.0, 60.0), (2.0, 0.5, 1.5)
    sc = hp.Scene(32, 32)
    sc.Add(obj, mat, translation=t, rotation=r, scale=s)
    sc.buildBVH()
    got = sc.flat_arrays()
    plain = hp.Scene(32, 32)
    plain.Add(obj, mat)
    plain.buildBVH()
    ref = plain.flat_arrays()
    ax, ay, az = np.radians(r)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    m = rx @ ry @ rz @ np.diag(s)
    # triangles are stored in leaf order, which the transform changes: compare them by primitive id
    def v0_by_prim(f):
        out = np.zeros((len(f["tris"]), 3))
        out[f["prim_of_tri"]] = f["tris"][:, 0:3]
        return out
    want = v0_by_prim(ref) @ m.T + np.asarray(t)
    assert np.abs(v0_by_prim(got) - want).max() < 2e-3
    # identity transform == no transform, bit for bit
    same = hp.Scene(32, 32)
    same.Add(obj, mat, translation=(0, 0, 0), rotation=(0, 0, 0), scale=1.0)
    same.buildBVH()
    assert np.array_equal(same.flat_arrays()["tris"].view(np.uint32), ref["tris"].view(np.uint32))


def test_scene_file(hp, tmp_path):
    """A scene description file that spells out the reference's Cornell box gives the built-in scene bit
    for bit; transforms, the height field statement and the error messages."""
    models = os.path.join(ROOT, "models", "cornellbox")
    e = [8.0 * np.float32(0.747 + 0.058) + 15.6 * np.float32(0.740 + 0.287) + 18.4 * np.float32(0.737 + 0.642)]
    text = """# the reference's Cornell box (common/Scene.cpp:25-73)
name Cornell Box
resolution 64 48
camera 278 273 -800 fov 40
material white lambert albedo 0.725 0.71 0.68
material red lambert albedo 0.63 0.065 0.05
material green lambert albedo 0.14 0.45 0.091
material box lambert roughness 0.2 ior 1.5
material metal metal albedo 0.7 0.5 0.2 metallic 1 roughness 0.005
material light lambert emission %s
mesh {m}/floor.obj white
mesh {m}/shortbox.obj box
mesh {m}/tallbox.obj metal
mesh {m}/left.obj red
mesh {m}/right.obj green
mesh {m}/light.obj light
""".replace("{m}", models)
    from cudapathtracer_amd import scenes
    emission = " ".join("%.9g" % v for v in scenes.light_emission())
    path = tmp_path / "cornell.scene"
    path.write_text(text % emission)
    a = hp.Scene.CreateNamedScene("file:" + str(path), 7)
    assert (a.width, a.height, a.name) == (64, 48, "Cornell Box")
    b = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 7, 64, 48)
    fa, fb = a.flat_arrays(), b.flat_arrays()
    raw = lambda v: np.atleast_1d(np.asarray(v)).tobytes()
    for k in fa:
        assert raw(fa[k]) == raw(fb[k]), k
    # relative paths, a transformed mesh, the height field statement
    (tmp_path / "box.obj").write_text(open(os.path.join(models, "shortbox.obj")).read())
    (tmp_path / "b.scene").write_text("material w lambert\nmaterial g glass roughness 0.1 ior 1.4\nmesh box.obj g translate 0 100 0 scale 2\nheightfield 4 w\n")
    c = hp.Scene.CreateNamedScene("file:" + str(tmp_path / "b.scene"), 3).flat_arrays()
    assert c["tris"].shape[0] == 10 + 32 and c["materials"][0][0] == 2
    for bad, msg in (("mesh nothing.obj w\n", "has not been defined"), ("material g glass\n", "ior"), ("colour red\n", "unknown statement"),
                     ("material w lambert\n", "defines no mesh"), ("material w lambert\nmesh missing.obj w\n", "cannot open")):
        (tmp_path / "bad.scene").write_text(bad)
        with pytest.raises(hp.HptError, match=msg):
            hp.Scene.CreateNamedScene("file:" + str(tmp_path / "bad.scene"), 3)


def test_sphere_in_the_flattened_scene(hp, oracle):
    """Scene::Add(Sphere*): the sphere is a leaf of the top-level BVH (box = center -+ radius), its
    record carries the sphere bit, material indices and primitive ids count through the objects in
    Add order -- and the host BVH equals the oracle's over the same objects."""
    from cudapathtracer_amd import scenes
    from orclib import Material as OMat
    shell = scenes.cornell_shell_triangles()
    sc = hp.Scene(32, 32, camPos=scenes.CORNELL_CAM)
    osc = oracle.scene(32, 32, scenes.CORNELL_CAM, 40.0)
    metal = dict(type=1, albedo=(.7, .5, .2), metallic=1.0, roughness=0.01)
    glass = dict(type=2, roughness=0.1, ior=1.5)
    sc.Add(shell[0][1], hp.Material(**shell[0][2])); osc.add_triangles(shell[0][1], OMat.make(**shell[0][2]))
    sc.AddSphere((200, 150, 300), 80.0, hp.Material(**metal)); osc.add_sphere((200, 150, 300), 80.0, OMat.make(**metal))
    sc.Add(shell[3][1], hp.Material(**shell[3][2])); osc.add_triangles(shell[3][1], OMat.make(**shell[3][2]))
    sc.AddSphere((400, 100, 200), 50.0, hp.Material(**glass)); osc.add_sphere((400, 100, 200), 50.0, OMat.make(**glass))
    sc.buildBVH(); osc.build()
    f = sc.flat_arrays()
    w15 = f["tris"][:, 15].view(np.uint32)
    is_sphere = (w15 >> 31) == 1
    assert is_sphere.sum() == 2 and len(f["tris"]) == 10
    assert sorted(f["prim_of_tri"][is_sphere]) == [6, 9] and sorted(f["prim_of_tri"][~is_sphere]) == [0, 1, 2, 3, 4, 5, 7, 8]
    rec = f["tris"][is_sphere][np.argsort(f["prim_of_tri"][is_sphere])]
    assert np.array_equal(rec[:, 0:3], np.array([[200, 150, 300], [400, 100, 200]], np.float32))
    assert np.array_equal(rec[:, 4:6], np.array([[80, 6400], [50, 2500]], np.float32))
    assert list(rec[:, 7].view(np.int32)) == [1, 3] and list((rec[:, 15].view(np.uint32) & 0x7FFFFFFF)) == [1, 3]
    assert f["materials"].shape[0] == 4 and f["materials"][3][0] == 2
    a, b = sc.bvh_dump(-1), osc.bvh_dump(-1)
    assert np.array_equal(bits(a["bounds"]), bits(b["bounds"])) and np.array_equal(a["miss"], b["miss"])
    leaf = f["nodes"][f["nodes"][:, 7].view(np.int32) >= 0]
    tri_of_leaf = leaf[:, 7].view(np.int32)
    k = np.flatnonzero(is_sphere[tri_of_leaf])[0]
    c, r = f["tris"][tri_of_leaf[k], 0:3], f["tris"][tri_of_leaf[k], 4]
    assert np.array_equal(leaf[k, 0:3], c - r) and np.array_equal(leaf[k, 4:7], c + r)
    with pytest.raises(hp.HptError, match="cannot be a light"):
        sc.AddSphere((0, 0, 0), 1.0, hp.Material(emission=(1, 1, 1)))
    with pytest.raises(hp.HptError, match="positive radius"):
        sc.AddSphere((0, 0, 0), 0.0, hp.Material())


HOST_MATS = {
    "lambert": ("cornell_shading.npz", dict(type=0, albedo=(0.725, 0.71, 0.68))),
    "metal": ("cornell_shading.npz", dict(type=1, albedo=(0.7, 0.5, 0.2), metallic=1.0, roughness=0.005)),
    "metal_rough": ("cornell_shading.npz", dict(type=1, albedo=(0.9, 0.6, 0.3), metallic=0.3, roughness=0.4)),
    "glass": ("glass.npz", dict(type=2, roughness=0.2, ior=1.5)),
    "glass_water": ("glass.npz", dict(type=2, roughness=0.05, ior=1.33)),
    "glass_rough": ("glass.npz", dict(type=2, roughness=0.7, ior=2.4)),
}


@pytest.mark.parametrize("name", sorted(HOST_MATS))
def test_host_material_equals_the_compiled_reference(hp, name):
    """Material::bsdf / pdf / sample of the C++ host class against the vectors the compiled reference's Material
    produced (common/Material.hpp:186-384; fixtures of tests/golden/make_golden.py): bit for bit."""
    fixture, kw = HOST_MATS[name]
    g = np.load(os.path.join(G, fixture))
    m = hp.Material(**kw)
    b = lambda a: np.ascontiguousarray(a).view(np.uint32)
    bsdf, pdf = hp.material_eval(m, g["wi"], g["wo"], g["nrm"])
    assert np.array_equal(b(bsdf), b(g[name + "_bsdf"])) and np.array_equal(b(pdf), b(g[name + "_pdf"]))
    wi = hp.material_sample(m, g[name + "_sample_u"], g["wo"], g["nrm"])
    assert np.array_equal(b(wi), b(g[name + "_sample_wi"]))
    bsdf, pdf = hp.material_eval(m, wi, g["wo"], g["nrm"])
    assert np.array_equal(b(bsdf), b(g[name + "_bsdf_at_sample"])) and np.array_equal(b(pdf), b(g[name + "_pdf_at_sample"]))


def test_host_rng_equals_the_reference_stream(hp):
    """RNG(iter, index) of the host side = the reference's mt19937 stream (common/MathUtils.hpp:22-36)."""
    g = np.load(os.path.join(G, "cornell_shading.npz"))
    assert np.array_equal(hp.rng_floats(3, 12345, 64).view(np.uint32), g["rng_k3_m12345"].view(np.uint32))
    assert np.array_equal(hp.rng_floats(0, 0, 64).view(np.uint32), g["rng_k0_m0"].view(np.uint32))


def test_image_diff_tool(hp, tmp_path):
    """tools/image_diff.py on PFM (linear radiance) and PPM (tone-mapped) files written by the host library."""
    rs = np.random.RandomState(5)
    a = (rs.rand(12 * 8, 3) * 1.5).astype(np.float32)
    b = a.copy()
    b[7] += np.float32(0.25)
    for name, img in (("a", a), ("b", b)):
        hp.SavePFM(str(tmp_path / (name + ".pfm")), img, 12, 8)
        hp.SavePPM(str(tmp_path / (name + ".ppm")), img, 12, 8)
    tool = os.path.join(ROOT, "tools", "image_diff.py")
    r = subprocess.run([sys.executable, tool, str(tmp_path / "a.pfm"), str(tmp_path / "a.pfm")], capture_output=True, text=True)
    assert r.returncode == 0 and "identical" in r.stdout and "rmse 0 " in r.stdout
    r = subprocess.run([sys.executable, tool, str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm"), "--max-rmse-clamped", "1e-4"], capture_output=True, text=True)
    want = np.sqrt(np.mean((a.astype(np.float64) - b) ** 2))
    assert r.returncode == 1 and "1 of 96 pixels differ" in r.stdout and ("rmse %.6g" % want) in r.stdout
    r = subprocess.run([sys.executable, tool, str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")], capture_output=True, text=True)
    assert r.returncode == 0 and "1 of 96 pixels differ" in r.stdout


def _tree_facts(f):
    """Checks the flattened pre-order tree of a scene (left child = next node, right child = the left child's miss
    link, leaves carry triangle records) and returns (surface-area cost, stack entries a depth-first walk needs,
    leaf triangle indices in order)."""
    nodes = f["nodes"]
    n = len(nodes)
    lo, hi = nodes[:, 0:3].astype(np.float64), nodes[:, 4:7].astype(np.float64)
    miss, tri = nodes[:, 3].view(np.int32), nodes[:, 7].view(np.int32)
    leaf = tri >= 0
    area = 2 * ((hi - lo)[:, 0] * (hi - lo)[:, 1] + (hi - lo)[:, 1] * (hi - lo)[:, 2] + (hi - lo)[:, 2] * (hi - lo)[:, 0])
    need = np.zeros(n, np.int64)
    assert miss[0] == n
    for i in range(n - 1, -1, -1):      # children follow their parent in pre-order
        if leaf[i]:
            continue
        left, right = i + 1, miss[i + 1]
        assert i < left < right < n and miss[right] == miss[i], i
        assert np.array_equal(np.minimum(lo[left], lo[right]), lo[i]) and np.array_equal(np.maximum(hi[left], hi[right]), hi[i]), i
        need[i] = max(need[left] + (0 if leaf[right] else 1), need[right])
    leaves = tri[leaf]
    assert sorted(leaves.tolist()) == list(range(len(f["tris"])))   # every triangle record in exactly one leaf
    return float(area[~leaf].sum()), int(need[0]), leaves


@pytest.mark.parametrize("name", ["bunny", "heightfield:40", "glass"])
def test_sah_build(hp, name):
    """Scene::BvhBuild::Sah (BVHAccel::SplitMethod::SAH, declared but not implemented by the reference,
    common/BVH.hpp:28): a valid two-level tree over the same triangle records with a lower surface-area cost than the
    reference's median splits, a traversal stack no deeper than log2 of the triangle count, the emitting mesh's
    subtree (which Scene::sampleLight descends) and all light tables untouched; deterministic."""
    ref = hp.Scene.CreateNamedScene(name, 5, 64, 64, bvh_build=hp.BVH_BUILD_HOST)
    sah = hp.Scene.CreateNamedScene(name, 5, 64, 64, bvh_build=hp.BVH_BUILD_SAH)
    assert ref.bvh_build == hp.BVH_BUILD_HOST and sah.bvh_build == hp.BVH_BUILD_SAH and not sah.gpu_build
    fr, fs = ref.flat_arrays(), sah.flat_arrays()
    # the default ("auto") picks the surface-area tree for scenes of more than 64 primitives, the reference's below
    auto = hp.Scene.CreateNamedScene(name, 5, 64, 64, bvh_build=hp.BVH_BUILD_AUTO)
    assert auto.bvh_build == (hp.BVH_BUILD_SAH if len(fr["tris"]) > 64 else hp.BVH_BUILD_HOST)
    assert hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5).bvh_build == hp.BVH_BUILD_HOST
    # the reference's own tree rides along for the rays with a zero direction component (hpt_scene_desc::ref_nodes):
    # bit for bit the node array of the reference build (leaf words index the same triangle records)
    assert fr["ref_nodes"] is None
    assert np.array_equal(fs["ref_nodes"].view(np.uint32), fr["nodes"].view(np.uint32))
    # and the trees the Scene shows its callers (Scene::bvh, Scene::mesh_bvhs) are the reference's either way
    for mesh in (-1, 0, len(fr["lights"])):
        dr, ds = ref.bvh_dump(mesh), sah.bvh_dump(mesh)
        assert all(np.array_equal(dr[k], ds[k]) for k in dr)
    cost_r, need_r, leaves_r = _tree_facts(fr)
    cost_s, need_s, leaves_s = _tree_facts(fs)
    nt = len(fr["tris"])
    assert cost_s < (0.9 if name != "glass" else 1.0001) * cost_r, (cost_s, cost_r)
    assert need_s <= int(np.log2(nt)) and need_s <= need_r + 1, (need_s, need_r)
    # the triangle records are the reference build's, in the reference's traversal order (the kernels' tie rule --
    # lowest record index among equal distances -- then names the triangle the reference names); only the tree differs
    assert np.array_equal(fr["prim_of_tri"], fs["prim_of_tri"])
    assert np.array_equal(fr["tris"].view(np.uint32), fs["tris"].view(np.uint32))
    assert np.array_equal(fr["tri_v12"].view(np.uint32), fs["tri_v12"].view(np.uint32))
    assert not np.array_equal(leaves_r, leaves_s) or name == "glass"
    # lights: same areas and materials; the light mesh's subtree has the reference's shape, areas and leaf order
    assert [(a, ra, m) for a, ra, _, m in fr["lights"]] == [(a, ra, m) for a, ra, _, m in fs["lights"]]
    assert (fr["sum_light_area"], fr["sum_light_power"]) == (fs["sum_light_area"], fs["sum_light_power"])
    for (_, _, root_r, _), (_, _, root_s, _) in zip(fr["lights"], fs["lights"]):
        end_r = fr["nodes"][root_r, 3].view(np.int32)
        k = int(end_r) - root_r
        assert np.array_equal(fr["nodes"][root_r:root_r + k, [0, 1, 2, 4, 5, 6]], fs["nodes"][root_s:root_s + k, [0, 1, 2, 4, 5, 6]])
        assert np.array_equal(fr["node_area"][root_r:root_r + k], fs["node_area"][root_s:root_s + k])
        tr, ts = fr["nodes"][root_r:root_r + k, 7].view(np.int32), fs["nodes"][root_s:root_s + k, 7].view(np.int32)
        assert np.array_equal(tr >= 0, ts >= 0)
        assert np.array_equal(fr["prim_of_tri"][tr[tr >= 0]], fs["prim_of_tri"][ts[ts >= 0]])
    again = hp.Scene.CreateNamedScene(name, 5, 64, 64, bvh_build=hp.BVH_BUILD_SAH).flat_arrays()
    assert np.array_equal(again["nodes"].view(np.uint32), fs["nodes"].view(np.uint32))
    # switching a scene back and forth
    sah.buildBVH(gpu=False)
    assert sah.bvh_build == hp.BVH_BUILD_HOST
    assert np.array_equal(sah.flat_arrays()["nodes"].view(np.uint32), fr["nodes"].view(np.uint32))
    sah.buildBVH(sah=True)
    assert np.array_equal(sah.flat_arrays()["nodes"].view(np.uint32), fs["nodes"].view(np.uint32))
