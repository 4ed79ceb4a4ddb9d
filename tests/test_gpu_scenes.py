"""GPU parity on the non-Cornell configurations: the bunny in the Cornell shell
(4980 triangles, records read through L1/L2 instead of LDS) and displaced
height fields up to ~1.0 M triangles (BVH of 2.0 M nodes).  Same bar as
test_gpu_parity.py: ids, t bits, ray counters and images bit-exact against the
CPU oracle built from the same triangle arrays."""
import os

import numpy as np
import pytest

from orclib import RNG_PHILOX, TRIG_PORTABLE, Material, primary_rays, random_rays, shadow_rays

pytestmark = pytest.mark.gpu


def _oracle_scene(oracle, parts, w, h):
    from cudapathtracer_amd import scenes
    sc = oracle.scene(w, h, scenes.CORNELL_CAM, 40.0)
    for _, tris, mat in parts:
        sc.add_triangles(tris, Material.make(**mat))
    return sc.build()


def _same_hits(ctx, osc, o, d, sh=None, tm=None):
    gp, gt = ctx.intersect(o, d, sh, tm)
    wp, wt = osc.intersect(o, d, sh, tm)
    assert np.array_equal(gp, wp), "%d ids differ" % (gp != wp).sum()
    assert np.array_equal(gt.view(np.int64), wt.view(np.int64))
    return wp


def _same_occlusion(ctx, osc, o, d, sh, tm):
    """any-hit queries on scenes of more than 64 triangles: the answer (occluded or not) is the
    reference's; the occluder named may be another triangle that passes the same test (the
    depth-first walk meets leaves in another order), so ids and t are compared where the ids agree"""
    gp, gt = ctx.intersect(o, d, sh, tm)
    wp, wt = osc.intersect(o, d, sh, tm)
    assert np.array_equal(gp >= 0, wp >= 0), "%d occlusion answers differ" % ((gp >= 0) != (wp >= 0)).sum()
    same = gp == wp
    assert np.array_equal(gt[same].view(np.int64), wt[same].view(np.int64))
    assert (gt[gp >= 0] < tm[gp >= 0] - 5e-4).all()
    return wp


def _shadow_batch(n, seed, down=False):
    """any-hit queries towards the light quad, or (down) from high points towards the floor:
    the height field faces up, so only rays that come from above can be occluded by it"""
    rs = np.random.RandomState(seed)
    if down:
        p = (rs.rand(n, 3) * [500, 300, 500] + [28, 200, 28]).astype(np.float32)
        x = (rs.rand(n, 3) * [500, 0, 500] + [28, 0, 28]).astype(np.float32)
    else:
        p = (rs.rand(n, 3) * [556, 300, 559.2]).astype(np.float32)
        x = (rs.rand(n, 3) * [130, 0, 105] + [213, 548.7, 227]).astype(np.float32)
    px = (x - p).astype(np.float32)
    ln = np.sqrt((px[:, 0] * px[:, 0] + px[:, 1] * px[:, 1]) + px[:, 2] * px[:, 2], dtype=np.float32)
    d = (px * (np.float32(1) / ln)[:, None]).astype(np.float32)
    return (p + d * np.float32(1e-5)).astype(np.float32), d, np.ones(n, np.uint8), ln.astype(np.float64)


def _zero_component_rays(parts, n, seed):
    """Rays with one or two zero direction components (infinite reciprocals: 0 * inf = NaN in the slab test, the case
    in which the leaves a ray reaches depend on the interior nodes of the tree), a third of them starting exactly ON
    planes of leaf boxes (at vertices of the scene's triangles), the rest at random points."""
    rs = np.random.RandomState(seed)
    verts = np.concatenate([np.asarray(t, np.float32).reshape(-1, 3) for _, t, _ in parts])
    o = (rs.rand(n, 3) * [556, 548.8, 559.2]).astype(np.float32)
    pick = rs.rand(n) < 1.0 / 3.0
    o[pick] = verts[rs.randint(0, len(verts), int(pick.sum()))]
    half = rs.rand(n) < 0.5          # of those, half are moved off the vertex along one axis only (still on two planes)
    o[pick & half, rs.randint(0, 3)] += np.float32(3.0)
    dirs = []
    for a in range(3):
        for sgn in (1.0, -1.0):
            v = np.zeros(3, np.float32)
            v[a] = sgn
            dirs.append(v)                       # two zero components
    for a in range(3):
        for k in range(6):
            ang = rs.rand() * 2 * np.pi
            v = np.zeros(3, np.float32)
            v[(a + 1) % 3], v[(a + 2) % 3] = np.cos(ang), np.sin(ang)
            if k == 0:
                v[a] = -0.0                      # a negative zero: dirIsNeg is false for it, the reciprocal is -inf
            dirs.append(v)                       # one zero component
    dirs = np.array(dirs, np.float32)
    dirs /= np.sqrt((dirs.astype(np.float64) ** 2).sum(1))[:, None].astype(np.float32)
    d = dirs[rs.randint(0, len(dirs), n)]
    return o, np.ascontiguousarray(d, np.float32)


@pytest.mark.parametrize("name", ["bunny", "heightfield128", "heightfield8", "heightfield4"])
def test_scene_hits_and_images(hp, oracle, gpu_ctx, name):
    """bunny / heightfield128: two-child records read through L1/L2; heightfield8 (140 triangles):
    the same records staged in LDS; heightfield4 (44 triangles): the marking variant with a
    64-bit mask and per-octant node copies."""
    from cudapathtracer_amd import scenes
    parts = scenes.bunny_in_cornell_parts() if name == "bunny" else scenes.heightfield_in_cornell_parts(int(name[11:]))
    w, h, depth, spp = 56, 40, 6, 2
    sc = scenes._assemble(parts, w, h, depth)
    osc = _oracle_scene(oracle, parts, w, h)
    gpu_ctx.upload(sc)
    assert gpu_ctx.stats()["variant"] == {"heightfield8": 6, "heightfield4": 3}.get(name, 5)
    o, d = primary_rays(192, 192, scenes.CORNELL_CAM, osc.info()["cam_scale"])
    hit = _same_hits(gpu_ctx, osc, o, d)
    big = 4 if name == "bunny" else 4      # mesh index of the large mesh
    base = sum(len(p[1]) for p in parts[:big])
    assert (hit >= base).mean() > 0.05, "the large mesh is in view"
    _same_hits(gpu_ctx, osc, *random_rays(1 << 17, 31))
    # rays with zero direction components, many of them starting on planes of leaf boxes: the default build of the
    # larger scenes is a surface-area tree, and these rays must walk the reference's own tree beside it
    zo, zd = _zero_component_rays(parts, 1 << 15, 33)
    assert (_same_hits(gpu_ctx, osc, zo, zd) >= 0).mean() > 0.3
    same = _same_hits if gpu_ctx.stats()["variant"] in (2, 3) else _same_occlusion
    same(gpu_ctx, osc, zo, zd, np.ones(len(zo), np.uint8), np.full(len(zo), 300.0))
    occ = same(gpu_ctx, osc, *_shadow_batch(1 << 16, 32, down=(name != "bunny")))
    assert 0.02 < (occ >= 0).mean() <= 1.0
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for mode in (1, 2):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 256, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode


@pytest.mark.parametrize("name", ["cornell", "cornell_water", "bunny"])
def test_glass_scenes(hp, oracle, gpu_ctx, name):
    """Material::Type::Glass: the Cornell box with a glass short box (LDS marking variant) and a
    glass bunny (two-child records): images and ray counters bit-exact against the oracle, whose
    glass is pinned to the compiled reference by tests/golden/glass.npz."""
    from cudapathtracer_amd import scenes
    if name == "bunny":
        parts = scenes.cornell_shell_triangles() + [("bunny", scenes.bunny_triangles(), dict(type=2, roughness=0.3, ior=1.45))]
        w, h, depth, spp = 48, 40, 10, 2
    else:
        parts = scenes.glass_cornell_parts(glass=dict(type=2, roughness=0.05, ior=1.33) if name == "cornell_water" else None)
        w, h, depth, spp = 64, 48, 12, 4
    sc = scenes._assemble(parts, w, h, depth)
    osc = _oracle_scene(oracle, parts, w, h)
    gpu_ctx.upload(sc)
    assert gpu_ctx.stats()["variant"] == (5 if name == "bunny" else 2)
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    # glass bounces are specular: the bounce after one sends no shadow ray
    assert ost["shadow_queries"] < ost["closest_queries"]
    for mode in (1, 2):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 256, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode
    if name == "cornell":
        plain = _oracle_scene(oracle, scenes.glass_cornell_parts(glass=scenes.WHITE), w, h)
        other, _ = plain.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
        assert not np.array_equal(other, want_acc)


@pytest.mark.parametrize("name", ["bunny", "heightfield128", "heightfield8"])
def test_culled_traversal(hp, oracle, gpu_ctx, name):
    """HPT_TRAVERSAL_CULLED (opt-in: near child first, boxes behind the origin or beyond the closest
    hit skipped).  It does not visit the reference's set of boxes, so it is held to its own bar,
    written here: at most 1 ray in 10^5 may report another (prim, t) than the oracle (float rounding
    at a box face), shadow-query counters may differ by the same fraction, and at most 1 pixel in
    10^4 of a rendered image may differ.  Measured: no difference at all on these inputs."""
    from cudapathtracer_amd import scenes
    parts = scenes.bunny_in_cornell_parts() if name == "bunny" else scenes.heightfield_in_cornell_parts(int(name[11:]))
    w, h, depth, spp = 56, 40, 6, 2
    sc = scenes._assemble(parts, w, h, depth)
    osc = _oracle_scene(oracle, parts, w, h)
    with pytest.raises(hp.HptError):
        gpu_ctx.set_traversal(2)
    gpu_ctx.set_traversal(hp.TRAVERSAL_CULLED)
    try:
        gpu_ctx.upload(sc)
        assert gpu_ctx.stats()["variant"] == 7
        o, d = primary_rays(192, 192, scenes.CORNELL_CAM, osc.info()["cam_scale"])
        batches = [(o, d, None, None), random_rays(1 << 17, 31) + (None, None), _shadow_batch(1 << 16, 32, down=(name != "bunny"))]
        for bo, bd, sh, tm in batches:
            gp, gt = gpu_ctx.intersect(bo, bd, sh, tm)
            wp, wt = osc.intersect(bo, bd, sh, tm)
            if sh is None:
                bad = (gp != wp) | (gt.view(np.int64) != wt.view(np.int64))
            else:  # any-hit: which occluder is found first depends on the order; occluded or not does not
                bad = (gp >= 0) != (wp >= 0)
            assert bad.mean() <= 1e-5, bad.sum()
        want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
        want = oracle.resolve(want_acc, spp)
        for mode in (1, 2):
            gpu_ctx.stats_reset()
            gpu_ctx.accum_clear()
            gpu_ctx.render(mode, 256, depth, 0, spp)
            got = gpu_ctx.resolve(spp)
            st = gpu_ctx.stats()
            assert st["closest_queries"] == ost["closest_queries"]
            assert abs(st["shadow_queries"] - ost["shadow_queries"]) <= 1e-5 * ost["shadow_queries"] + 1
            assert (got.view(np.int32) != want.view(np.int32)).any(-1).mean() <= 1e-4, "mode %d" % mode
        # scenes of up to 64 triangles keep their marking variant, whatever is asked for
        gpu_ctx.upload(hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5, 16, 16))
        assert gpu_ctx.stats()["variant"] == 2
    finally:
        gpu_ctx.set_traversal(hp.TRAVERSAL_REFERENCE)


@pytest.mark.parametrize("sah", [False, True])
@pytest.mark.parametrize("variant", [0, 1])
def test_forced_variants(hp, oracle, gpu_ctx, monkeypatch, variant, sah):
    """the variants that are no scene's default: the threaded walk through L1/L2 (0) and from
    LDS (1), forced through the environment at upload; on the reference's tree and on the surface-area tree
    (whose leaves the walk meets in another order: ties between equal distances still go to the reference's triangle)."""
    from cudapathtracer_amd import scenes
    parts = scenes.heightfield_in_cornell_parts(8)
    w, h, depth, spp = 40, 24, 5, 2
    sc = scenes._assemble(parts, w, h, depth)
    sc.buildBVH(sah=True) if sah else sc.buildBVH(gpu=False)
    osc = _oracle_scene(oracle, parts, w, h)
    monkeypatch.setenv("HPT_TRACE_VARIANT", str(variant))
    gpu_ctx.upload(sc)
    monkeypatch.delenv("HPT_TRACE_VARIANT")
    # (the threaded walk from LDS holds one node array: a scene that brings the reference's tree beside its own
    # keeps its two-child-record walk, which holds both)
    assert gpu_ctx.stats()["variant"] == (6 if (variant == 1 and sah) else variant)
    _same_hits(gpu_ctx, osc, *random_rays(1 << 16, 51))
    _same_hits(gpu_ctx, osc, *primary_rays(128, 128, scenes.CORNELL_CAM, osc.info()["cam_scale"]))
    (_same_occlusion if sah else _same_hits)(gpu_ctx, osc, *_shadow_batch(1 << 15, 52, down=True))
    want_acc, _ = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for mode in (1, 2):
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 256, depth, 0, spp)
        assert np.array_equal(gpu_ctx.resolve(spp).view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))


@pytest.mark.parametrize("name", ["cornell", "soup40"])
def test_octant_copies_on_the_reference_topology(hp, oracle, gpu_ctx, monkeypatch, name):
    """HPT_OCTANT_TREES=0: the marking variants walk the reference's own tree (pruned per octant)
    instead of the trees they build for themselves -- the path taken by scenes whose boxes are not
    nested.  Same hits, same image, and both equal the oracle."""
    from cudapathtracer_amd import scenes
    if name == "cornell":
        parts = scenes.glass_cornell_parts(glass=scenes.WHITE)
    else:
        light = dict(type=0, emission=scenes.light_emission())
        quad = np.array([[[100, 500, 100], [400, 500, 100], [400, 500, 400]],
                         [[100, 500, 100], [400, 500, 400], [100, 500, 400]]], np.float32)[:, ::-1]
        parts = [("light", quad, light), ("a", _soup(25, 14), scenes.WHITE), ("b", _soup(13, 15), scenes.METAL)]
    w, h, depth, spp = 40, 32, 6, 3
    sc = scenes._assemble(parts, w, h, depth)
    osc = _oracle_scene(oracle, parts, w, h)
    want_acc, _ = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for own_trees in ("0", "1"):
        monkeypatch.setenv("HPT_OCTANT_TREES", own_trees)
        gpu_ctx.upload(sc)
        monkeypatch.delenv("HPT_OCTANT_TREES")
        assert gpu_ctx.stats()["variant"] == (2 if name == "cornell" else 3)
        _same_hits(gpu_ctx, osc, *random_rays(1 << 15, 71))
        _same_hits(gpu_ctx, osc, *_shadow_batch(1 << 14, 72))
        for mode in (1, 2):
            gpu_ctx.accum_clear()
            gpu_ctx.render(mode, 128, depth, 0, spp)
            assert np.array_equal(gpu_ctx.resolve(spp).view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))


def _soup(n, seed, size=200.0):
    """n random triangles around the centre of the Cornell volume"""
    rs = np.random.RandomState(seed)
    c = rs.rand(n, 1, 3) * [400, 400, 400] + [78, 70, 80]
    return (c + (rs.rand(n, 3, 3) - 0.5) * size).astype(np.float32)


@pytest.mark.parametrize("name", ["one_triangle", "no_light", "degenerate", "soup33", "soup64", "soup65"])
def test_small_custom_scenes(hp, oracle, gpu_ctx, name):
    """edges of the scene description: a tree that is a single leaf, no emitter at all, zero-area
    triangles (NaN normals), and triangle counts on both sides of the variant thresholds."""
    from cudapathtracer_amd import scenes
    light = dict(type=0, emission=scenes.light_emission())
    quad = np.array([[[100, 500, 100], [400, 500, 100], [400, 500, 400]],
                     [[100, 500, 100], [400, 500, 400], [100, 500, 400]]], np.float32)[:, ::-1]  # faces down
    if name == "one_triangle":
        parts = [("light", quad[:1], light)]
    elif name == "no_light":
        parts = [("a", _soup(5, 1), scenes.WHITE), ("b", _soup(3, 2), scenes.METAL)]
    elif name == "degenerate":
        t = _soup(6, 3)
        t[1, 2] = t[1, 1]          # two equal vertices
        t[4] = t[4, 0]             # a point
        parts = [("light", quad, light), ("a", t, scenes.WHITE)]
    else:
        n = int(name[4:])
        parts = [("light", quad, light), ("a", _soup(n - 2 - n // 3, 4), scenes.WHITE), ("b", _soup(n // 3, 5), scenes.METAL)]
    w, h, depth, spp = 24, 16, 4, 3
    sc = scenes._assemble(parts, w, h, depth)
    osc = _oracle_scene(oracle, parts, w, h)
    gpu_ctx.upload(sc)
    nt = sum(len(p[1]) for p in parts)
    assert gpu_ctx.stats()["variant"] == (2 if nt <= 32 else 3 if nt <= 64 else 6)
    o, d = primary_rays(64, 64, scenes.CORNELL_CAM, osc.info()["cam_scale"])
    _same_hits(gpu_ctx, osc, o, d)
    _same_hits(gpu_ctx, osc, *random_rays(1 << 14, 61))
    (_same_hits if nt <= 64 else _same_occlusion)(gpu_ctx, osc, *_shadow_batch(1 << 13, 62))
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for mode in (1, 2):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 64, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode


def test_million_triangle_heightfield(hp, oracle, gpu_ctx):
    """~1.0 M triangles / 2.0 M nodes: topology equal to the oracle's, hits bit-exact."""
    from cudapathtracer_amd import scenes
    parts = scenes.heightfield_in_cornell_parts(708)
    sc = scenes._assemble(parts, 32, 32, 4)
    f = sc.flat_arrays()
    n_hf = 2 * 708 * 708
    # 4 top-level interior nodes + mesh BVHs of 6, 2, 2, 2 and n_hf triangles (2n - 1 nodes each)
    assert f["tris"].shape[0] == n_hf + 12 and f["nodes"].shape[0] == 4 + 11 + 3 + 3 + 3 + (2 * n_hf - 1)
    osc = _oracle_scene(oracle, parts, 32, 32)
    a, b = sc.bvh_dump(4), osc.bvh_dump(4)
    assert np.array_equal(a["payload"], b["payload"]) and np.array_equal(a["miss"], b["miss"])
    assert np.array_equal(a["bounds"].view(np.int32), b["bounds"].view(np.int32))
    gpu_ctx.upload(sc)
    _same_hits(gpu_ctx, osc, *random_rays(1 << 17, 41))
    occ = _same_occlusion(gpu_ctx, osc, *_shadow_batch(1 << 16, 42, down=True))
    assert (occ >= 0).mean() > 0.5
    want_acc, _ = osc.render(RNG_PHILOX, TRIG_PORTABLE, 4, 2, fixed=True)
    gpu_ctx.accum_clear()
    gpu_ctx.render(2, 256, 4, 0, 2)
    assert np.array_equal(gpu_ctx.resolve(2).view(np.int32), oracle.resolve(want_acc, 2).view(np.int32))


def test_full_size_properties(hp, gpu_ctx):
    """At the headline size (1024x1024, depth 20) the oracle is too slow to render,
    so the test uses properties that hold by construction: mode 1 and mode 2 give
    the same accumulators, a split sample range sums to the whole, ray counters are
    identical, every path is accounted for."""
    import torch
    w = h = 1024
    depth, spp = 20, 4
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    gpu_ctx.upload(sc)
    st = torch.cuda.current_stream()
    a = torch.zeros(3 * w * h, dtype=torch.int64, device="cuda:0")
    b = torch.zeros_like(a)
    gpu_ctx.stats_reset()
    gpu_ctx.render(1, 256, depth, 0, spp, accum=a, stream=st)
    torch.cuda.synchronize()
    s1 = gpu_ctx.stats()
    gpu_ctx.stats_reset()
    gpu_ctx.render(2, 256, depth, 0, 1, accum=b, stream=st)
    gpu_ctx.render(1, 64, depth, 1, 2, accum=b, stream=st)
    gpu_ctx.render(2, 192, depth, 3, 1, accum=b, stream=st)
    torch.cuda.synchronize()
    s2 = gpu_ctx.stats()
    assert torch.equal(a, b)
    assert s1["paths"] == s2["paths"] == w * h * spp
    assert (s1["closest_queries"], s1["shadow_queries"]) == (s2["closest_queries"], s2["shadow_queries"])
    rays_per_path = (s1["closest_queries"] + s1["shadow_queries"]) / s1["paths"]
    assert 11.0 < rays_per_path < 12.3          # 11.64 measured on the reference
    img = gpu_ctx.resolve(spp, accum=a).reshape(h, w, 3)
    assert np.isfinite(img).all() and (img >= 0).all()
    assert img[:8].max() == 0.0                 # the rows above the box see no geometry: no background radiance
    assert 0.3 < img[400:600, 400:600].mean() < 1.5


def test_full_size_heightfield_properties(hp, gpu_ctx):
    """The last benchmark configuration's scene and resolution (1.0 M triangles, 2048 x 2048, depth 20)
    through the C++ host's own scene builder: the single-kernel and the streamed mode (wave scheduler)
    give the same accumulators, a split sample range sums to the whole, every path is accounted for,
    and hpt_render_static-sized pieces are not needed (one call traces 4.2 M paths per sample)."""
    import torch
    w = h = 2048
    depth, spp = 20, 2
    sc = hp.Scene.CreateNamedScene("heightfield:708", depth, w, h)
    gpu_ctx.upload(sc)
    assert gpu_ctx.stats()["variant"] == 5
    st = torch.cuda.current_stream()
    a = torch.zeros(3 * w * h, dtype=torch.int64, device="cuda:0")
    b = torch.zeros_like(a)
    gpu_ctx.stats_reset()
    gpu_ctx.render(2, 256, depth, 0, spp, accum=a, stream=st)
    gpu_ctx.synchronize()
    s1 = gpu_ctx.stats()
    gpu_ctx.stats_reset()
    gpu_ctx.render(1, 256, depth, 0, 1, accum=b, stream=st)
    gpu_ctx.render(2, 128, depth, 1, 1, accum=b, stream=st)
    gpu_ctx.synchronize()
    s2 = gpu_ctx.stats()
    assert torch.equal(a, b)
    assert s1["paths"] == s2["paths"] == w * h * spp
    assert (s1["closest_queries"], s1["shadow_queries"]) == (s2["closest_queries"], s2["shadow_queries"])
    assert 9.5 < (s1["closest_queries"] + s1["shadow_queries"]) / s1["paths"] < 11.5
    img = gpu_ctx.resolve(spp, accum=a).reshape(h, w, 3)
    assert np.isfinite(img).all() and (img >= 0).all() and img[:16].max() == 0.0
    assert 0.2 < img[1200:1500, 800:1200].mean() < 2.0   # the lit height field fills the lower half of the picture


def test_cli_renders_the_large_scene_configurations(tmp_path):
    """gpu_path_tracer on the bunny (streamed mode) and on the 1.0 M-triangle height field, as the
    benchmark configurations name them, at a reduced sample count."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "out" / "gpu").mkdir(parents=True)
    for scene, res, name in (("bunny", "1024", "Bunny"), ("heightfield:708", "2048", "Heightfield 708")):
        env = dict(os.environ, HPT_MODELS_DIR=os.path.join(root, "models"), HPT_SCENE=scene, HPT_WIDTH=res, HPT_HEIGHT=res)
        r = subprocess.run([os.path.join(root, "build", "gpu_path_tracer"), "4", "20", "256", "2"], cwd=str(tmp_path), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=110)
        assert r.returncode == 0, r.stdout
        assert "Resolution: %sx%s" % (res, res) in r.stdout and "HIP Mode: Streamed" in r.stdout
        data = (tmp_path / "out" / "gpu" / (name + ".ppm")).read_bytes()
        assert data.startswith(b"P6\n%s %s\n255\n" % (res.encode(), res.encode())) and len(data) > 3 * int(res) ** 2
        body = np.frombuffer(data[-3 * int(res) ** 2:], np.uint8)
        assert body.max() == 255 and 20 < body.mean() < 200
        # HPT_BVH=gpu / sah: the same picture from traversal records built on the device / from surface-area splits
        for build in ("gpu", "sah"):
            r = subprocess.run([os.path.join(root, "build", "gpu_path_tracer"), "4", "20", "256", "2"], cwd=str(tmp_path), env=dict(env, HPT_BVH=build),
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=110)
            assert r.returncode == 0, r.stdout
            assert (tmp_path / "out" / "gpu" / (name + ".ppm")).read_bytes() == data, build


@pytest.mark.parametrize("name", ["cornell", "heightfield4", "heightfield8", "bunny"])
@pytest.mark.parametrize("scheduler", ["all", "off"])
def test_wave_scheduler_and_lockstep_kernels_agree(hp, oracle, gpu_ctx, monkeypatch, name, scheduler):
    """Both kernel families on every traversal variant that has both: the wave scheduler (default for
    the two-child-record walks through L1/L2, forced here for the LDS variants too) and the lock-step
    loops (default for the LDS variants, forced here for the large scenes) -- images and ray counters
    equal the oracle's bit for bit in both modes."""
    from cudapathtracer_amd import scenes
    if name == "cornell":
        parts = scenes.glass_cornell_parts(glass=scenes.WHITE)
    else:
        parts = scenes.bunny_in_cornell_parts() if name == "bunny" else scenes.heightfield_in_cornell_parts(int(name[11:]))
    w, h, depth, spp = 52, 36, 7, 3
    sc = scenes._assemble(parts, w, h, depth)
    osc = _oracle_scene(oracle, parts, w, h)
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    if scheduler == "all":
        monkeypatch.setenv("HPT_MEGA_SCHEDULER_VARIANTS", "0xEC")
        monkeypatch.setenv("HPT_WF_SCHEDULER_VARIANTS", "0xEC")
    else:
        monkeypatch.setenv("HPT_SCHEDULER", "0")
    gpu_ctx.upload(sc)
    assert gpu_ctx.stats()["variant"] == {"cornell": 2, "heightfield4": 3, "heightfield8": 6, "bunny": 5}[name]
    for mode, block in ((1, 256), (2, 256), (1, 100), (2, 64)):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, block, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"], st["paths"]) == (ost["closest_queries"], ost["shadow_queries"], w * h * spp)
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), (mode, block)


def _sphere_scene(hp, oracle, w, h, depth, big=False):
    """Cornell shell + the short box mesh (or the bunny) + a near-mirror metal sphere, a glass sphere and a
    Lambertian sphere, in one top-level BVH; the same objects in the oracle."""
    from cudapathtracer_amd import scenes
    shell = scenes.cornell_shell_triangles()
    box = scenes.read_obj_triangles(os.path.join(scenes.DEFAULT_MODELS_DIR, "cornellbox", "shortbox.obj"))
    extra = ("bunny", scenes.bunny_triangles(), scenes.WHITE) if big else ("shortbox", box, scenes.WHITE)
    spheres = [((420, 90, 330), 90.0, dict(type=1, albedo=(.7, .5, .2), metallic=1.0, roughness=0.01)),
               ((150, 330, 250), 70.0, dict(type=2, roughness=0.1, ior=1.5)),
               ((300, 60, 120), 60.0, dict(type=0, albedo=(.3, .4, .8)))]
    sc = hp.Scene(w, h, camPos=scenes.CORNELL_CAM, fov=40.0, maxDepth=depth)
    osc = oracle.scene(w, h, scenes.CORNELL_CAM, 40.0)
    order = [shell[0], spheres[0], shell[1], shell[2], spheres[1], extra, shell[3], spheres[2]]
    for item in order:
        if len(item) == 3 and isinstance(item[0], str):
            sc.Add(item[1], hp.Material(**item[2]))
            osc.add_triangles(item[1], Material.make(**item[2]))
        else:
            sc.AddSphere(item[0], item[1], hp.Material(**item[2]))
            osc.add_sphere(item[0], item[1], Material.make(**item[2]))
    return sc.buildBVH(), osc.build()


@pytest.mark.parametrize("big", [False, True])
def test_spheres(hp, oracle, gpu_ctx, big):
    """Sphere primitives (common/Sphere.hpp is commented out in the reference: the oracle restates the
    commented text, parity unpinned except for solveQuadratic) among meshes: hits on closest and
    any-hit batches, images and ray counters in both modes, bit-exact against the oracle -- with the
    two-child records in LDS (24 primitives) and through L1/L2 under the wave scheduler (bunny)."""
    from cudapathtracer_amd import scenes
    w, h, depth, spp = 56, 44, 8, 3
    sc, osc = _sphere_scene(hp, oracle, w, h, depth, big)
    gpu_ctx.upload(sc)
    assert gpu_ctx.stats()["variant"] == (5 if big else 6)   # never a marking variant: those test triangles only
    f = sc.flat_arrays()
    sphere_prims = set(int(p) for p in f["prim_of_tri"][(f["tris"][:, 15].view(np.uint32) >> 31) == 1])
    o, d = primary_rays(160, 160, scenes.CORNELL_CAM, osc.info()["cam_scale"])
    hit = _same_hits(gpu_ctx, osc, o, d)
    assert all((hit == p).mean() > 0.005 for p in sphere_prims), "every sphere is in view"
    _same_hits(gpu_ctx, osc, *random_rays(1 << 16, 81))
    # rays that start inside the spheres (far root) and any-hit queries through them
    rs = np.random.RandomState(82)
    inside = (np.array([420, 90, 330]) + (rs.rand(4096, 3) - 0.5) * 80).astype(np.float32)
    dirs = rs.randn(4096, 3)
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    got = _same_hits(gpu_ctx, osc, inside, dirs)
    assert (got == min(sphere_prims)).mean() > 0.5   # (the bunny reaches into the metal sphere)
    _same_occlusion(gpu_ctx, osc, *_shadow_batch(1 << 15, 83))
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for mode in (1, 2):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 256, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode


@pytest.mark.parametrize("scene", ["glass_cornell", "bunny_metal"])
@pytest.mark.parametrize("rr,reg", [(True, False), (False, True), (True, True)])
def test_russian_roulette_and_path_regularisation(hp, oracle, gpu_ctx, scene, rr, reg):
    """The two steps of Scene::TracePath the reference switches off at compile time
    (common/Scene.hpp:141-142, :186-189, :252-264), as run-time options: images and counters bit-exact
    against the oracle's restatement in both modes, and different from the default integrator's."""
    from cudapathtracer_amd import scenes
    if scene == "glass_cornell":
        parts = scenes.glass_cornell_parts()
        w, h, depth, spp = 48, 40, 12, 4
    else:
        parts = scenes.cornell_shell_triangles() + [("bunny", scenes.bunny_triangles(), scenes.METAL)]
        w, h, depth, spp = 40, 32, 8, 2
    sc = scenes._assemble(parts, w, h, depth)
    osc = _oracle_scene(oracle, parts, w, h)
    plain, _ = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    osc.set_integrator(rr, reg)
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    assert not np.array_equal(plain, want_acc)
    gpu_ctx.upload(sc)
    gpu_ctx.set_integrator(rr, reg)
    try:
        for mode in (1, 2):
            gpu_ctx.stats_reset()
            gpu_ctx.accum_clear()
            gpu_ctx.render(mode, 256, depth, 0, spp)
            got = gpu_ctx.resolve(spp)
            st = gpu_ctx.stats()
            assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
            assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode
    finally:
        gpu_ctx.set_integrator(False, False)
    gpu_ctx.accum_clear()
    gpu_ctx.render(1, 256, depth, 0, spp)
    assert np.array_equal(gpu_ctx.resolve(spp).view(np.int32), oracle.resolve(plain, spp).view(np.int32))
    with pytest.raises(hp.HptError):
        hp.api._ck(gpu_ctx._L.hpt_set_integrator(gpu_ctx._h, 8))   # an unknown flag


def test_host_scene_intersect_equals_oracle(hp, oracle):
    """Scene::intersect of the C++ host class (common/Scene.hpp:95-97), which runs its queries on the GPU:
    happened / primitive / distance bit-exact against the oracle, coords = ray(distance), the hit triangle's normal,
    area and material; closest-hit and any-hit rays in one batch; a single-ray call; an empty batch."""
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5)
    osc = oracle.cornell()
    o1, d1 = random_rays(20000, seed=21)
    o2, d2, _, tm2 = shadow_rays(20000, seed=22)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    sh = np.concatenate([np.zeros(20000, np.uint8), np.ones(20000, np.uint8)])
    tm = np.concatenate([np.full(20000, 1e30), tm2])
    want_p, want_t = osc.intersect(o, d, sh, tm)
    got = sc.intersect(o, d, sh, tm)
    assert np.array_equal(got["prim"], want_p) and np.array_equal(got["happened"] != 0, want_p >= 0)
    assert np.array_equal(got["distance"].view(np.int64), want_t.view(np.int64))
    hit = want_p >= 0
    coords = (o + d * want_t.astype(np.float32)[:, None]).astype(np.float32)
    assert np.array_equal(got["coords"][hit].view(np.uint32), coords[hit].view(np.uint32))
    fa = sc.flat_arrays()
    slot = np.full(fa["prim_of_tri"].max() + 1, -1)
    slot[fa["prim_of_tri"]] = np.arange(len(fa["prim_of_tri"]))
    T = fa["tris"][slot[want_p[hit]]]
    assert np.array_equal(got["normal"][hit].view(np.uint32), T[:, 12:15].view(np.uint32))
    assert np.array_equal(got["triangleArea"][hit].view(np.uint32), T[:, 3].view(np.uint32))
    mat = T[:, 7].view(np.uint32)
    assert np.array_equal(got["material_type"][hit], fa["materials"][mat, 0].astype(np.int32))
    assert (got["material_type"][~hit] == -1).all()
    one = sc.intersect(o[:1], d[:1])
    assert one["prim"][0] == want_p[0] and one["distance"][0] == want_t[0]
    assert len(sc.intersect(np.zeros((0, 3)), np.zeros((0, 3)))["prim"]) == 0


def test_host_scene_cast_ray_equals_oracle(hp, oracle):
    """Scene::castRay of the C++ host class (common/Scene.hpp:274-285), which runs the paths on the GPU: radiance
    bit-exact against the oracle for random eye rays and keys, with another seed and Russian roulette too; a
    single-ray call; an empty batch; mismatched lengths."""
    depth = 7
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth)
    osc = oracle.cornell()
    n = 20000
    o, d = random_rays(n, seed=61)
    rs = np.random.RandomState(62)
    pk, sk = rs.randint(0, 1 << 18, n), rs.randint(0, 256, n)
    want, _ = osc.cast_rays(RNG_PHILOX, TRIG_PORTABLE, depth, o, d, sk, pk)
    got = sc.castRay(o, d, pk, sk)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want.max() > 0
    one = sc.castRay(o[5:6], d[5:6], pk[5:6], sk[5:6])
    assert np.array_equal(one.view(np.uint32), want[5:6].view(np.uint32))
    osc.set_integrator(True, False)
    want2, _ = osc.cast_rays(RNG_PHILOX, TRIG_PORTABLE, depth, o, d, sk, pk, seed=9)
    got2 = sc.castRay(o, d, pk, sk, seed=9, integrator=hp.api.INTEGRATOR_RUSSIAN_ROULETTE)
    assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32)) and not np.array_equal(want2, want)
    assert np.array_equal(sc.castRay(o, d, pk, sk).view(np.uint32), want.view(np.uint32))   # the options do not stick
    assert sc.castRay(np.zeros((0, 3)), np.zeros((0, 3)), [], []).shape == (0, 3)
    with pytest.raises(ValueError):
        sc.castRay(o[:4], d[:4], pk[:3], sk[:4])


def _hits_equal_up_to_ties(got, want, what):
    """Closest hits of a tree other than the reference's over the same leaf boxes: every distance is the
    reference's bit for bit; the primitive may differ only where two are hit at that very distance."""
    gp, gt = got
    wp, wt = want
    assert np.array_equal(gt.view(np.int64), wt.view(np.int64)), "%s: %d distances differ" % (what, (gt != wt).sum())
    assert np.array_equal(gp >= 0, wp >= 0)
    return int((gp != wp).sum())


@pytest.mark.parametrize("name", ["bunny", "heightfield128", "heightfield8", "spheres"])
def test_gpu_bvh_build(hp, oracle, gpu_ctx, name):
    """hpt_set_bvh_build(HPT_BVH_BUILD_GPU): the traversal records are built on the device from the triangle array
    (median split by centroid, level by level, one tree for the scene; the host builds only the light's tree).
    Closest distances, occlusion answers, ray counters and rendered images equal the oracle's bit for bit; a
    primitive id may differ only between triangles hit at bitwise the same distance."""
    from cudapathtracer_amd import scenes
    w, h, depth, spp = 56, 40, 6, 2
    if name == "spheres":
        sc, osc = _sphere_scene(hp, oracle, w, h, depth, big=True)
        sc.buildBVH(gpu=True)
    else:
        parts = scenes.bunny_in_cornell_parts() if name == "bunny" else scenes.heightfield_in_cornell_parts(int(name[11:]))
        sc = hp.Scene(w, h, camPos=scenes.CORNELL_CAM, fov=40.0, maxDepth=depth)
        for _, tris, mat in parts:
            sc.Add(tris, hp.Material(**mat))
        sc.buildBVH(gpu=True)
        osc = _oracle_scene(oracle, parts, w, h)
    assert sc.gpu_build
    fa = sc.flat_arrays()
    assert np.array_equal(fa["prim_of_tri"], np.arange(len(fa["prim_of_tri"])))  # primitive-id order
    gpu_ctx.upload(sc)
    info = gpu_ctx.build_info()
    assert info["records"] == len(fa["prim_of_tri"]) - 1 and info["gpu_build_ms"] > 0 and 1 <= info["stack_depth"] <= 40
    assert gpu_ctx.stats()["variant"] in (5, 6)
    ties = 0
    o, d = primary_rays(192, 192, scenes.CORNELL_CAM, osc.info()["cam_scale"])
    ties += _hits_equal_up_to_ties(gpu_ctx.intersect(o, d), osc.intersect(o, d), "primary")
    o, d = random_rays(1 << 17, 41)
    ties += _hits_equal_up_to_ties(gpu_ctx.intersect(o, d), osc.intersect(o, d), "random")
    assert ties <= 4, "%d primitive ids differ at equal distances" % ties
    _same_occlusion(gpu_ctx, osc, *_shadow_batch(1 << 16, 42, down=name.startswith("heightfield")))
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for mode in (1, 2):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 256, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode
    # the host build of the same scene object gives the reference's tree again
    sc.buildBVH(gpu=False)
    assert not sc.gpu_build
    gpu_ctx.upload(sc)
    assert gpu_ctx.build_info()["gpu_build_ms"] == 0
    _same_hits(gpu_ctx, osc, *random_rays(1 << 14, 43))


def test_gpu_bvh_build_of_the_million_triangle_scene(hp, gpu_ctx):
    """The 1.0 M-triangle height field with its records built on the device: the build takes milliseconds, and
    closest hits / occlusion answers equal those of the host-built reference tree on the same GPU."""
    import time
    t0 = time.time()
    sc_gpu = hp.Scene.CreateNamedScene("heightfield:708", 5, 64, 64, bvh_build=hp.BVH_BUILD_GPU)
    t_gpu_host_part = time.time() - t0
    gpu_ctx.upload(sc_gpu)
    info = gpu_ctx.build_info()
    assert info["records"] == len(sc_gpu.flat_arrays()["prim_of_tri"]) - 1 > 1_000_000
    assert 0 < info["gpu_build_ms"] < 500, info
    o, d = random_rays(1 << 18, 51)
    so, sd, sh, tm = _shadow_batch(1 << 17, 52, down=True)
    got = gpu_ctx.intersect(o, d)
    got_s = gpu_ctx.intersect(so, sd, sh, tm)
    gpu_ctx.accum_clear()
    gpu_ctx.render(2, 256, 5, 0, 1)
    img_gpu = gpu_ctx.accum_download()
    sc_ref = hp.Scene.CreateNamedScene("heightfield:708", 5, 64, 64)
    gpu_ctx.upload(sc_ref)
    want = gpu_ctx.intersect(o, d)
    want_s = gpu_ctx.intersect(so, sd, sh, tm)
    # primitive ids: the reference tree reports leaf-order-mapped global ids; both are global primitive ids
    assert _hits_equal_up_to_ties(got, want, "1M random") <= 8
    assert np.array_equal(got_s[0] >= 0, want_s[0] >= 0)
    gpu_ctx.accum_clear()
    gpu_ctx.render(2, 256, 5, 0, 1)
    assert np.array_equal(img_gpu, gpu_ctx.accum_download())
    print("host part of the GPU-build scene %.2f s, device build %.2f ms" % (t_gpu_host_part, info["gpu_build_ms"]))


@pytest.mark.parametrize("name", ["no_light", "degenerate", "two_triangles", "many_objects", "one_triangle"])
def test_gpu_bvh_build_edges(hp, oracle, gpu_ctx, name):
    """The device-side build at the edges of the scene description: no emitter (no sampling tree at all), zero-area
    triangles (NaN normals, NaN-free boxes), the smallest scene it accepts, a scene of many small objects (the tree
    over the objects carries most of the work), and the one it refuses (a single primitive)."""
    from cudapathtracer_amd import scenes
    light = dict(type=0, emission=scenes.light_emission())
    quad = np.array([[[100, 500, 100], [400, 500, 100], [400, 500, 400]],
                     [[100, 500, 100], [400, 500, 400], [100, 500, 400]]], np.float32)[:, ::-1]
    if name == "no_light":
        parts = [("a", _soup(5, 1), scenes.WHITE), ("b", _soup(3, 2), scenes.METAL)]
    elif name == "degenerate":
        t = _soup(6, 3)
        t[1, 2] = t[1, 1]
        t[4] = t[4, 0]
        parts = [("light", quad, light), ("a", t, scenes.WHITE)]
    elif name == "two_triangles":
        parts = [("light", quad, light)]
    elif name == "one_triangle":
        parts = [("light", quad[:1], light)]
    else:
        parts = [("light", quad, light)] + [("o%d" % k, _soup(1 + k % 5, 100 + k, size=60.0), scenes.WHITE if k % 2 else scenes.METAL) for k in range(37)]
    w, h, depth, spp = 24, 16, 4, 3
    sc = hp.Scene(w, h, camPos=scenes.CORNELL_CAM, fov=40.0, maxDepth=depth)
    for _, tris, mat in parts:
        sc.Add(tris, hp.Material(**mat))
    sc.buildBVH(gpu=True)
    if name == "one_triangle":
        with pytest.raises(hp.HptError, match="at least two primitives"):
            gpu_ctx.upload(sc)
        return
    osc = _oracle_scene(oracle, parts, w, h)
    gpu_ctx.upload(sc)
    assert gpu_ctx.stats()["variant"] == 6 and gpu_ctx.build_info()["records"] == sum(len(p[1]) for p in parts) - 1
    o, d = primary_rays(64, 64, scenes.CORNELL_CAM, osc.info()["cam_scale"])
    assert _hits_equal_up_to_ties(gpu_ctx.intersect(o, d), osc.intersect(o, d), "primary") == 0
    assert _hits_equal_up_to_ties(gpu_ctx.intersect(*random_rays(1 << 14, 61)), osc.intersect(*random_rays(1 << 14, 61)), "random") == 0
    _same_occlusion(gpu_ctx, osc, *_shadow_batch(1 << 13, 62))
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for mode in (1, 2):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 64, depth, 0, spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
        assert np.array_equal(gpu_ctx.resolve(spp).view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode
    # the entry point that allocates nothing runs on the device-built records as well
    gpu_ctx.accum_clear()
    gpu_ctx.render_static(2, 64, depth, 0, spp)
    assert np.array_equal(gpu_ctx.resolve(spp).view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))


def test_gpu_bvh_build_random_soups(hp, gpu_ctx):
    """Device-built records against the host-built reference tree on the same GPU, over soups of many sizes
    (odd, even, powers of two and their neighbours), with coincident triangles (every centroid ties) and objects of
    one primitive: closest distances bit-equal, primitive ids equal except at equal distances, occlusion equal."""
    from cudapathtracer_amd import scenes
    rs = np.random.RandomState(77)
    o, d = random_rays(1 << 13, 78)
    so, sd, sh, tm = _shadow_batch(1 << 12, 79)
    for case, n in enumerate([2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 100, 255, 256, 257, 1000, 4097]):
        soup = _soup(n, 200 + case, size=120.0)
        if case % 4 == 1:
            soup[n // 2:] = soup[0]                     # half of the triangles coincide
        parts = [("a", soup[: max(1, n // 3)], scenes.WHITE), ("b", soup[max(1, n // 3):], scenes.METAL)] if n >= 2 else []
        parts = [p for p in parts if len(p[1])]
        results = []
        for gpu in (False, True):
            sc = hp.Scene(16, 16, camPos=scenes.CORNELL_CAM, fov=40.0, maxDepth=3)
            for _, tris, mat in parts:
                sc.Add(tris, hp.Material(**mat))
            sc.buildBVH(gpu=gpu)
            gpu_ctx.upload(sc)
            results.append((gpu_ctx.intersect(o, d), gpu_ctx.intersect(so, sd, sh, tm)))
        (hp_c, ht_c), (hp_s, _) = results[0]
        (gp_c, gt_c), (gp_s, _) = results[1]
        assert np.array_equal(gt_c.view(np.int64), ht_c.view(np.int64)), "n = %d" % n
        differ = gp_c != hp_c
        if differ.any():   # only between coincident triangles: the other tree's choice is hit at the same distance
            assert case % 4 == 1, "n = %d: %d ids differ" % (n, differ.sum())
        assert np.array_equal(gp_s >= 0, hp_s >= 0), "n = %d" % n


@pytest.mark.parametrize("gpu_build", [False, True, "sah"])
def test_bunny_full_resolution_render_matches_oracle_bit_exact(hp, oracle, gpu_ctx, gpu_build):
    """The bunny configuration at its benchmark resolution and depth (1024 x 1024, depth 20, streamed mode under the
    wave scheduler), 1 spp = 11 M rays: every accumulator and both ray counters equal the oracle's, on the host-built
    tree, on device-built records and on the host tree with surface-area splits."""
    from cudapathtracer_amd import scenes
    parts = scenes.bunny_in_cornell_parts()
    w = h = 1024
    depth, spp = 20, 1
    sc = hp.Scene(w, h, camPos=scenes.CORNELL_CAM, fov=40.0, maxDepth=depth)
    for _, tris, mat in parts:
        sc.Add(tris, hp.Material(**mat))
    sc.buildBVH(sah=True) if gpu_build == "sah" else sc.buildBVH(gpu=gpu_build)
    osc = _oracle_scene(oracle, parts, w, h)
    gpu_ctx.upload(sc)
    gpu_ctx.stats_reset()
    gpu_ctx.accum_clear()
    gpu_ctx.render(2, 256, depth, 0, spp)
    got = gpu_ctx.accum_download()
    st = gpu_ctx.stats()
    want, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, "%d of %d pixels differ" % (bad.size, w * h)


@pytest.mark.parametrize("name", ["spheres", "spheres_bunny", "bunny_metal_rr_reg", "heightfield4", "heightfield128"])
def test_cast_rays_on_other_scenes(hp, oracle, gpu_ctx, name):
    """hpt_cast_rays (Scene::castRay for caller-supplied eye rays) through the other kernels: glass and spheres with
    the two-child records in LDS and through L1/L2, Russian roulette + regularisation on a metal bunny, the 64-bit
    marking variant, a height field; radiance floats and counters bit-exact against the oracle."""
    from cudapathtracer_amd import scenes
    w, h, depth, n = 32, 32, 9, 1 << 14
    rr = reg = False
    if name.startswith("spheres"):
        sc, osc = _sphere_scene(hp, oracle, w, h, depth, big=name.endswith("bunny"))
    else:
        if name.startswith("bunny"):
            parts = scenes.cornell_shell_triangles() + [("bunny", scenes.bunny_triangles(), scenes.METAL)]
            rr = reg = True
        else:
            parts = scenes.heightfield_in_cornell_parts(int(name[11:]))
        sc = scenes._assemble(parts, w, h, depth)
        osc = _oracle_scene(oracle, parts, w, h)
    osc.set_integrator(rr, reg)
    gpu_ctx.upload(sc)
    gpu_ctx.set_integrator(rr, reg)
    try:
        o, d = random_rays(n, 91)
        rs = np.random.RandomState(92)
        pk, sk = rs.randint(0, 1 << 24, n), rs.randint(0, 1 << 16, n)
        want, ost = osc.cast_rays(RNG_PHILOX, TRIG_PORTABLE, depth, o, d, sk, pk)
        gpu_ctx.stats_reset()
        got = gpu_ctx.cast_rays(o, d, pk, sk, depth)
        st = gpu_ctx.stats()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert (st["closest_queries"], st["shadow_queries"], st["paths"]) == (ost["closest_queries"], ost["shadow_queries"], n)
        assert want.max() > 0
    finally:
        gpu_ctx.set_integrator(False, False)


def test_random_scenes_sample():
    """A sample of the randomised differential test (tools/fuzz_parity.py: random scenes, materials, sizes, depths,
    sample ranges, seeds, modes, workgroup sizes, integrator options and BVH builders; accumulators, counters, hits,
    occlusion and castRay radiance bit-exact against the oracle).  The tool itself runs thousands of cases."""
    import io
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    log = io.StringIO()
    fuzz_parity.run(120, first=20000, out=log)
    assert log.getvalue().count("  ok") == 120


@pytest.mark.parametrize("name", ["bunny", "heightfield128", "heightfield8", "heightfield4", "spheres", "glass"])
def test_sah_build_renders_the_reference_results(hp, oracle, gpu_ctx, name):
    """Scene::BvhBuild::Sah (surface-area splits on the host for the top level and the non-emitting meshes, the
    reference's leaves): closest distances, occlusion answers, ray counters and rendered images equal the oracle's bit
    for bit in every traversal variant, primitive ids included (ties at equal distance go to the triangle the reference
    names, because the records keep the reference's order); the traversal stack is no deeper than the reference tree's."""
    from cudapathtracer_amd import scenes
    w, h, depth, spp = 56, 40, 8, 2
    if name == "spheres":
        sc, osc = _sphere_scene(hp, oracle, w, h, depth, big=True)
    else:
        parts = (scenes.bunny_in_cornell_parts() if name == "bunny" else scenes.glass_cornell_parts() if name == "glass"
                 else scenes.heightfield_in_cornell_parts(int(name[11:])))
        sc = hp.Scene(w, h, camPos=scenes.CORNELL_CAM, fov=40.0, maxDepth=depth)
        for _, tris, mat in parts:
            sc.Add(tris, hp.Material(**mat))
        osc = _oracle_scene(oracle, parts, w, h)
    sc.buildBVH(gpu=False)   # the reference's tree alone
    assert sc.bvh_build == hp.BVH_BUILD_HOST
    gpu_ctx.upload(sc)
    ref_info, ref_variant = gpu_ctx.build_info(), gpu_ctx.stats()["variant"]
    sc.buildBVH(sah=True)
    assert sc.bvh_build == hp.BVH_BUILD_SAH
    gpu_ctx.upload(sc)
    info = gpu_ctx.build_info()
    assert gpu_ctx.stats()["variant"] == ref_variant and info["gpu_build_ms"] == 0
    # the two-child-record walks hold the records of both trees (the reference's behind the surface-area tree's, for
    # the rays with a zero direction component); the stack serves the deeper of the two, i.e. the reference's
    assert info["records"] == 2 * ref_info["records"]
    assert info["stack_depth"] <= ref_info["stack_depth"], (info, ref_info)
    # ids too: the records lie in the reference's traversal order, so ties go to the triangle the reference names
    o, d = primary_rays(192, 192, scenes.CORNELL_CAM, osc.info()["cam_scale"])
    _same_hits(gpu_ctx, osc, o, d)
    _same_hits(gpu_ctx, osc, *random_rays(1 << 17, 41))
    if name != "spheres":
        zo, zd = _zero_component_rays(parts, 1 << 15, 43)
        _same_hits(gpu_ctx, osc, zo, zd)
        (_same_hits if ref_variant in (2, 3) else _same_occlusion)(gpu_ctx, osc, zo, zd, np.ones(len(zo), np.uint8), np.full(len(zo), 250.0))
    _same_occlusion(gpu_ctx, osc, *_shadow_batch(1 << 16, 42, down=name.startswith("heightfield")))
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for mode in (1, 2):
        gpu_ctx.stats_reset()
        gpu_ctx.accum_clear()
        gpu_ctx.render(mode, 256, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), "mode %d" % mode


def test_sah_build_of_the_million_triangle_scene(hp, gpu_ctx):
    """The 1.0 M-triangle height field with surface-area splits: closest distances, occlusion answers and a rendered
    image equal those of the reference tree on the same GPU; fewer boxes are tested (the counting build says how many
    in tools/count_tests.py; here: the stack is no deeper)."""
    sc = hp.Scene.CreateNamedScene("heightfield:708", 5, 64, 64, bvh_build=hp.BVH_BUILD_SAH)
    gpu_ctx.upload(sc)
    info = gpu_ctx.build_info()
    o, d = random_rays(1 << 18, 51)
    so, sd, sh, tm = _shadow_batch(1 << 17, 52, down=True)
    got, got_s = gpu_ctx.intersect(o, d), gpu_ctx.intersect(so, sd, sh, tm)
    gpu_ctx.accum_clear()
    gpu_ctx.render(2, 256, 5, 0, 1)
    img = gpu_ctx.accum_download()
    gpu_ctx.upload(hp.Scene.CreateNamedScene("heightfield:708", 5, 64, 64, bvh_build=hp.BVH_BUILD_HOST))
    assert info["stack_depth"] <= gpu_ctx.build_info()["stack_depth"] and info["records"] == 2 * gpu_ctx.build_info()["records"]
    assert _hits_equal_up_to_ties(got, gpu_ctx.intersect(o, d), "1M random") == 0
    assert np.array_equal(got_s[0] >= 0, gpu_ctx.intersect(so, sd, sh, tm)[0] >= 0)
    gpu_ctx.accum_clear()
    gpu_ctx.render(2, 256, 5, 0, 1)
    assert np.array_equal(img, gpu_ctx.accum_download())
