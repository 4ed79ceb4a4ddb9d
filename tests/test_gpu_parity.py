"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on identical inputs.

Bars: primitive ids, hit distances (float64 bits), any-hit results and the
fixed-point radiance accumulators are compared BIT-EXACT -- the device
arithmetic is specified operation by operation, so no tolerance is needed or
used.  north_star's own bar for t is 1e-5 relative; it is asserted as well so
that a failure of the stricter check still reports against the contract.
"""
import numpy as np
import pytest

from orclib import (RNG_PHILOX, TRIG_PORTABLE, primary_rays, random_rays, shadow_rays)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cornell(hp, oracle, gpu_ctx):
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5)
    osc = oracle.cornell()
    gpu_ctx.upload(sc)
    return sc, osc, gpu_ctx


def _check_hits(got, want, what):
    gp, gt = got
    wp, wt = want
    bad = np.flatnonzero(gp != wp)
    assert bad.size == 0, "%s: %d primitive ids differ, first ray %d: gpu %d oracle %d" % (
        what, bad.size, bad[0], gp[bad[0]], wp[bad[0]])
    rel = np.abs(gt - wt) / np.maximum(np.abs(wt), 1e-30)
    assert rel.max() <= 1e-5, "%s: t differs by %g relative" % (what, rel.max())
    bad = np.flatnonzero(gt.view(np.int64) != wt.view(np.int64))
    assert bad.size == 0, "%s: %d hit distances differ in bits (max rel %g)" % (what, bad.size, rel.max())


def test_intersect_primary_grid(cornell):
    sc, osc, ctx = cornell
    o, d = primary_rays(512, 512, (278, 273, -800), osc.info()["cam_scale"])
    want = osc.intersect(o, d)
    assert (want[0] >= 0).sum() == 234737  # known answer of the compiled reference
    _check_hits(ctx.intersect(o, d), want, "primary grid")


def test_intersect_random_closest(cornell):
    sc, osc, ctx = cornell
    o, d = random_rays(1 << 18, seed=1)
    _check_hits(ctx.intersect(o, d), osc.intersect(o, d), "random closest")


def test_intersect_shadow(cornell):
    sc, osc, ctx = cornell
    o, d, sh, tm = shadow_rays(1 << 18, seed=2)
    want = osc.intersect(o, d, sh, tm)
    assert 0 < (want[0] >= 0).mean() < 1
    _check_hits(ctx.intersect(o, d, sh, tm), want, "shadow")


def test_intersect_mixed_and_degenerate(cornell):
    """closest and any-hit queries interleaved in one batch; axis-parallel
    directions (infinite reciprocal, NaN slab products); rays starting on surfaces."""
    sc, osc, ctx = cornell
    o1, d1 = random_rays(4096, seed=3)
    o2, d2, _, tm2 = shadow_rays(4096, seed=4)
    o = np.concatenate([o1, o2])
    d = np.concatenate([d1, d2])
    sh = np.concatenate([np.zeros(4096, np.uint8), np.ones(4096, np.uint8)])
    tm = np.concatenate([np.zeros(4096), tm2])
    perm = np.random.RandomState(5).permutation(len(o))
    o, d, sh, tm = o[perm], d[perm], sh[perm], tm[perm]
    _check_hits(ctx.intersect(o, d, sh, tm), osc.intersect(o, d, sh, tm), "mixed")
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, -0.0, 1], [1, 1, 0]], np.float32)
    axes[-1] /= np.float32(np.sqrt(2))
    rs = np.random.RandomState(6)
    o = (rs.rand(8 * 512, 3) * [556, 548.8, 559.2]).astype(np.float32)
    o[::7, 1] = 0.0      # on the floor plane
    o[::11, 0] = 0.0     # on the right wall plane
    d = np.tile(axes, (512, 1))
    _check_hits(ctx.intersect(o, d), osc.intersect(o, d), "axis-parallel")


def test_intersect_edges_vertices_and_limits(cornell):
    """The cases the division-free triangle classification hands to the exact sequence: rays aimed at points
    ON triangle edges, shared diagonals and vertices (u, v, u + v within rounding of 0 or 1; two triangles with
    near-equal t), and any-hit queries whose tmax lies within a few ulps / 1e-4 of a hit distance."""
    sc, osc, ctx = cornell
    fa = sc.flat_arrays()
    T = fa["tris"]
    v0, e1, e2 = T[:, 0:3], T[:, 4:7], T[:, 8:11]
    rs = np.random.RandomState(11)
    n = 1 << 15
    tri = rs.randint(0, len(T), n)
    s = rs.rand(n).astype(np.float32)[:, None]
    kind = rs.randint(0, 5, n)[:, None]
    target = np.where(kind == 0, v0[tri] + s * e1[tri],                       # edge v0-v1
             np.where(kind == 1, v0[tri] + s * e2[tri],                       # edge v0-v2
             np.where(kind == 2, v0[tri] + e1[tri] + s * (e2[tri] - e1[tri]),  # edge v1-v2 (the rectangles' diagonals)
             np.where(kind == 3, v0[tri] + np.where(s < 0.5, e1[tri], e2[tri]), v0[tri])))).astype(np.float32)
    o = (rs.rand(n, 3) * [556, 548.8, 559.2]).astype(np.float32)
    v = (target - o).astype(np.float32)
    ln = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2], dtype=np.float32)
    keep = ln > 1e-3
    o, v, ln = o[keep], v[keep], ln[keep]
    d = (v * (np.float32(1) / ln)[:, None]).astype(np.float32)
    want = osc.intersect(o, d)
    _check_hits(ctx.intersect(o, d), want, "edges and vertices, closest")
    # any-hit with tmax around the closest hit's distance: the occlusion rule t < tmax - 1e-4 at its boundary
    hit = want[0] >= 0
    o, d, t = o[hit], d[hit], want[1][hit]
    deltas = np.array([0.0, 1e-4, 9.99e-5, 1.001e-4, 1e-4 + 1e-9, 1e-4 - 1e-9, 2e-4, 5e-5, -1e-4, 1e-3, -1e-3])
    tm = (t[:, None] + deltas[None, :]).astype(np.float32).astype(np.float64)  # Ray::shadowDistance holds a float's value
    o = np.repeat(o, len(deltas), axis=0)
    d = np.repeat(d, len(deltas), axis=0)
    tm = tm.reshape(-1)
    sh = np.ones(len(o), np.uint8)
    want = osc.intersect(o, d, sh, tm)
    assert 0.05 < (want[0] >= 0).mean() < 0.95
    _check_hits(ctx.intersect(o, d, sh, tm), want, "any-hit at the limit")


def test_intersect_empty_batch(cornell):
    sc, osc, ctx = cornell
    p, t = ctx.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert p.shape == (0,) and t.shape == (0,)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("depth,spp", [(5, 4), (20, 3)])
def test_render_matches_oracle_bit_exact(hp, oracle, gpu_ctx, mode, depth, spp):
    w = h = 96
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    osc = oracle.cornell(w, h)
    gpu_ctx.upload(sc)
    gpu_ctx.stats_reset()
    gpu_ctx.accum_clear()
    gpu_ctx.render(mode, 256, depth, 0, spp)
    got = gpu_ctx.resolve(spp)
    st = gpu_ctx.stats()
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    want = oracle.resolve(want_acc, spp)
    assert st["paths"] == ost["paths"] == w * h * spp
    assert st["closest_queries"] == ost["closest_queries"]
    assert st["shadow_queries"] == ost["shadow_queries"]
    bad = np.flatnonzero((got.view(np.int32) != want.view(np.int32)).any(axis=1))
    assert bad.size == 0, "%d of %d pixels differ, max abs diff %g" % (bad.size, w * h, np.abs(got - want).max())


def test_accumulators_bit_exact_and_split_invariant(hp, oracle, gpu_ctx):
    """raw int64 accumulators == oracle's; [0,6) == [0,2) + [2,6); mode 1 == mode 2."""
    import ctypes as C
    w, h, depth, spp = 64, 48, 8, 6
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    gpu_ctx.upload(sc)
    osc = oracle.cornell(w, h)
    want, _ = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)

    def download():
        return gpu_ctx.accum_download()

    gpu_ctx.accum_clear()
    gpu_ctx.render(1, 256, depth, 0, spp)
    a = download()
    assert np.array_equal(a, want)
    gpu_ctx.accum_clear()
    gpu_ctx.render(1, 128, depth, 0, 2)
    gpu_ctx.render(2, 64, depth, 2, 4)
    assert np.array_equal(download(), want)


@pytest.mark.parametrize("block", [64, 100, 512, 1024])
@pytest.mark.parametrize("mode", [1, 2])
def test_block_sizes_give_identical_images(hp, oracle, gpu_ctx, monkeypatch, block, mode):
    # workgroups of exactly `block` lanes (LDS-resident scenes would otherwise run with at least 256)
    monkeypatch.setenv("HPT_EXACT_BLOCK", "1")
    w, h, depth, spp = 40, 24, 6, 2   # 40x24: partial 8x8 tiles on neither edge... 24 = 3 tiles, 40 = 5 tiles
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w + 3, h + 5)  # ragged edges
    osc = oracle.cornell(w + 3, h + 5)
    gpu_ctx.upload(sc)
    gpu_ctx.accum_clear()
    gpu_ctx.stats_reset()
    gpu_ctx.render(mode, block, depth, 0, spp)
    got = gpu_ctx.resolve(spp)
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))
    st = gpu_ctx.stats()    # counters are reduced per wave: a partial last wave (block 100) must count right
    assert (st["closest_queries"], st["shadow_queries"], st["paths"]) == \
        (ost["closest_queries"], ost["shadow_queries"], ost["paths"])


def test_sample_chunks_and_spare_waves(hp, oracle, gpu_ctx):
    """mode 1 splits a launch into (tile, chunk of 8 samples) units: 70 samples = 8 chunks of 8 and one of 6;
    waves draw units from a counter and hold paths of two units at a time."""
    w, h, depth, spp = 40, 24, 4, 70
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    osc = oracle.cornell(w, h)
    gpu_ctx.upload(sc)
    want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    for block in (128, 1024):    # 128 is run as 256: the scene is staged in LDS per workgroup
        gpu_ctx.accum_clear()
        gpu_ctx.stats_reset()
        gpu_ctx.render(1, block, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        assert (st["closest_queries"], st["shadow_queries"], st["paths"]) == \
            (ost["closest_queries"], ost["shadow_queries"], ost["paths"])
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))


@pytest.mark.parametrize("exact_block", ["1", "0"])
@pytest.mark.parametrize("mode", [1, 2])
def test_extreme_shapes(hp, oracle, gpu_ctx, monkeypatch, mode, exact_block):
    """a one-pixel image with one lane per workgroup, a one-row image, depth 1, and a sample range
    that ends at the top of the 32-bit sample index space; with the workgroup size as given and
    widened to 256."""
    monkeypatch.setenv("HPT_EXACT_BLOCK", exact_block)
    for (w, h, block, depth, s0, spp) in ((1, 1, 1, 3, 0, 5), (19, 1, 7, 1, 0, 3), (9, 10, 64, 4, 0xFFFFFFF0, 15)):
        sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
        osc = oracle.cornell(w, h)
        gpu_ctx.upload(sc)
        gpu_ctx.accum_clear()
        gpu_ctx.stats_reset()
        gpu_ctx.render(mode, block, depth, s0, spp)
        got = gpu_ctx.resolve(spp)
        st = gpu_ctx.stats()
        want_acc, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, sample_begin=s0, fixed=True)
        assert (st["closest_queries"], st["shadow_queries"], st["paths"]) == \
            (ost["closest_queries"], ost["shadow_queries"], ost["paths"]), (w, h, block)
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32)), (w, h, block)


def test_cpp_hiprenderer_equals_c_abi(hp, oracle, gpu_ctx):
    """the C++ HipRenderer class (PrepareRender/Render/FinishRender) and the Python mirror."""
    w = h = 64
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5, w, h)
    osc = oracle.cornell(w, h)
    want_acc, _ = osc.render(RNG_PHILOX, TRIG_PORTABLE, 5, 4, fixed=True)
    want = oracle.resolve(want_acc, 4)
    for mode in (1, 2):
        rgb, st = sc.render_cpp(4, 5, 256, mode)
        assert st["paths"] == w * h * 4
        assert np.array_equal(rgb.view(np.int32), want.view(np.int32))
        # PrepareRender, PathTrace(scene, iter) x spp, FinishRender: one download at the end
        rgb, st = sc.render_cpp(4, 5, 256, mode, iterative=True)
        assert st["paths"] == w * h * 4 and st["launches"] == (4 if mode == 1 else 8)
        assert np.array_equal(rgb.view(np.int32), want.view(np.int32))
    r = hp.HipRenderer()
    r.spp = 4
    r.SetMode(hp.Streamed)
    r.PrepareRender(sc)
    fb = r.Render(sc)
    assert np.array_equal(fb.view(np.int32), want.view(np.int32))
    # PathTrace(scene, iter): one sample per call on top of a cleared buffer
    r.PrepareRender(sc)
    for it in range(4):
        r.PathTrace(sc, it)
    assert np.array_equal(r.framebuffer.view(np.int32), want.view(np.int32))
    r.FinishRender(sc)


def test_many_renders_without_reading_stats_hold_few_events(hp, oracle, gpu_ctx):
    """A PathTrace-style loop (one hpt_render per sample, statistics never read in between): the
    timing events of completed renders are recycled, kernel time and launch count still add up."""
    w, h, depth, n = 32, 32, 4, 300
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    gpu_ctx.upload(sc)
    gpu_ctx.stats_reset()
    gpu_ctx.accum_clear()
    for it in range(n):
        gpu_ctx.render(1 + it % 2, 256, depth, it, 1)
    st = gpu_ctx.stats()
    assert st["paths"] == w * h * n and st["launches"] == n // 2 + 2 * (n // 2) and st["kernel_ms"] > 0
    want_acc, _ = oracle.cornell(w, h).render(RNG_PHILOX, TRIG_PORTABLE, depth, n, fixed=True)
    assert np.array_equal(gpu_ctx.resolve(n).view(np.int32), oracle.resolve(want_acc, n).view(np.int32))


def test_renders_on_two_streams_of_one_context_are_ordered(hp, oracle, gpu_ctx):
    """The unit counter and the wavefront queues belong to the context; a render queued on another
    stream waits for the one before it instead of racing with it."""
    import torch
    w, h, depth = 64, 64, 8
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    gpu_ctx.upload(sc)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    acc = torch.zeros(3 * w * h, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    for k in range(6):
        gpu_ctx.render(1 + k % 2, 256, depth, 4 * k, 4, accum=acc, stream=(s1 if k % 2 == 0 else s2))
    torch.cuda.synchronize()
    want_acc, _ = oracle.cornell(w, h).render(RNG_PHILOX, TRIG_PORTABLE, depth, 24, fixed=True)
    assert np.array_equal(acc.cpu().numpy().reshape(-1, 3), want_acc)


def test_wavefront_small_capacity_batches(hp, oracle, gpu_ctx):
    """streamed mode with a queue far smaller than the number of paths"""
    w, h, depth, spp = 72, 40, 7, 5
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    gpu_ctx.upload(sc)
    gpu_ctx.set_wavefront_capacity(4096)
    try:
        gpu_ctx.accum_clear()
        gpu_ctx.render(2, 256, depth, 0, spp)
        got = gpu_ctx.resolve(spp)
    finally:
        gpu_ctx.set_wavefront_capacity(1 << 28)
    want_acc, _ = oracle.cornell(w, h).render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))


@pytest.mark.parametrize("generations", ["1", "3", "16", "200"])
def test_wavefront_segment_generations(hp, oracle, gpu_ctx, monkeypatch, generations):
    """The streamed mode deals a batch's paths to more queue segments than workgroups are resident at once (16 per slot
    for the lock-step kernels, so that the thin late bounces of one workgroup overlap the first bounces of the next);
    HPT_WF_GENERATIONS sets the number.  However many there are, the sums are the oracle's bit for bit."""
    w, h, depth, spp = 200, 160, 9, 6
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    monkeypatch.setenv("HPT_WF_GENERATIONS", generations)
    gpu_ctx.upload(sc)
    gpu_ctx.accum_clear()
    gpu_ctx.render(2, 256, depth, 0, spp)
    got = gpu_ctx.resolve(spp)
    want_acc, _ = oracle.cornell(w, h).render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))


def test_wavefront_queues_shrink_when_the_card_is_full(hp, monkeypatch):
    """The streamed mode sizes its two path queues for the call (up to 2^28 paths, 32 GiB); when the card cannot give
    that much it halves the batch until the queues fit, and the sums are the same.  Card fullness is simulated with
    HPT_ALLOC_LIMIT (a single allocation beyond it is refused as out of memory)."""
    w, h, depth, spp = 512, 512, 6, 8                      # 2 M paths: 32 MiB per queue plane
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    ctx = hp.Context(0)
    try:
        ctx.upload(sc)
        ctx.accum_clear()
        ctx.render(1, 256, depth, 0, spp)
        want = ctx.accum_download()
        ctx.close()
        ctx = hp.Context(0)
        ctx.upload(sc)
        monkeypatch.setenv("HPT_ALLOC_LIMIT", str(5 << 20))   # 5 MiB: batches of 2^18 paths at most
        ctx.stats_reset()
        ctx.accum_clear()
        ctx.render(2, 256, depth, 0, spp)
        assert np.array_equal(ctx.accum_download(), want)
        assert ctx.stats()["launches"] >= 2 * 8            # generate + bounces per batch, at least 8 batches
        monkeypatch.setenv("HPT_ALLOC_LIMIT", "4096")       # nothing fits: the error comes out
        ctx.set_wavefront_capacity(1 << 24)
        ctx2 = hp.Context(0)
        try:
            with pytest.raises(hp.HptError, match="out of device memory"):
                ctx2.upload(sc)
        finally:
            monkeypatch.delenv("HPT_ALLOC_LIMIT")
            ctx2.close()
    finally:
        monkeypatch.delenv("HPT_ALLOC_LIMIT", raising=False)
        ctx.close()


def test_error_paths(hp, gpu_ctx):
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5, 16, 16)
    gpu_ctx.upload(sc)
    with pytest.raises(hp.HptError):
        gpu_ctx.render(3, 256, 5, 0, 1)          # bad mode
    with pytest.raises(hp.HptError):
        gpu_ctx.render(1, 0, 5, 0, 1)            # bad block
    with pytest.raises(hp.HptError):
        gpu_ctx.render(1, 2048, 5, 0, 1)
    fresh = hp.Context(0)
    with pytest.raises(hp.HptError):
        fresh.render(1, 256, 5, 0, 1)            # no scene
    fresh.close()
    d = sc.describe()
    d.num_nodes = 0
    with pytest.raises(hp.HptError):
        gpu_ctx.upload(d)
    # zero samples / zero depth are no-ops, not errors (castRay's loop does not run)
    gpu_ctx.upload(sc)
    gpu_ctx.accum_clear()
    gpu_ctx.render(1, 256, 0, 0, 4)
    gpu_ctx.render(2, 256, 5, 0, 0)
    assert not gpu_ctx.resolve(1).any()


@pytest.mark.parametrize("mode", [1, 2])
def test_full_resolution_render_matches_oracle_bit_exact(hp, oracle, gpu_ctx, mode):
    """The benchmark's resolution and depth (Cornell 1024 x 1024, depth 20) at 4 spp: 49 M rays per mode, every
    accumulator and both ray counters equal the oracle's -- a soak over the rarely taken branches (exact fallbacks of
    the triangle classification, axis-parallel rays, NaN resets) that the small images cannot reach."""
    w = h = 1024
    depth, spp = 20, 4
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    osc = oracle.cornell(w, h)
    gpu_ctx.upload(sc)
    gpu_ctx.stats_reset()
    gpu_ctx.accum_clear()
    gpu_ctx.render(mode, 256, depth, 0, spp)
    got = gpu_ctx.accum_download()
    st = gpu_ctx.stats()
    want, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    assert (st["closest_queries"], st["shadow_queries"]) == (ost["closest_queries"], ost["shadow_queries"])
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, "%d of %d pixels differ" % (bad.size, w * h)


@pytest.mark.parametrize("spp_per_unit", ["1", "3", "40"])
def test_unit_sizes_and_edge_tiles(hp, oracle, monkeypatch, spp_per_unit):
    """The single-kernel mode's work units at other sizes than the default (HPT_SPP_PER_UNIT: pools of 64, 192 and
    2 560 items for a full tile) on an image whose right and bottom tiles are partial (43 x 21: pools that are no
    multiple of the wave size, batches of camera directions that end inside a pool): accumulators bit-exact."""
    monkeypatch.setenv("HPT_SPP_PER_UNIT", spp_per_unit)
    ctx = hp.Context(0)
    try:
        w, h, depth, spp = 43, 21, 6, 7
        sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
        osc = oracle.cornell(w, h)
        ctx.upload(sc)
        want, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
        for block in (256, 100):
            ctx.stats_reset()
            ctx.accum_clear()
            ctx.render(1, block, depth, 0, spp)
            st = ctx.stats()
            assert (st["closest_queries"], st["shadow_queries"], st["paths"]) == (ost["closest_queries"], ost["shadow_queries"], w * h * spp)
            assert np.array_equal(ctx.accum_download(), want), (spp_per_unit, block)
    finally:
        ctx.close()


def test_cast_rays_equals_oracle_bit_exact(cornell, hp):
    """Scene::castRay on caller-supplied eye rays (hpt_cast_rays): radiance floats and query counters equal the
    oracle's for random rays inside the box with random stream keys, and for the camera rays of two samples, which
    also equal the paths of the render (oracle cast_paths)."""
    sc, osc, ctx = cornell
    depth = 8
    n = 1 << 15
    o, d = random_rays(n, seed=21)
    rs = np.random.RandomState(22)
    pk, sk = rs.randint(0, 1 << 20, n), rs.randint(0, 4096, n)
    want, ost = osc.cast_rays(RNG_PHILOX, TRIG_PORTABLE, depth, o, d, sk, pk)
    ctx.stats_reset()
    got = ctx.cast_rays(o, d, pk, sk, depth)
    st = ctx.stats()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (st["closest_queries"], st["shadow_queries"], st["paths"]) == (ost["closest_queries"], ost["shadow_queries"], n)
    assert want.max() > 0 and (want.sum(1) == 0).mean() < 0.9

    w = h = 512
    pix = np.arange(w * h)[:: 37]
    for k in (0, 5):
        bias = np.zeros(w * h, np.float32)
        bias[pix] = [osc.lib.orc_philox_uniform(int(m), k, 0xFFFFFFFF, 0, 0) for m in pix]
        o, d = primary_rays(w, h, (278, 273, -800), osc.info()["cam_scale"], bias.reshape(h, w))
        want, _ = osc.cast_paths(RNG_PHILOX, TRIG_PORTABLE, depth, np.full(len(pix), k), pix)
        got = ctx.cast_rays(o[pix], d[pix], pix, np.full(len(pix), k), depth)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))

    # depth 0, an empty batch, bad arguments
    assert not ctx.cast_rays(o[:100], d[:100], np.arange(100), np.zeros(100), 0).any()
    assert ctx.cast_rays(np.zeros((0, 3)), np.zeros((0, 3)), [], [], depth).shape == (0, 3)
    with pytest.raises(hp.HptError):
        ctx.cast_rays(o[:4], d[:4], np.arange(4), np.zeros(4), -1)
    with pytest.raises(hp.HptError):
        ctx.cast_rays_dev(4, 0, 0, 0, 0, depth, 0)
    with pytest.raises(hp.HptError):
        import torch
        t = torch.zeros(64, device="cuda")
        ctx.cast_rays_dev(4, t, t, t, t, depth, t, block=512)


def test_intersect_bounce_rays(cornell):
    """The rays camera paths trace at bounce 3 (captured from the oracle: incoherent, starting on surfaces, many of
    them grazing): closest hits bit-exact."""
    sc, osc, ctx = cornell
    pix = np.arange(512 * 512)[::2]
    o, d = osc.bounce_rays(RNG_PHILOX, TRIG_PORTABLE, 3, np.full(len(pix), 1), pix)
    assert len(o) > 50000
    _check_hits(ctx.intersect(o, d), osc.intersect(o, d), "bounce-3 rays")
