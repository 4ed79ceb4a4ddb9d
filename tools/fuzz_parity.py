#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the CPU oracle (run on a GPU box):

    python tools/fuzz_parity.py [--cases N] [--first SEED] > gpurun_out/fuzz.txt

Every case draws a scene (the Cornell shell or a bare light, plus random triangle soups, boxes of the reference's
Cornell box, the bunny and spheres, with random Lambertian / microfacet / glass materials and sometimes a second
emitter), an image size, depth, sample range, seed, render mode, workgroup size, integrator options and BVH builder (reference / device / SAH / the default),
and compares -- bit for bit -- the fixed-point accumulators and the ray counters of hpt_render with the oracle's, the
closest hits of a random ray batch, the occlusion answers of a shadow batch and the radiance of hpt_cast_rays.
Prints one line per case and stops at the first difference with the case's seed (exit status 1)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime: torch first)
import cudapathtracer_amd as hp  # noqa: E402
from cudapathtracer_amd import scenes  # noqa: E402
from orclib import RNG_PHILOX, TRIG_PORTABLE, Material, Oracle, random_rays, shadow_rays  # noqa: E402


def random_material(rs, allow_glass=True):
    kind = rs.choice(["lambert", "metal", "rough", "glass"] if allow_glass else ["lambert", "metal", "rough"])
    albedo = tuple(float(v) for v in rs.rand(3) * 0.9 + 0.05)
    if kind == "lambert":
        return dict(type=0, albedo=albedo)
    if kind == "metal":
        return dict(type=1, albedo=albedo, metallic=float(rs.choice([1.0, 0.5, 0.0])), roughness=float(rs.choice([0.005, 0.05, 0.3])))
    if kind == "rough":
        return dict(type=1, albedo=albedo, metallic=float(rs.rand()), roughness=float(rs.rand() * 0.9 + 0.1))
    return dict(type=2, roughness=float(rs.choice([0.01, 0.1, 0.4])), ior=float(rs.choice([1.33, 1.5, 2.4])))


def soup(rs, n, size):
    c = rs.rand(n, 1, 3) * [400, 400, 400] + [78, 70, 80]
    return (c + (rs.rand(n, 3, 3) - 0.5) * size).astype(np.float32)


def build_case(rs, orc, big=False):
    """-> (hp scene, oracle scene, description)"""
    if big:
        w, h = int(rs.randint(64, 400)), int(rs.randint(64, 300))
        depth = int(rs.choice([3, 8, 20]))
    else:
        w, h = int(rs.randint(6, 72)), int(rs.randint(6, 56))
        depth = int(rs.choice([1, 2, 3, 5, 8, 12]))
    items = []  # ("tris", triangles, material) | ("sphere", centre, radius, material)
    what = []
    light = dict(type=0, emission=scenes.light_emission())
    if rs.rand() < 0.75:
        for name, tris, mat in scenes.cornell_shell_triangles():
            items.append(("tris", tris, mat))
        what.append("shell")
    else:
        quad = np.array([[[100, 500, 100], [400, 500, 100], [400, 500, 400]],
                         [[100, 500, 100], [400, 500, 400], [100, 500, 400]]], np.float32)[:, ::-1]
        items.append(("tris", quad, light))
        what.append("quad light")
    for _ in range(int(rs.randint(0, 4))):
        kind = rs.choice(["soup", "soup", "box", "sphere", "bunny", "emitter"], p=[0.3, 0.2, 0.15, 0.2, 0.05, 0.1])
        if kind == "soup":
            n = int(rs.choice([1, 2, 7, 20, 31, 40, 90, 300]))
            tris = soup(rs, n, float(rs.choice([30.0, 120.0, 300.0])))
            if n >= 7 and rs.rand() < 0.2:   # zero-area triangles: two equal vertices, a point
                tris[1, 2] = tris[1, 1]
                tris[4] = tris[4, 0]
                what.append("degenerate")
            items.append(("tris", tris, random_material(rs)))
            what.append("soup%d" % n)
        elif kind == "box":
            name = rs.choice(["shortbox", "tallbox"])
            items.append(("tris", scenes.read_obj_triangles(os.path.join(scenes.DEFAULT_MODELS_DIR, "cornellbox", name + ".obj")), random_material(rs)))
            what.append(str(name))
        elif kind == "sphere":
            c = tuple(float(v) for v in rs.rand(3) * [400, 300, 400] + [78, 80, 80])
            items.append(("sphere", c, float(rs.rand() * 80 + 20), random_material(rs)))
            what.append("sphere")
        elif kind == "bunny":
            items.append(("tris", scenes.bunny_triangles(), random_material(rs)))
            what.append("bunny")
        else:
            e = tuple(float(v) for v in rs.rand(3) * 20 + 1)
            items.append(("tris", soup(rs, int(rs.randint(1, 4)), 80.0), dict(type=0, emission=e)))
            what.append("emitter")
    if rs.rand() < 0.1:   # many objects = many materials: more than the kernels keep in LDS
        for _ in range(14):
            items.append(("tris", soup(rs, int(rs.randint(1, 3)), 60.0), random_material(rs)))
        what.append("14 small")
    order = rs.permutation(len(items))
    sc = hp.Scene(w, h, camPos=scenes.CORNELL_CAM, fov=40.0, maxDepth=depth)
    osc = orc.scene(w, h, scenes.CORNELL_CAM, 40.0)
    for i in order:
        it = items[i]
        if it[0] == "tris":
            sc.Add(it[1], hp.Material(**it[2]))
            osc.add_triangles(it[1], Material.make(**it[2]))
        else:
            sc.AddSphere(it[1], it[2], hp.Material(**it[3]))
            osc.add_sphere(it[1], it[2], Material.make(**it[3]))
    build = rs.choice(["host", "gpu", "sah", "auto"], p=[0.3, 0.2, 0.2, 0.3])   # auto = the library's default
    # The same mesh twice = coincident triangles with different materials: every ray that hits them hits two triangles
    # at bitwise the same distance.  The reference keeps the one its traversal meets first, and so does every host-built
    # tree here (records in the reference's order); the device-built tree documents that it may name the other one
    # (include/hpt.h, HPT_BVH_BUILD_GPU), which then shades with the other material.  Such scenes take a host build.
    meshes_drawn = [x for x in what if x in ("shortbox", "tallbox", "bunny")]
    if build == "gpu" and len(set(meshes_drawn)) != len(meshes_drawn):
        build = "host"
    if build == "auto":
        sc.buildBVH()
    elif build == "sah":
        sc.buildBVH(sah=True)
    else:
        sc.buildBVH(gpu=bool(build == "gpu"))
    osc.build()
    # gpu_build: device-built records keep their own leaf order (ids may differ where several triangles are hit at one
    # distance); the SAH tree keeps the reference's record order, ids included
    return sc, osc, "%dx%d d%d %s%s" % (w, h, depth, "+".join(what), "" if build == "host" else " %s-bvh" % build), depth, build == "gpu"


class Mismatch(AssertionError):
    pass


def run(cases, first=0, out=sys.stdout, big=False):
    """Runs cases [first, first + cases); raises Mismatch (message = the case line) at the first difference.
    big: images of 64..400 x 64..300 pixels with up to 40 samples (several work units per tile, chunks of samples)."""
    orc = Oracle()
    ctx = hp.Context(0)
    try:
        _run(orc, ctx, cases, first, out, big)
    finally:
        ctx.close()


def _run(orc, ctx, cases, first, out, big):
    t0 = time.time()
    for case in range(first, first + cases):
        rs = np.random.RandomState(1000 + case)
        sc, osc, desc, depth, gpu_build = build_case(rs, orc, big)
        rr, reg = bool(rs.rand() < 0.3), bool(rs.rand() < 0.3)
        seed = int(rs.choice([0, 1, 0xDEADBEEF]))
        s0, spp = int(rs.choice([0, 0, 5, 1000])), int(rs.choice([1, 7, 16, 17, 40]) if big else rs.randint(1, 4))
        mode, block = int(rs.choice([1, 2])), int(rs.choice([64, 100, 256, 256, 512]))
        ctx.upload(sc)
        ctx.set_integrator(rr, reg)
        ctx.set_seed(seed)
        osc.set_integrator(rr, reg)
        variant = ctx.stats()["variant"]
        line = "case %4d  %-44s variant %d mode %d block %4d spp [%d,+%d) seed %x%s%s" % (
            case, desc, variant, mode, block, s0, spp, seed, " rr" if rr else "", " reg" if reg else "")

        def fail(msg):
            print(line + "  FAIL: " + msg, file=out, flush=True)
            raise Mismatch(line + "  FAIL: " + msg)

        want, ost = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, sample_begin=s0, seed=seed, fixed=True)
        ctx.stats_reset()
        ctx.accum_clear()
        # how the sample range is rendered: in one call; split over two calls (the sums do not depend on the split);
        # by the entry point that allocates nothing; in streamed batches of a few thousand paths
        how = rs.choice(["one call", "split", "static", "small batches"])
        ctx.set_wavefront_capacity(int(rs.choice([4096, 20000])) if how == "small batches" else 1 << 28)
        if how == "split" and spp > 1:
            ctx.render(mode, block, depth, s0 + 1, spp - 1)
            ctx.render(3 - mode, block, depth, s0, 1)
        elif how == "static" and block <= 256 and ((sc.width + 7) // 8) * ((sc.height + 7) // 8) * 64 <= 1 << 18:
            per_call = max(1, (1 << 18) // (((sc.width + 7) // 8) * ((sc.height + 7) // 8) * 64))   # its limit: 2^18 camera paths
            for s in range(s0, s0 + spp, per_call):
                ctx.render_static(mode, block, depth, s, min(per_call, s0 + spp - s))
        else:
            ctx.render(mode, block, depth, s0, spp)
        line += "  " + how
        got = ctx.accum_download()
        st = ctx.stats()
        if not np.array_equal(got, want):
            fail("%d accumulator words differ" % int((got != want).sum()))
        if (st["closest_queries"], st["shadow_queries"]) != (ost["closest_queries"], ost["shadow_queries"]):
            fail("ray counters %r vs %r" % ((st["closest_queries"], st["shadow_queries"]), (ost["closest_queries"], ost["shadow_queries"])))
        # ray batches (a GPU-built tree may name another of several triangles that are hit at bitwise equal t)
        o, d = random_rays(4096, case)
        # some axis-parallel rays (zero direction components: infinite reciprocals, the NaN-aware box test) and rays
        # that start on the planes of the room
        d[:256, rs.randint(0, 3)] = 0.0
        d[:64, rs.randint(0, 3)] = 0.0
        nrm = np.sqrt((d[:256].astype(np.float64) ** 2).sum(1))
        keep = nrm > 0
        d[:256][keep] = (d[:256][keep] / nrm[keep, None]).astype(np.float32)
        d[:256][~keep] = (0, 0, 1)
        o[256:512, 1] = 0.0
        gp, gt = ctx.intersect(o, d)
        wp, wt = osc.intersect(o, d)
        if not np.array_equal(gt.view(np.int64), wt.view(np.int64)):
            fail("closest t differs on %d rays" % int((gt.view(np.int64) != wt.view(np.int64)).sum()))
        if not gpu_build and not np.array_equal(gp, wp):
            fail("closest ids differ on %d rays" % int((gp != wp).sum()))
        o, d, sh, tm = shadow_rays(4096, case + 7)
        gp, gt = ctx.intersect(o, d, sh, tm)
        wp, wt = osc.intersect(o, d, sh, tm)
        if not np.array_equal(gp >= 0, wp >= 0):
            fail("occlusion differs on %d rays" % int(((gp >= 0) != (wp >= 0)).sum()))
        # castRay on free rays
        o, d = random_rays(2048, case + 13)
        pk, sk = rs.randint(0, 1 << 20, 2048), rs.randint(0, 64, 2048)
        wantc, _ = osc.cast_rays(RNG_PHILOX, TRIG_PORTABLE, depth, o, d, sk, pk, seed=seed)
        gotc = ctx.cast_rays(o, d, pk, sk, depth)
        if not np.array_equal(gotc.view(np.uint32), wantc.view(np.uint32)):
            fail("cast_rays differs on %d rays" % int((gotc.view(np.uint32) != wantc.view(np.uint32)).any(1).sum()))
        print(line + "  ok", file=out, flush=True)
        osc.close()
    print("%d cases bit-exact in %.0f s" % (cases, time.time() - t0), file=out, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--big", action="store_true", help="larger images and more samples per case")
    args = ap.parse_args()
    try:
        run(args.cases, args.first, big=args.big)
    except Mismatch:
        sys.exit(1)


if __name__ == "__main__":
    main()
