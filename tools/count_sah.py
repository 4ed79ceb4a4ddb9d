#!/usr/bin/env python3
"""Box and triangle tests per ray the kernels execute on the reference's median-split tree and on the surface-area
tree of the same scene (diagnostic counting build):

    HPT_LIBHPT=$PWD/cudapathtracer_amd/libhpt_count.so python tools/count_sah.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "count" not in os.environ.get("HPT_LIBHPT", ""):
    sys.exit("set HPT_LIBHPT to cudapathtracer_amd/libhpt_count.so (make count)")
import cudapathtracer_amd as hp  # noqa: E402

ctx = hp.Context(0)
for name, res, spp in (("bunny", 1024, 2), ("heightfield:708", 2048, 1)):
    for build, label in ((hp.BVH_BUILD_HOST, "median split (reference)"), (hp.BVH_BUILD_SAH, "surface-area heuristic")):
        sc = hp.Scene.CreateNamedScene(name, 20, res, res, bvh_build=build)
        ctx.upload(sc)
        ctx.stats_reset()
        ctx.accum_clear()
        ctx.render(2, 256, 20, 0, spp)
        st = ctx.stats()
        rays = st["closest_queries"] + st["shadow_queries"]
        print("%-16s %-26s %6.2f box tests, %5.2f triangle tests per ray; stack depth %d" % (
            name, label, st["box_tests"] / rays, st["tri_tests"] / rays, ctx.build_info()["stack_depth"]))
