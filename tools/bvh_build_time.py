"""Times the device-side build of the traversal records (HPT_BVH_BUILD_GPU) for a few scenes, repeated within one process.
    python tools/bvh_build_time.py      (on the GPU box, from the repository root)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import time, cudapathtracer_amd as hp
ctx = hp.Context(0)
for name in ("heightfield:708", "bunny", "heightfield:708", "heightfield:1000"):
    sc = hp.Scene.CreateNamedScene(name, 5, 64, 64, bvh_build=hp.BVH_BUILD_GPU)
    t = time.time(); ctx.upload(sc); dt = time.time() - t
    print(name, ctx.build_info(), "upload wall %.1f ms" % (dt * 1e3), flush=True)
