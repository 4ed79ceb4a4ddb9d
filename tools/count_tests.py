#!/usr/bin/env python3
"""Box and triangle tests the kernels EXECUTE per ray, per scene and mode, counted by the
diagnostic build of the library (`make count` -> cudapathtracer_amd/libhpt_count.so).

    HPT_LIBHPT=$PWD/cudapathtracer_amd/libhpt_count.so python tools/count_tests.py > profiles/executed_tests.json

The counts are deterministic (they do not depend on timing or on which lane traces which path),
so the file is committed and bench.py reads it to state executed work beside the
reference-defined (algorithmic) work of its roofline block.  Run on the GPU box."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if "count" not in os.environ.get("HPT_LIBHPT", ""):
        sys.exit("set HPT_LIBHPT to cudapathtracer_amd/libhpt_count.so (make count)")
    import cudapathtracer_amd as hp
    from cudapathtracer_amd import scenes
    out = {}
    cases = [("cornell", 1024, 1024, 20, 4), ("glass", 1024, 1024, 20, 4), ("bunny", 1024, 1024, 20, 2),
             ("heightfield:708", 2048, 2048, 20, 1)]
    for name, w, h, depth, spp in cases:
        if name == "cornell":
            sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
        elif name == "glass":
            sc = hp.Scene.CreateBuiltinScene(hp.Scene.GlassCornell, depth, w, h)
        elif name == "bunny":
            sc = scenes.bunny_in_cornell(w, h, depth)
        else:
            sc = scenes.heightfield_in_cornell(int(name.split(":")[1]), w, h, depth)
        for traversal in ("reference", "culled"):
            if traversal == "culled" and name in ("cornell", "glass"):
                continue
            ctx = hp.Context(0)
            ctx.set_traversal(1 if traversal == "culled" else 0)
            ctx.upload(sc)
            for mode in (1, 2):
                ctx.stats_reset()
                ctx.accum_clear()
                ctx.render(mode, 256, depth, 0, spp)
                st = ctx.stats()
                rays = st["closest_queries"] + st["shadow_queries"]
                key = "%s_%dx%d_d%d_mode%d" % (name, w, h, depth, mode) + ("" if traversal == "reference" else "_culled")
                out[key] = {"variant": st["variant"], "spp": spp, "rays": rays, "rays_per_path": rays / st["paths"],
                            "executed_box_tests_per_ray": st["box_tests"] / rays,
                            "executed_tri_tests_per_ray": st["tri_tests"] / rays}
            ctx.close()
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
